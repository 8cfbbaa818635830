"""Round 5: would a small restarted Krylov space beat the power method on the mixed transfer maps of config 4 (no: profiles/EXPERIMENTS.md)?
Usage (repo root): python profiles/experiments/r05/accel_sim.py <npz of d16_trajectory_probe.py>"""
import numpy as np, sys
sys.path.insert(0, '.')
from oracle import qmps_oracle as O
z = np.load(sys.argv[1] if len(sys.argv) > 1 else 'probe3.npz')      # written by d16_trajectory_probe.py (4th argument)
X=z['x_trajectory']; WW=z['WW']
rng=np.random.default_rng(1)
def Tmat(xa, xb):
    A=O.unitary_to_tensor(O.shallow_cnot_unitary(16, xa)); B=O.unitary_to_tensor(O.shallow_cnot_unitary(16, xb))
    C=np.tensordot(WW,O.merge(A,A),[1,0]); Bm=O.merge(B,B)
    return O.transfer_matrix(C,Bm)
def dominant(T):
    w,v=np.linalg.eig(T); k=np.argmax(abs(w)); return w[k], v[:,k]/np.linalg.norm(v[:,k])
def power(T,x,tol,maxit=2000):
    x=x/np.linalg.norm(x)
    for k in range(1,maxit+1):
        n=T@x; eta=np.vdot(x,n); res=np.linalg.norm(n-eta*x)
        if res<tol: return k
        x=n/np.linalg.norm(n)
    return maxit
def arnoldi_restart(T,x,tol,m,maxit=2000):
    # restarted Arnoldi(m): build m-dim Krylov space, Rayleigh-Ritz, restart with the dominant Ritz vector; counts applications
    x=x/np.linalg.norm(x); apps=0
    while apps<maxit:
        V=[x]; W=[]
        for j in range(m):
            w=T@V[j]; apps+=1; W.append(w)
            if j==0:
                eta=np.vdot(V[0],w); res=np.linalg.norm(w-eta*V[0])
                if res<tol: return apps
            if j<m-1:
                h=w.copy()
                for _ in range(2):
                    for v in V: h-=np.vdot(v,h)*v
                nh=np.linalg.norm(h)
                if nh<1e-14: break
                V.append(h/nh)
        Vm=np.array(V).T; Wm=np.array(W).T
        k=min(Vm.shape[1],Wm.shape[1]); Vm=Vm[:,:k]; Wm=Wm[:,:k]
        H=Vm.conj().T@Wm
        w,S=np.linalg.eig(H); i=np.argmax(abs(w))
        x=Vm@S[:,i]; x/=np.linalg.norm(x)
    return maxit
def run(t, dstep, tol):
    x0=X[t]; d=rng.standard_normal(8); d*=dstep/np.linalg.norm(d)
    Told=Tmat(x0,x0); _,r0=dominant(Told)
    Tnew=Tmat(x0,x0+d)
    out=[power(Tnew,r0,tol)]
    for m in (2,3,4,6): out.append(arnoldi_restart(Tnew,r0,tol,m))
    return out
for t in (84,247,47,0,1,100):
    for dstep,tol in ((1e-2,1e-7),(1e-3,1e-8)):
        print(t,dstep,tol,'power / arnoldi m=2,3,4,6:',run(t,dstep,tol))
