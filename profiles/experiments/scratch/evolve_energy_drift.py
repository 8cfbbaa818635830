import sys, numpy as np
sys.path.insert(0,'/root/repo')
from scipy.linalg import expm
from qmps_amd import new_time_evolve as NT, represent as R, _runtime, _lib as L
from oracle import qmps_oracle as O
h=O.hamiltonian_matrix({'ZZ':-1.0,'X':1.0})
for D,P in ((4,4),(16,8)):
    rng=np.random.default_rng(5)
    X0=rng.standard_normal((4,P))
    dt=0.02; WW=expm(-1j*dt*h)
    H,info=NT.evolve(X0,WW,40,method='BFGS',D=D,state_tensor=R.ShallowCNOTStateTensor,options={'maxiter':60,'carry_hessian':True,'speculative':True},return_info=True)
    eng=_runtime.engine(D,4)
    Es=[]
    for k in range(0,41,5):
        E,it,st=eng.energies_from_params(L.ANSATZ_SHALLOW_CNOT,H[k],h)
        Es.append(E[:,0])
    Es=np.array(Es)
    print('D',D,'final objective per step (last 3):',[float(np.mean(f[-1])) for f in info['fun'][-3:]])
    print(' energy per site at t=0, 0.1 .. 0.8:\n',np.round(Es.T,5))
