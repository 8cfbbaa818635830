"""CPU emulation (numpy) of overlap_quad_charpoly (qmps_amd/csrc/qmps_evolve_d2.hip) - experiment tooling, not product and not oracle: the
starting points, termination and hand-back rules of the Aberth iteration were tuned here over maps of the special grid before they went into the
kernel.  The hand-back (`clustered`) is answered with numpy eigvals here; the kernel runs its squaring solve."""
import numpy as np
U4=np.array([1.0,0.8,1.25,0.9])*np.exp(1j*np.array([0.7,2.1,4.0,5.3]))
U3=np.array([1.0,0.8,1.2])*np.exp(1j*np.array([0.7,2.6,4.9]))
def solve(E, BS=1e-4, CL=1e-6, SMALL=1e-6, E0=10, KAP=1e-6):      # (the kernel's values)
    """returns eta, iterations, fallback flag"""
    m2=(abs(E)**2).sum()
    if not m2>1e-300: return 0.0,0,False
    sc=1/np.sqrt(m2); X=E*sc; X2=X@X
    p1=np.trace(X); p2=np.trace(X2); p3=np.trace(X2@X); p4=np.trace(X2@X2)
    e1=p1; e2=0.5*(e1*p1-p2); e3=(e2*p1-e1*p2+p3)/3; e4=0.25*(e3*p1-e2*p2+e1*p3-p4)
    P=lambda z: (((z-e1)*z+e2)*z-e3)*z+e4
    dP=lambda z: ((4*z-3*e1)*z+2*e2)*z-e3
    n3=abs(p3)**2; n4=abs(p4)**2
    informed = n3>1e-12 and n4<4*n3
    if informed:
        a=p4/p3; b2=a-e1; b1=e2+a*b2; b0=-e3+a*b1; c3=-b2/3
        A3=(3*c3+2*b2)*c3+b1
        Q=((c3+b2)*c3+b1)*c3+b0; rad3=abs(Q)**(1/3)
        if rad3<1e-3: rad3=max(rad3, abs(A3)**0.5)
        z=np.concatenate([[a], c3+rad3*U3])
    else:
        c4=e1/4; rad4=abs(P(c4))**0.25
        if rad4<1e-3:
            A=(6*c4-3*e1)*c4+e2; B=dP(c4)
            rad4=max(rad4, abs(A)**0.5, abs(B)**(1/3))
        z=c4+rad4*U4
    prev=np.full(4,1e300); it=0; clustered=False
    for it in range(40):
        Pz=P(z); d=dP(z); w=np.where(abs(d)>0, Pz/np.where(d==0,1,d), 0)
        s=np.zeros(4,complex)
        for k in range(4):
            for j in range(4):
                if j!=k and z[k]!=z[j]: s[k]+=1/(z[k]-z[j])
        g=1-w*s; step=np.where(abs(g)>0,w/np.where(g==0,1,g),0)
        z=z-step
        s2=abs(step)**2; z2=abs(z)**2; zmax=z2.max()
        settled=~(s2>1e-28*z2+1e-300) | ((s2<1e-20*z2)&(s2>0.04*prev)); prev=s2
        below=((np.sqrt(z2)+4*np.sqrt(s2))**2<zmax)&(s2<BS*zmax)
        if np.all(settled|below): break
        if it+1>=E0:
            k=np.argmax(z2)
            if abs(d[k])<KAP*zmax**1.5: clustered=True; break      # (|P'| at the pre-step point of the largest root's lane)
    k=np.argmax(abs(z)); d2=abs(z-z[k])**2; zmax=abs(z[k])**2
    kappa=np.prod([abs(z[k]-z[j]) for j in range(4) if j!=k])
    near=min(abs(z[k]-z[j]) for j in range(4) if j!=k)
    if kappa<KAP*zmax**1.5 or near**2<CL*zmax or zmax<SMALL or not np.isfinite(z[k]): clustered=True      # (KAP: the conditioning of the largest root, prod_j |1 - z_j / z|)
    if clustered:
        return max(abs(np.linalg.eigvals(E))), it+1, True
    return z[k]/sc, it+1, False
