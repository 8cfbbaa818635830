import sys, numpy as np
sys.path.insert(0,'/root/repo')
from scipy.linalg import expm
from qmps_amd import new_time_evolve as NT, represent as R
from oracle import qmps_oracle as O
h=O.hamiltonian_matrix({'ZZ':-1.0,'X':1.0}); WW=expm(-0.05j*h)
for D,P in ((16,8),(8,6),(4,4)):
    X0=np.random.default_rng(9).standard_normal((8,P))
    out={}
    for native in (True, False):
        H,info=NT.evolve(X0,WW,30,method='BFGS',D=D,state_tensor=R.ShallowCNOTStateTensor,options={'maxiter':60,'carry_hessian':True,'native':native},return_info=True)
        out[native]=(H,np.array([f[-1] for f in info['fun']]),info['nit'])
    df=np.abs(out[True][1]-out[False][1]).max(); dx=np.abs(out[True][0]-out[False][0]).max()
    print('D',D,'30 steps: max |f_native - f_numpy| %.2e, max |x diff| %.2e, nit native %s numpy %s, final f %.8f'%(df,dx,sum(out[True][2]),sum(out[False][2]),out[True][1][-1].mean()))
