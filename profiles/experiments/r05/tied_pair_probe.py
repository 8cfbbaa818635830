import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import qmps_oracle as O
import evolve_replay as ER
from scipy.linalg import expm
from qmps_amd import EnergyEngine
H=O.hamiltonian_matrix({'ZZ':-1.0,'X':1.0})
xp=np.array([-0.5535838447638809, 0.5535837654426777]); WW=expm(-1j*0.05*H)
A=ER.tensor(0,2,xp)
eng=EnergyEngine(2,1024)
for cand in (xp, xp+np.array([1e-6,0]), xp+np.array([0,-1e-6]), np.array([-0.5535838,0.5535838])):
    eta,rounds,st=eng.overlaps(A[None],cand[None],WW,kind='params',ansatz=0,tol=1e-13)
    print(cand, eta, abs(eta), rounds, st)
r=eng.evolve_bfgs(0, xp[None], WW, n_steps=1, maxiter=60, tol=1e-13, counters=False)
print('host', r['fun_start'], r['fun'], r['nit'])
d=eng.evolve_bfgs_device(0, xp[None], WW, n_steps=1, maxiter=60, tol=1e-13)
print('dev', d['fun_start'], d['fun'], d['nit'], d['failed_evaluations'])
