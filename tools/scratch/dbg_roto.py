import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'tests'))
import numpy as np
from oracle import qmps_oracle as O, c_oracle as C
import test_rotosolve_gpu as T
from qmps_amd import EnergyEngine
C.build()
D,kind=4,0
name,builder,per=T.KINDS[kind]
rng=np.random.default_rng(100*D+kind)
R,sweeps=24,3
P0=rng.standard_normal((R,per*2))
h=O.hamiltonian_matrix({'ZZ':-1,'X':1})
es_ref,p_ref,bad,_=T.oracle_trajectory(C,builder,D,P0,h[None],sweeps,False)
eng=EnergyEngine(D,4096); eng.set_hamiltonian(h)
es,p=eng.rotosolve(kind,P0,sweeps)
d=np.abs(es-es_ref)
print('bad',bad.nonzero()[0])
for r in range(R):
    print(r,'dE',d[:,r],'dp',np.abs(T.wrap(p[r]-p_ref[r])).round(9))
# step-by-step: single update of param 0 on device vs oracle
es1,p1=eng.rotosolve(kind,P0,1)
# emulate first update only in oracle
params=P0.copy()
batch=np.repeat(params[:,None,:],3,axis=1); batch[:,:,0]+=T.SHIFTS3
e,st=T.oracle_energies(C,builder,D,batch.reshape(-1,4),h[None]); e=e.reshape(R,3)
eng.set_ansatz_params(kind,batch.reshape(-1,4)); eng.launch(); Eg,itg,stg=eng.results()
print('first-shift energies max diff',np.abs(Eg[:,0].reshape(R,3)-e).max(),'status',np.unique(stg),np.unique(st))
