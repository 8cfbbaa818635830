import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine
rng=np.random.default_rng(1)
h=O.hamiltonian_matrix({'ZZ':-1,'X':1})
for B in (1,2,3,16,17,40):
    A=O.unitary_to_tensor(O.haar_unitaries(rng,8,B))
    with EnergyEngine(4,64) as eng:
        E,it,st=eng.energies(A,h)
        eng.set_solver('squaring',handoff=0)
        E2,it2,st2=eng.energies(A,h)
    print(B,'iters',it[:20],'st',st[:20],'dE',np.abs(E-E2)[:,0].round(12)[:20])
