import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
rng=np.random.default_rng(1); B=65536
A=O.unitary_to_tensor(O.haar_unitaries(rng,8,B))
eng=EnergyEngine(4,B); eng.set_tensors(A); eng.set_hamiltonian(O.hamiltonian_matrix({'ZZ':-1,'X':1}))
for _ in range(5): eng.launch(B); eng.cost_launch(B)
eng.sync()
for rep in range(12):
    eng.timer_begin()
    for _ in range(100): eng.launch(B); eng.cost_launch(B)
    print('%.5f ms/step' % (eng.timer_end()/100))
