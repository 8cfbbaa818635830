import sys, numpy as np
sys.path.insert(0, '/root/repo')
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O, c_oracle as C
rng = np.random.default_rng(5)
for B in (100003, 262147):
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    eng = EnergyEngine(4, B)
    E, it, st = eng.energies(A, h)
    idx = np.concatenate([np.arange(0, 300), np.arange(B - 300, B), rng.integers(0, B, 400)])
    ref = C.energy_batch(A[idx], h, handoff=0, skip=6, period=4, threads=8)
    print(B, 'max|dE|', np.abs(E[idx, 0] - ref['E'][:, 0]).max(), 'iters equal', (it[idx] == ref['iters']).mean(), 'status', np.bincount(st), 'cost', eng.summed_cost()[0] - E.sum())
    eng.close()
