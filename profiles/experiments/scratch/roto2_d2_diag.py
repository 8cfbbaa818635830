import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import qmps_oracle as O
from oracle import c_oracle as C
from qmps_amd import EnergyEngine, _lib as L
C.build()
D, kind = 2, 0
rng = np.random.default_rng(300 * D + kind)
R, sweeps = 24, 2
P0 = rng.standard_normal((R, 2))
h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
eng = EnergyEngine(D, 4096); eng.set_hamiltonian(h)
es, p = eng.double_rotosolve(kind, P0, sweeps)
E, it, st = eng.energies_from_params(kind, p, h)
A = O.unitary_to_tensor(np.stack([O.shallow_cnot_unitary(D, q) for q in p]))
for b in range(R):
    Eo = O.energy_closed_form(A[b], h)
    r = O.right_environment(A[b]) if hasattr(O,'right_environment') else None
    d = abs(Eo - es[-1][b])
    if d > 1e-10:
        T = sum(np.kron(A[b][s], A[b][s].conj()) for s in range(2))
        ev = np.sort(np.abs(np.linalg.eigvals(T)))[::-1]
        print(b, 'rec', es[-1][b], 'dev eval at p', E[b,0], 'oracle', Eo, 'diff rec-oracle %.2e dev-oracle %.2e' % (d, abs(E[b,0]-Eo)), 'it', it[b], 'st', st[b], 'p', p[b], 'transfer |eig|', ev[:3])
