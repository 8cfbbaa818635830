"""D = 8 overlap power method with deflation steps against dense eigen-solves: far (Haar-random) and near candidates."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import expm
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
D = 8
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = 3000
eng = EnergyEngine(D, n)
WW = expm(-1j * 0.2 * O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}))
for name, eps in (('far (Haar)', None), ('near 0.1', 0.1), ('near 0.5', 0.5)):
    U = O.haar_unitaries(rng, 2 * D, n)
    A = O.unitary_to_tensor(U)
    if eps is None:
        C = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, n))
    else:
        K = rng.standard_normal((n, 2 * D, 2 * D)) + 1j * rng.standard_normal((n, 2 * D, 2 * D))
        K = K - K.conj().transpose(0, 2, 1)
        C = O.unitary_to_tensor(np.stack([expm(eps * K[b] / np.linalg.norm(K[b])) @ U[b] for b in range(n)]))
    eta, rounds, st = eng.overlaps(A, C, WW, max_rounds=20000)
    bad = 0
    worst = 0.0
    for b in range(n):
        if st[b] == 0:
            ref = O.overlap_eta(A[b], C[b], WW)[0]
            err = abs(eta[b] - ref)
            worst = max(worst, err)
            bad += err > 1e-9
    print('%-12s converged %d / %d, WRONG %d, worst error of the rest %.2e, rounds mean %.1f max %d' % (name, (st == 0).sum(), n, bad, worst, rounds.mean(), rounds.max()))
