"""overlap power method with deflation steps (D = 8, 16) against the plain one and dense eigen-solves: far (Haar-random) and near
candidates.  usage: deflation_stress.py [seed] [D] [n] [dense-check stride]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import expm
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1
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
    os.environ['QMPS_NO_DEFLATION'] = '1'
    eta_p, rounds_p, st_p = eng.overlaps(A, C, WW, max_rounds=20000)
    del os.environ['QMPS_NO_DEFLATION']
    eta, rounds, st = eng.overlaps(A, C, WW, max_rounds=20000)
    both = (st == 0) & (st_p == 0)
    print('   plain: converged %d, rounds mean %.1f max %d, ten slowest %s | with deflation: ten slowest %s; max |eta - eta_plain| %.2e' % ((st_p == 0).sum(), rounds_p.mean(), rounds_p.max(), np.sort(rounds_p)[-10:], np.sort(rounds)[-10:], np.abs(eta - eta_p)[both].max()))
    bad = 0
    worst = 0.0
    for b in range(0, n, stride):
        if st[b] == 0:
            ref = O.overlap_eta(A[b], C[b], WW)[0]
            err = abs(eta[b] - ref)
            worst = max(worst, err)
            bad += err > 1e-9
    print('%-12s converged %d / %d, WRONG %d, worst error of the rest %.2e, rounds mean %.1f max %d' % (name, (st == 0).sum(), n, bad, worst, rounds.mean(), rounds.max()))
