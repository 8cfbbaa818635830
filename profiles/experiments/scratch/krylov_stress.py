"""Krylov fall-back of the overlap solves on the device against dense eigen-solves: Haar-far candidates and constructed near-ties.
usage: krylov_stress.py D n_far n_tie [max_rounds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from scipy.linalg import expm
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
import overlap_cases as OC

D = int(sys.argv[1]); n = int(sys.argv[2]); n_tie = int(sys.argv[3])
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
rng = np.random.default_rng(11)
WW = expm(-1j * 0.2 * O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}))
eng = EnergyEngine(D, max(n, n_tie, 16))
A = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, n))
C = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, n))
for adj in (False,):
    t0 = time.time()
    eta, rounds, st, r = eng.overlaps(A, C, WW, max_rounds=cap, tol=1e-12, want_r=True)
    t1 = time.time()
    eta, rounds, st, r = eng.overlaps(A, C, WW, max_rounds=cap, tol=1e-12, want_r=True)
    t2 = time.time()
    worst = 0.0; wrong = 0; resid = 0.0
    for b in range(n):
        E = OC.dense_map(A[b], C[b], WW)
        w = np.linalg.eigvals(E); ref = w[np.argmax(np.abs(w))]
        err = abs(eta[b] - ref)
        if st[b] == 0:
            worst = max(worst, err); wrong += err > 1e-10
            x = r[b].reshape(-1)
            resid = max(resid, np.linalg.norm(E @ x - eta[b] * x))
    print(f'D={D} far n={n}: status0 {int((st == 0).sum())}, WRONG {wrong}, worst err {worst:.2e}, worst residual of r {resid:.2e}; map applications mean {rounds.mean():.0f} '
          f'median {np.median(rounds):.0f} max {rounds.max()}; wall {t2 - t1:.4f} s (first call {t1 - t0:.3f})')
for ratio in (1 - 1e-4, 1 - 1e-6, 1 - 1e-8, 1.0):
    At = []; Ct = []
    for k in range(n_tie):
        a, c = OC.near_tie_pair(rng, D, ratio, WW)
        At.append(a); Ct.append(c)
    At = np.array(At); Ct = np.array(Ct)
    eta, rounds, st = eng.overlaps(At, Ct, WW, max_rounds=cap, tol=1e-12)
    errs = []
    for k in range(n_tie):
        ref, rat = OC.dominant(At[k], Ct[k], WW)
        errs.append(abs(eta[k] - ref))
    print(f'D={D} tie ratio {ratio}: status {st.tolist()} rounds {rounds.tolist()} max err {max(errs):.2e}')
