"""Round 5: randomised stress of the time-evolution OVERLAP path against the oracle (not part of the suite): candidates given as ansatz parameters,
references as tensors, D in {2, 4, 8, 16}, W in {1, exp(-i dt h)}, candidates near the reference state, far from it, and on the special-angle grid.
Looking for SILENT errors: status 0 with an eigenvalue that is not the dominant one of the dense mixed transfer matrix, or status != 0 where the
dominant eigenvalue is clearly separated in modulus.  Also the right fixed point: residual of the map's eigen-equation with the returned r.
Usage: python profiles/experiments/r05/stress_overlap.py [n_cases] [seed]"""
import sys, json, time
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
import evolve_replay as ER

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
Hm = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
GRID = np.array([0.0, np.pi / 4, -np.pi / 4, np.pi / 2, -np.pi / 2, np.pi])
engines = {}
tot = {'evals': 0, 'status0': 0, 'status1': 0, 'separated': 0, 'max_abs_deta_status0': 0.0, 'max_r_residual_status0': 0.0, 'tied': 0}
bad, t0 = [], time.time()
for case in range(n_cases):
    D = int(rng.choice([2, 4, 8, 16]))
    kind = int(rng.choice([0, 3] if D > 2 else [0, 1, 2, 3]))
    depth = int(rng.integers(1, 5))
    P = {0: 2 * depth, 1: 2 * depth, 2: 15, 3: 3 * depth}[kind]
    B = int(rng.integers(1, 12)) if D == 16 else int(rng.integers(1, 60))
    ref = rng.standard_normal((B, P))
    mode = rng.choice(['near', 'far', 'grid'])
    if mode == 'near':
        cand = ref + 10.0 ** rng.uniform(-6, -1) * rng.standard_normal((B, P))
    elif mode == 'far':
        cand = rng.standard_normal((B, P))
    else:
        cand = GRID[rng.integers(0, len(GRID), size=(B, P))] + (1e-9 * rng.standard_normal((B, P)) if rng.random() < 0.3 else 0.0)
        if rng.random() < 0.5:
            ref = GRID[rng.integers(0, len(GRID), size=(B, P))].astype(float)
    WW = np.eye(4, dtype=complex) if rng.random() < 0.3 else expm(-1j * float(rng.choice([0.02, 0.05, 0.1, 0.3])) * Hm)
    A = np.stack([ER.tensor(kind, D, p) for p in ref])
    if D not in engines:
        engines[D] = EnergyEngine(D, 1024)
    try:
        eta, rounds, st, r = engines[D].overlaps(A, cand, WW, kind='params', ansatz=kind, tol=1e-12, want_r=True,
                                                 max_rounds=(40 if D in (2, 4) else 20000))
    except Exception as e:
        bad.append({'case': case, 'D': D, 'kind': kind, 'error': str(e)[:200]})
        continue
    for b in range(B):
        Bt = ER.tensor(kind, D, cand[b])
        C = np.tensordot(WW, O.merge(A[b], A[b]), [1, 0])
        Bm = O.merge(Bt, Bt)
        w = np.linalg.eigvals(O.transfer_matrix(C, Bm))
        w = w[np.argsort(-np.abs(w))]
        sep = 1.0 - abs(w[1]) / abs(w[0]) if abs(w[0]) > 0 else 0.0
        tot['evals'] += 1
        tot['status0'] += int(st[b] == 0)
        tot['status1'] += int(st[b] not in (0, 4))
        tot['status_tied'] = tot.get('status_tied', 0) + int(st[b] == 4)      # QMPS_STATUS_TIED (ABI 6.4): |eta| valid, no fixed point
        if sep < 1e-7:
            tot['tied'] += 1
            if st[b] in (0, 4) and abs(abs(eta[b]) - abs(w[0])) > 1e-8:
                bad.append({'case': case, 'D': D, 'b': b, 'what': 'tied moduli, status 0, |eta| is not the dominant modulus', 'eta': [eta[b].real, eta[b].imag], 'top': np.abs(w[:3]).tolist()})
            continue
        tot['separated'] += 1
        if st[b] == 4 and abs(abs(eta[b]) - abs(w[0])) > 1e-8 * max(1.0, 1e-7 / sep):
            bad.append({'case': case, 'D': D, 'b': b, 'what': 'status 4 (tie) on a separated spectrum with a wrong modulus', 'sep': float(sep), 'eta': [eta[b].real, eta[b].imag], 'top': np.abs(w[:3]).tolist()})
        if st[b] == 0:
            d = abs(eta[b] - w[0])
            tot['max_abs_deta_status0'] = max(tot['max_abs_deta_status0'], float(d))
            x = r[b]
            Tx = np.einsum('sij,jk,slk->il', C, x, Bm.conj())
            res = float(np.abs(Tx - eta[b] * x).max())
            tot['max_r_residual_status0'] = max(tot['max_r_residual_status0'], res)
            if not d < 1e-8 * max(1.0, 1.0 / sep * 1e-3) or not res < 1e-8:
                bad.append({'case': case, 'D': D, 'kind': kind, 'mode': str(mode), 'b': b, 'what': 'status 0, wrong eigenvalue or fixed point', 'deta': float(d), 'r_residual': res, 'sep': float(sep),
                            'eta': [eta[b].real, eta[b].imag], 'dominant': [w[0].real, w[0].imag], 'rounds': int(rounds[b])})
        elif sep > 1e-4:
            bad.append({'case': case, 'D': D, 'kind': kind, 'mode': str(mode), 'b': b, 'what': f'status {int(st[b])} although the dominant eigenvalue is separated', 'sep': float(sep), 'rounds': int(rounds[b]),
                        'top': np.abs(w[:3]).tolist(), 'ref': ref[b].tolist(), 'cand': cand[b].tolist()})
print(json.dumps({'cases': n_cases, 'seed': seed, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:10]}))
