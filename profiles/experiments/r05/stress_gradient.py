"""Round 5: randomised stress of the TWO-SIDED gradient (qmps_overlap_gradient: objective and central-difference gradient of every iterate from one right
and one left eigen-solve, eta' = <y, T'(r)>/<y, r>) against the oracle's central differences of dense eigen-solves, D = 4, 8, 16: iterates near the
reference state (the BFGS regime), at moderate distance and far from it, time steps 0.02 ... 0.3.  A silent error here = a gradient that is wrong although
both solves report status 0 - e.g. the right and the left solve settling on DIFFERENT members of a nearly tied pair.
Usage: python profiles/experiments/r05/stress_gradient.py [n_cases] [seed]"""
import sys, json, time
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
import evolve_replay as ER

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
Hm = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
engines = {}
tot = {'iterates': 0, 'status0': 0, 'checked': 0, 'max_df': 0.0, 'max_dg': 0.0, 'skipped_small_gap': 0}
bad, t0 = [], time.time()
h = 1e-6
for case in range(n_cases):
    D = int(rng.choice([4, 8, 16]))
    kind = int(rng.choice([0, 3]))
    depth = int(np.log2(D)) if rng.random() < 0.7 else int(rng.integers(1, 4))
    P = (3 if kind == 3 else 2) * depth
    T = 3 if D == 16 else 6
    ref = rng.standard_normal((T, P))
    dist = float(10.0 ** rng.uniform(-4, 0.3))
    X = ref + dist * rng.standard_normal((T, P)) / np.sqrt(P)
    dt = float(rng.choice([0.02, 0.05, 0.1, 0.3]))
    WW = expm(-1j * dt * Hm)
    if D not in engines:
        engines[D] = EnergyEngine(D, 1024)
    eng = engines[D]
    eng.overlap_set_refs_params(kind, ref, WW)
    f, g, st = eng.overlap_gradient(kind, X, h=h, tol=1e-13)
    for t in range(T):
        tot['iterates'] += 1
        tot['status0'] += int(st[t] == 0)
        A = ER.tensor(kind, D, ref[t])
        f0, gap = ER.objective(kind, D, A, X[t], WW, want_gap=True)
        if st[t] != 0:
            if gap < 1 - 1e-3:
                bad.append({'case': case, 'D': D, 't': t, 'what': f'status {int(st[t])} although the dominant eigenvalue is separated', 'ratio': float(gap), 'dist': dist})
            continue
        if gap > 1 - 3e-3:
            tot['skipped_small_gap'] += 1       # (the objective itself is not smooth across a crossing of moduli)
            continue
        go = np.empty(P)
        for k in range(P):
            e = np.zeros(P); e[k] = h
            go[k] = (ER.objective(kind, D, A, X[t] + e, WW) - ER.objective(kind, D, A, X[t] - e, WW)) / (2 * h)
        tot['checked'] += 1
        df, dg = abs(f[t] - f0), float(np.abs(g[t] - go).max())
        tot['max_df'] = max(tot['max_df'], float(df))
        tot['max_dg'] = max(tot['max_dg'], dg)
        # (the oracle's own central difference of dense eigen-solves carries ~1e-16 / h * cond ~ 1e-8 .. 1e-7 of noise)
        if not df < 1e-9 or not dg < 2e-6 * max(1.0, float(np.abs(go).max())):
            bad.append({'case': case, 'D': D, 'kind': kind, 'P': P, 't': t, 'dist': dist, 'dt': dt, 'ratio': float(gap), 'df': float(df), 'dg': dg, 'g_dev': g[t].tolist(), 'g_oracle': go.tolist()})
print(json.dumps({'cases': n_cases, 'seed': seed, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:6]}))
