"""Replay ONE case of stress_evolve.py (same seed, same random stream) and say which records differ between the device-algebra path and the
host loop, and by how much.  Usage: python profiles/experiments/r05/stress_repro.py <seed> <case> [repeat]"""
import os, sys, json
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, '.')
from qmps_amd import EnergyEngine
from qmps_amd.ground_state import Hamiltonian

seed, target = int(sys.argv[1]), int(sys.argv[2])
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(seed)
H = Hamiltonian({'ZZ': -1.0, 'X': 1.0}).to_matrix()
for case in range(target + 1):
    D = int(rng.choice([8, 16, 16]))
    P = 6 if D == 8 else 8
    T = int(rng.integers(1, 70)) if rng.random() < 0.8 else int(rng.integers(200, 520))
    carry, adaptive, far = bool(rng.integers(2)), bool(rng.integers(2)), rng.random() < 0.4
    dt = float(rng.choice([0.02, 0.05, 0.1]))
    X0 = rng.standard_normal((T, P))
    kick = 0.3 * rng.standard_normal((T, P)) if far else 0.0
print(json.dumps({'case': target, 'D': D, 'T': T, 'carry': carry, 'adaptive': adaptive, 'far': far, 'dt': dt}))
WW = expm(-1j * dt * H)
need = 1 << max(10, int(np.ceil(np.log2(T * (2 * P + 1)))))
for rep in range(repeat):
    eng = EnergyEngine(D, need)
    out = {}
    for name in ('device', 'host', 'device2'):
        if name == 'host':
            os.environ['QMPS_EVOLVE_HOST_ALGEBRA'] = '1'
        else:
            os.environ.pop('QMPS_EVOLVE_HOST_ALGEBRA', None)
        a = eng.evolve_bfgs(0, X0, WW, n_steps=1, maxiter=30, tol=1e-12, carry_hessian=carry, counters=False, adaptive_gradient=adaptive)
        b = eng.evolve_bfgs(0, a['x'] + kick, WW, n_steps=3, maxiter=30, tol=1e-12, carry_hessian=carry, hess_inv=a['hess_inv'] if carry else None,
                            warm=not far, counters=False, adaptive_gradient=adaptive)
        out[name] = (a, b)
    os.environ.pop('QMPS_EVOLVE_HOST_ALGEBRA', None)
    for other in ('host', 'device2'):
        for k in (0, 1):
            for f in ('nit', 'fun', 'fun_start', 'x', 'params_hist', 'hess_inv'):
                u, v = np.asarray(out['device'][k][f], dtype=float), np.asarray(out[other][k][f], dtype=float)
                if not np.array_equal(u, v):
                    d = np.abs(u - v)
                    idx = np.unravel_index(np.nanargmax(d), d.shape)
                    print(f'rep {rep}: device vs {other}, call {k}, {f}: max |diff| {np.nanmax(d):.3e} at {idx}; differing entries {int((u != v).sum())} of {u.size}')
    print(f'rep {rep}: nit device {out["device"][1]["nit"]} host {out["host"][1]["nit"]}')
    eng.close() if hasattr(eng, 'close') else None

# ---- where is the NaN, and what does the map look like there?
from oracle import qmps_oracle as O
b = out['device'][1]
fun, fs = np.asarray(b['fun']), np.asarray(b['fun_start'])
for (s, t) in zip(*np.where(np.isnan(fs) | np.isnan(fun))):
    prev = (out['device'][0]['x'] + kick)[t] if s == 0 else b['params_hist'][s - 1][t]
    A = O.unitary_to_tensor(O.shallow_cnot_unitary(D, prev))
    C = np.tensordot(WW, O.merge(A, A), [1, 0])
    w = np.linalg.eigvals(O.transfer_matrix(C, O.merge(A, A)))
    w = w[np.argsort(-np.abs(w))]
    print(f'NaN at step {s} trajectory {t}: fun_start {fs[s, t]} fun {fun[s, t]}; |eta| of the map at the start: {np.abs(w[:5])}, phases {np.angle(w[:5])}')
    print('   parameters', prev.tolist())
    # the same single trajectory on its own
    e1 = EnergyEngine(D, 1024)
    r1 = e1.evolve_bfgs(0, prev[None], WW, n_steps=1, maxiter=30, tol=1e-12, counters=True, adaptive_gradient=adaptive)
    print('   alone: fun_start', r1['fun_start'], 'fun', r1['fun'], 'nit', r1['nit'])
    eta, rounds, st = e1.overlaps(A[None], prev[None], WW, kind='params', ansatz=0, max_rounds=100000, tol=1e-12) if hasattr(e1, 'overlaps') else (None, None, None)
    print('   overlaps(): eta', eta, 'status', st)
