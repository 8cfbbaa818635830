"""Round 5: randomised stress of the evolve drivers (not part of the test suite: minutes of GPU time).  Per case a random size, carried or fresh
inverse Hessians, fixed or adaptive tolerance, a near or a far (rejections, ladders) second start:
  * D = 8, 16: qmps_evolve_bfgs with the algebra on the device (blind chains) against the round-4 host loop (QMPS_EVOLVE_HOST_ALGEBRA) - every
    number must be IDENTICAL (iteration counts, objectives, parameters, inverse Hessians);
  * D = 16: qmps_evolve_bfgs_device (a workgroup per trajectory) against the lock-step - same minima to 1e-7, no failed evaluation.
Usage: python profiles/experiments/r05/stress_evolve.py [n_cases] [seed]"""
import os, sys, json, time
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, '.')
from qmps_amd import EnergyEngine
from qmps_amd.ground_state import Hamiltonian

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
H = Hamiltonian({'ZZ': -1.0, 'X': 1.0}).to_matrix()
engines = {}


def engine(D, n):
    need = 1 << max(10, int(np.ceil(np.log2(n))))
    key = (D, need)
    if key not in engines:
        engines[key] = EnergyEngine(D, need)
    return engines[key]


bad, worst_traj, t0 = [], 0.0, time.time()
for case in range(n_cases):
    D = int(rng.choice([8, 16, 16]))
    P = 6 if D == 8 else 8
    T = int(rng.integers(1, 70)) if rng.random() < 0.8 else int(rng.integers(200, 520))
    carry, adaptive, far = bool(rng.integers(2)), bool(rng.integers(2)), rng.random() < 0.4
    dt = float(rng.choice([0.02, 0.05, 0.1]))
    WW = expm(-1j * dt * H)
    X0 = rng.standard_normal((T, P))
    kick = 0.3 * rng.standard_normal((T, P)) if far else 0.0
    eng = engine(D, T * (2 * P + 1))
    out = {}
    for name in ('device', 'host'):
        if name == 'host':
            os.environ['QMPS_EVOLVE_HOST_ALGEBRA'] = '1'
        else:
            os.environ.pop('QMPS_EVOLVE_HOST_ALGEBRA', None)
        a = eng.evolve_bfgs(0, X0, WW, n_steps=1, maxiter=30, tol=1e-12, carry_hessian=carry, counters=False, adaptive_gradient=adaptive)
        b = eng.evolve_bfgs(0, a['x'] + kick, WW, n_steps=3, maxiter=30, tol=1e-12, carry_hessian=carry, hess_inv=a['hess_inv'] if carry else None,
                            warm=not far, counters=False, adaptive_gradient=adaptive)
        out[name] = (a, b)
    os.environ.pop('QMPS_EVOLVE_HOST_ALGEBRA', None)
    same = all(np.array_equal(np.asarray(out['device'][k][f], dtype=float), np.asarray(out['host'][k][f], dtype=float), equal_nan=True)
               for k in (0, 1) for f in ('nit', 'fun', 'fun_start', 'x', 'params_hist', 'hess_inv'))
    n_nan = int(np.isnan(out['device'][1]['fun']).sum())
    rec = {'case': case, 'D': D, 'T': T, 'carry': carry, 'adaptive': adaptive, 'far': far, 'dt': dt, 'identical': bool(same), 'nan_objectives': n_nan}
    if D == 16:
        dv = eng.evolve_bfgs_device(0, out['device'][0]['x'] + kick, WW, n_steps=3, maxiter=30, tol=1e-12, carry_hessian=carry,
                                    hess_inv=out['device'][0]['hess_inv'] if carry else None, adaptive_gradient=adaptive)
        d = float(np.nanmax(np.abs(dv['fun'] - out['device'][1]['fun'])))
        rec.update(trajectory_driver_max_df=d, trajectory_driver_failed=dv['failed_evaluations'])
        worst_traj = max(worst_traj, d)
        if d > 1e-6 or (dv['failed_evaluations'] and not n_nan):
            rec['identical'] = rec['identical'] and False
            rec['trajectory_mismatch'] = True
    if not rec['identical']:
        bad.append(rec)
    print(json.dumps(rec), flush=True)
print(json.dumps({'cases': n_cases, 'seed': seed, 'not_identical_or_mismatch': len(bad), 'worst_trajectory_driver_df': worst_traj, 'seconds': time.time() - t0, 'bad': bad}))
