"""Round 5, second session: the per-trajectory device-resident optimiser at D = 16 (qmps_evolve_d16.hip) against the lock-step driver
on config 4's inputs (TFIM quench, depth 4, 8 angles).  Prints agreement of the per-step minima and the time per time step of both.
Usage: python d16_trajectory_probe.py [T] [n_steps] [carry 0|1]"""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from qmps_amd import new_time_evolve as NT, represent as R
from qmps_amd.ground_state import Hamiltonian
from scipy.linalg import expm


def WW_of(dt):
    return expm(-1j * dt * Hamiltonian({'ZZ': -1.0, 'X': 1.0}).to_matrix())


T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
carry = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
D, P = 16, 8
rng = np.random.default_rng(2560)
X0 = rng.standard_normal((T, P))
WW = WW_of(0.05)
out = {}
for name, dd in (('lockstep', 'lockstep'), ('trajectory', 'trajectory')):
    ev = NT.LockstepEvolver(D, T, P, cls=R.ShallowCNOTStateTensor, tol=1e-12, maxiter=30, carry_hessian=carry, speculative=True, device_driver=dd)
    X = X0.copy()
    r = ev.steps(X, WW, 2, counters=False)          # warm-up (first steps are far from the steady state)
    X = r['x']
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        r = ev.steps(X, WW, n_steps, counters=False)
        ts.append((time.perf_counter() - t0) / n_steps * 1e3)
        X = r['x']
        if rep == 0:
            first = r
    rc = ev.steps(X, WW, 2, counters=True)
    out[name] = dict(ms_per_step=ts, fun=first['fun'], x=first['x'], nit=np.asarray(first['nit']), counters={k: rc[k] for k in rc if k in ('nfev', 'failed_evaluations', 'kernel_ms', 'squarings', 'gradient_batches', 'ladder_batches')})
    ev.close()
a, b = out['lockstep'], out['trajectory']
rep = {'T': T, 'n_steps': n_steps, 'carry': carry,
       'ms_per_step_lockstep': a['ms_per_step'], 'ms_per_step_trajectory': b['ms_per_step'],
       'max_abs_df_end': float(np.abs(a['fun'] - b['fun']).max()), 'mean_f_lockstep': float(a['fun'][-1].mean()), 'mean_f_trajectory': float(b['fun'][-1].mean()),
       'max_abs_dx': float(np.abs(a['x'] - b['x']).max()),
       'nit_lockstep_mean': float(np.mean(a['nit'])), 'nit_trajectory_mean': float(np.mean(b['nit'])), 'nit_trajectory_max': int(np.max(b['nit'])),
       'counters_lockstep': a['counters'], 'counters_trajectory': b['counters']}
print(json.dumps(rep))
if len(sys.argv) > 4:
    np.savez(sys.argv[4], x_start=X0, x_lockstep=a['x'], x_trajectory=b['x'], WW=WW)
