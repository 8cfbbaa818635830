"""Round 5: randomised stress of the per-trajectory device-resident optimisers at D = 2 and 4 (qmps_evolve_bfgs_device: the whole BFGS time evolution in
one launch) against the host-loop driver qmps_evolve_bfgs on the same inputs: random ansatz kinds, sizes, time steps, carried or fresh inverse
Hessians, near and far starts.  The two drivers evaluate the same formulae in different kernels (rounding-level differences, not bit-identity): per time
step the minima must agree to 1e-6 (a far start may legitimately end in another local minimum: counted, listed), nothing may be NaN unless the host
loop's is, no evaluation may fail where the host loop's do not.
Usage: python profiles/experiments/r05/stress_evolve_device.py [n_cases] [seed]"""
import sys, json, time
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import evolve_replay as ER
from qmps_amd import EnergyEngine
from qmps_amd.ground_state import Hamiltonian

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
H = Hamiltonian({'ZZ': -1.0, 'X': 1.0}).to_matrix()
engines = {}
tot = {'oracle_checked': 0, 'max_d_oracle_dev': 0.0, 'max_d_oracle_host': 0.0, 'cases': 0, 'trajectory_steps': 0, 'max_df_near': 0.0, 'far_other_minimum': 0, 'device_failed_evaluations': 0, 'nan_device_only': 0}
bad, t0 = [], time.time()
for case in range(n_cases):
    D = int(rng.choice([2, 4]))
    kind = int(rng.choice([0, 1, 2, 3, 6] if D == 2 else [0, 1, 3]))
    depth = int(rng.integers(1, 4))
    P = {0: 2 * depth, 1: 2 * depth, 2: 15, 3: 3 * depth, 6: 8}[kind]
    T = int(rng.integers(1, 48))
    carry, far = bool(rng.integers(2)), rng.random() < 0.3
    dt = float(rng.choice([0.02, 0.05, 0.1]))
    WW = expm(-1j * dt * H)
    X0 = rng.standard_normal((T, P)) if not far else 2.0 * rng.standard_normal((T, P))
    n_steps = 3
    key = D
    if key not in engines:
        engines[key] = EnergyEngine(D, 4096)
    eng = engines[key]
    try:
        host = eng.evolve_bfgs(kind, X0, WW, n_steps=n_steps, maxiter=60, tol=1e-13, carry_hessian=carry, counters=False)
        dev = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=60, tol=1e-13, carry_hessian=carry)
    except Exception as e:
        bad.append({'case': case, 'D': D, 'kind': kind, 'P': P, 'T': T, 'error': str(e)[:200]})
        continue
    tot['cases'] += 1
    tot['trajectory_steps'] += T * n_steps
    tot['device_failed_evaluations'] += int(dev['failed_evaluations'])
    fd, fh = np.asarray(dev['fun']), np.asarray(host['fun'])
    nan_dev_only = np.isnan(fd) & ~np.isnan(fh)
    tot['nan_device_only'] += int(nan_dev_only.sum())
    d = np.abs(fd - fh)
    d = np.where(np.isnan(d), 0.0, d)
    # the first time step from the common start is the cleanest comparison; later steps start from each driver's own parameters
    if not far:
        tot['max_df_near'] = max(tot['max_df_near'], float(d.max()))
    other = int((d > 1e-6).sum())
    if other and far:
        tot['far_other_minimum'] += other
    # the recorded objectives are the ORACLE's at each driver's own parameters (what matters: an evaluation that is silently wrong)
    for name, res in (('dev', dev), ('host', host)):
        ph, fun = res['params_hist'], np.asarray(res['fun'])
        for t in rng.choice(T, size=min(T, 4), replace=False):
            prev = X0[t]
            for st_ in range(n_steps):
                if not np.isnan(fun[st_, t]):
                    f_or = ER.objective(kind, D, ER.tensor(kind, D, prev), ph[st_][t], WW)
                    dd = abs(f_or - fun[st_, t])
                    tot['oracle_checked'] += 1
                    tot['max_d_oracle_' + name] = max(tot['max_d_oracle_' + name], float(dd))
                    if dd > 1e-8:
                        bad.insert(0, {'case': case, 'D': D, 'kind': kind, 'what': name + ': recorded objective is not the oracle\'s at its parameters', 'step': st_, 't': int(t), 'recorded': float(fun[st_, t]), 'oracle': float(f_or)})
                prev = ph[st_][t]
    if nan_dev_only.any():
        s_, t_ = np.argwhere(nan_dev_only)[0]
        bad.insert(0, {'case': case, 'D': D, 'kind': kind, 'P': P, 'T': T, 'carry': carry, 'far': far, 'dt': dt, 'what': 'NaN on the device only', 'at': [int(s_), int(t_)], 'f_host': float(fh[s_, t_]),
                       'x_prev': (X0[t_] if s_ == 0 else dev['params_hist'][s_ - 1][t_]).tolist(), 'x0': X0[t_].tolist()})
    if False:
        s, t = np.unravel_index(np.argmax(d), d.shape)
        bad.append({'case': case, 'D': D, 'kind': kind, 'P': P, 'T': T, 'carry': carry, 'far': far, 'dt': dt, 'max_df': float(d.max()), 'at': [int(s), int(t)],
                    'f_dev': float(fd[s, t]), 'f_host': float(fh[s, t]), 'nan_device_only': int(nan_dev_only.sum()), 'x0': X0[t].tolist()})
print(json.dumps({'seed': seed, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:8]}))
