"""D = 2 device-resident BFGS (qmps_evolve_bfgs_device): the characteristic-polynomial eigenvalue solve (round 6) against the squaring solve
(QMPS_EVOLVE_D2_SQUARING=1) and the oracle (dense eig) - objectives along the trajectories, iteration counts, squarings / Aberth iterations, time."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine, _lib as L
import evolve_replay as ER
from scipy.linalg import expm
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
H = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
out = {}
for kind, P, name in ((L.ANSATZ_SHALLOW_FULL, 15, 'full'), (L.ANSATZ_SHALLOW_CNOT, 8, 'cnot8'), (L.ANSATZ_SHALLOW_CNOT, 2, 'cnot2')):
    T, n_steps = 64, 4
    X0 = rng.standard_normal((T, P))
    X0[::5] = np.round(X0[::5] / (np.pi / 4)) * (np.pi / 4)          # a fifth of the starts on the special grid
    WW = expm(-0.05j * H)
    res = {}
    with EnergyEngine(2, T * (2 * P + 1 + 8)) as eng:
        for mode in ('charpoly', 'squaring'):
            if mode == 'squaring':
                os.environ['QMPS_EVOLVE_D2_SQUARING'] = '1'
            else:
                os.environ.pop('QMPS_EVOLVE_D2_SQUARING', None)
            t0 = time.perf_counter()
            res[mode] = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=60, tol=1e-13)
            res[mode]['wall'] = time.perf_counter() - t0
        os.environ.pop('QMPS_EVOLVE_D2_SQUARING', None)
    a, b = res['charpoly'], res['squaring']
    # oracle objective at the device's parameters, step by step
    worst = 0.0
    prev = X0
    for step in range(n_steps):
        for t in range(0, T, 3):
            f_or = ER.objective(kind, 2, ER.tensor(kind, 2, prev[t]), a['params_hist'][step, t], WW)
            if np.isfinite(a['fun'][step, t]):
                worst = max(worst, abs(f_or - a['fun'][step, t]))
        prev = a['params_hist'][step]
    out[name] = {'max_abs_df_final_vs_squaring': float(np.nanmax(np.abs(a['fun'] - b['fun']))), 'max_abs_f_vs_oracle': worst,
                 'nan_charpoly': int(np.isnan(a['fun']).sum()), 'nan_squaring': int(np.isnan(b['fun']).sum()),
                 'failed_charpoly': a['failed_evaluations'], 'failed_squaring': b['failed_evaluations'],
                 'nit_mean_charpoly': float(a['nit'].mean()), 'nit_mean_squaring': float(b['nit'].mean()),
                 'rounds_per_eval_charpoly': a['squarings'] / max(a['nfev'], 1), 'rounds_per_eval_squaring': b['squarings'] / max(b['nfev'], 1),
                 'kernel_ms_charpoly': a['kernel_ms'], 'kernel_ms_squaring': b['kernel_ms']}
print(json.dumps(out, indent=1))
