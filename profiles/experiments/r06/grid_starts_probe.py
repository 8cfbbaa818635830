"""Probe (GPU box): the device-resident BFGS drivers started ON the special grid (multiples of pi / 4, pi / 2), every recorded objective - the start of
each step and its end - against Gelfand's formula (tests/evolve_replay.spectral_radius).  D = 2 (characteristic-polynomial solve + squaring hand-back)
and D = 4 (squaring solve on the matrix cores).  Prints the worst deviation, the failed evaluations and the non-finite objectives per case."""
import os, sys, json
R = os.environ.get('GRAFT_REPO_ROOT', os.path.abspath(os.path.join(os.path.dirname(__file__), '../../..')))
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import numpy as np
from scipy.linalg import expm
import evolve_replay as ER
from oracle import qmps_oracle as O
from qmps_amd import _lib as L
from qmps_amd.engine import EnergyEngine

H = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
out = []
CASES = ((2, L.ANSATZ_SHALLOW_CNOT, 2), (2, L.ANSATZ_SHALLOW_CNOT, 8), (2, L.ANSATZ_SHALLOW_FULL, 15), (4, L.ANSATZ_SHALLOW_CNOT, 4), (4, L.ANSATZ_SHALLOW_CNOT, 8))
if len(sys.argv) > 2:
    CASES = tuple(c for c in CASES if c[0] == int(sys.argv[2]) and c[2] == int(sys.argv[3]))
for D, kind, P in CASES:
    for dt in (0.0, 0.05, 0.3):
        WW = expm(-1j * dt * H)
        X0 = np.concatenate([rng.integers(-4, 5, (150, P)) * (np.pi / 4), rng.integers(-2, 3, (150, P)) * (np.pi / 2)])
        T, n_steps = len(X0), 2
        eng = EnergyEngine(D, max_batch=max(4096, T * (2 * P + 1 + 8)))
        res = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=4, tol=1e-13)
        prev = X0
        worst = worst0 = 0.0
        nan = 0
        bad = []
        for step in range(n_steps):
            for t in range(T):
                A = ER.tensor(kind, D, prev[t])
                f0 = ER.objective_gelfand(kind, D, A, prev[t], WW)
                f1 = ER.objective_gelfand(kind, D, A, res['params_hist'][step, t], WW)
                g0, g1 = res['fun_start'][step, t], res['fun'][step, t]
                if not (np.isfinite(g0) and np.isfinite(g1)):
                    nan += 1
                    continue
                worst0 = max(worst0, abs(f0 - g0)); worst = max(worst, abs(f1 - g1))
                if max(abs(f0 - g0), abs(f1 - g1)) > 1e-8 and len(bad) < 3:
                    bad.append({'t': t, 'step': step, 'x0': (prev[t] / (np.pi / 4)).round(3).tolist(), 'x0_exact': [float.hex(float(v)) for v in prev[t]], 'f0': f0, 'g0': g0, 'f1': f1, 'g1': g1})
            prev = res['params_hist'][step]
        rec = {'D': D, 'kind': int(kind), 'P': P, 'dt': dt, 'worst_start': worst0, 'worst_end': worst, 'nonfinite': nan, 'failed_evaluations': res['failed_evaluations'], 'bad': bad}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del eng
json.dump(out, open(os.path.join(R, 'gpurun_out', 'grid_starts_probe.json'), 'w'), indent=1)
