"""Probe (GPU box): the device-resident rotosolve time evolution (qmps_evolve_rotosolve) started ON the special grid, D = 2 and 4, single and double
frequency: the objective recorded after the last sweep of each step against Gelfand's formula at the recorded parameters; NaN counted."""
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
for D, kind, P in ((2, L.ANSATZ_SHALLOW_CNOT, 2), (2, L.ANSATZ_SHALLOW_CNOT, 8), (2, L.ANSATZ_SHALLOW_FULL, 15), (4, L.ANSATZ_SHALLOW_CNOT, 4), (4, L.ANSATZ_SHALLOW_CNOT, 8)):
    for dt in (0.0, 0.05, 0.3):
        WW = expm(-1j * dt * H)
        X0 = np.concatenate([rng.integers(-4, 5, (100, P)) * (np.pi / 4), rng.integers(-2, 3, (100, P)) * (np.pi / 2)])
        T = len(X0)
        for dbl in (False, True):
            eng = EnergyEngine(D, max_batch=max(4096, 8 * T))
            x, ph, fh = eng.evolve_rotosolve(kind, X0, WW, n_steps=2, n_sweeps=2, double_frequency=dbl, tol=1e-13)
            prev, worst, nan, bad = X0, 0.0, 0, []
            for step in range(2):
                for t in range(T):
                    g = fh[step, -1, t]
                    if not np.isfinite(g):
                        nan += 1
                        continue
                    f = ER.objective_gelfand(kind, D, ER.tensor(kind, D, prev[t]), ph[step, t], WW)
                    worst = max(worst, abs(f - g))
                    if abs(f - g) > 1e-8 and len(bad) < 2:
                        bad.append({'t': t, 'step': step, 'x0': (prev[t] / (np.pi / 4)).round(3).tolist(), 'f': f, 'g': g})
                prev = ph[step]
            print(json.dumps({'D': D, 'P': P, 'dt': dt, 'double': dbl, 'worst': worst, 'nan': nan, 'bad': bad}), flush=True)
            del eng
