"""Probe (GPU box): D = 4 trajectories STARTED at tied points of the special grid - do they leave?  Per start: nit, objective at the start / the end of the
first step on the device; beside it the central-difference gradient of the Gelfand objective on the CPU (is the point stationary?) and what scipy's BFGS
reaches from there on the same objective."""
import os, sys, json
R = os.environ.get('GRAFT_REPO_ROOT', os.path.abspath(os.path.join(os.path.dirname(__file__), '../../..')))
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import numpy as np
from scipy.linalg import expm
from scipy.optimize import minimize
import evolve_replay as ER
from oracle import qmps_oracle as O
from qmps_amd import _lib as L
from qmps_amd.engine import EnergyEngine
H = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
grid = np.array([[2, -4, 0, 4], [4, 2, -4, 2], [2, 4, 0, -2], [-2, -4, 0, 2], [-4, 0, 2, 2]]) * (np.pi / 4)
for dt in (0.05, 0.3):
    WW = expm(-1j * dt * H)
    eng = EnergyEngine(4, max_batch=4096)
    res = eng.evolve_bfgs_device(L.ANSATZ_SHALLOW_CNOT, grid, WW, n_steps=1, maxiter=30, tol=1e-13)
    for t, x0 in enumerate(grid):
        A = ER.tensor(0, 4, x0)
        f = lambda x: ER.objective_gelfand(0, 4, A, x, WW)
        g = np.array([(f(x0 + 1e-6 * e) - f(x0 - 1e-6 * e)) / 2e-6 for e in np.eye(4)])
        sp = minimize(f, x0, method='BFGS', options={'maxiter': 30})
        print(json.dumps({'dt': dt, 'x0': (x0 / (np.pi / 4)).tolist(), 'nit': int(res['nit'][0, t]), 'f_start': res['fun_start'][0, t], 'f_end': res['fun'][0, t],
                          'f_end_gelfand': f(res['params_hist'][0, t]), 'cpu_grad_max': float(np.abs(g).max()), 'scipy_fun': float(sp.fun), 'scipy_nit': int(sp.nit)}), flush=True)
