"""time per evaluation pass of the device-resident D = 2 optimiser: n_steps passes with maxiter = 0 (one pass + one reference tensor per step);
QMPS_EVOLVE_PROBE (debug build): 1 = without the eigen-solve, 2 = without the circuits, 3 = neither"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import expm
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
kind, P = (2, 15) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
WW = expm(-1j * 0.05 * O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
for T in (256, 4096):
    X0 = np.random.default_rng(1).standard_normal((T, P))
    eng = EnergyEngine(2, 64)
    eng.evolve_bfgs_device(kind, X0, WW, n_steps=10, maxiter=0)
    r = eng.evolve_bfgs_device(kind, X0, WW, n_steps=400, maxiter=0, tol=1e-12)
    print('probe', os.environ.get('QMPS_EVOLVE_PROBE'), 'T', T, 'kind', kind, 'us per pass %.2f' % (r['kernel_ms'] / 400 * 1e3), 'squarings per evaluation %.2f' % (r['squarings'] / max(1, r['nfev'])))
    r = eng.evolve_bfgs_device(kind, X0, WW, n_steps=20, maxiter=30, tol=1e-12)
    print('   full run: us per (step) %.1f, iterations mean %.1f max %d, passes per trajectory-step %.1f' % (r['kernel_ms'] / 20 * 1e3, r['nit'].mean(), r['nit'].max(), r['nfev'] / (T * 20) / (2 * P + 1 + 7)))
    eng.close()
