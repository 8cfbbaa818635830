"""device-resident D = 2 BFGS against the host driver, bit level: where do the two start to differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import expm
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
kind, P, T = 0, 8, 7
rng = np.random.default_rng(900 + kind + P)
X0 = rng.standard_normal((T, P))
WW = expm(-1j * 0.05 * O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
eng = EnergyEngine(2, T * (2 * P + 1))
for carry in (False, True):
    for it in (0, 1, 2, 3, 5, 40):
        h = eng.evolve_bfgs(kind, X0, WW, n_steps=1, maxiter=it, tol=1e-13, carry_hessian=carry)
        d = eng.evolve_bfgs_device(kind, X0, WW, n_steps=1, maxiter=it, tol=1e-13, carry_hessian=carry)
        print('carry', carry, 'maxiter', it, 'fun_start equal', np.array_equal(h['fun_start'], d['fun_start']), 'max |dx|', np.abs(h['x'] - d['x']).max(),
              'max |df|', np.abs(h['fun'] - d['fun']).max(), 'nit host', h['nit'], 'dev', d['nit'][0], 'Hinv diff', np.abs(h['hess_inv'] - d['hess_inv']).max())
