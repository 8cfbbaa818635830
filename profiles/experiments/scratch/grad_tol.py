"""how accurate are objective and gradient of qmps_overlap_gradient as a function of the tolerance of its two eigen-solves?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import expm
from qmps_amd import EnergyEngine, _lib as L
from oracle import qmps_oracle as O
D, P, T = 16, 8, 256
rng = np.random.default_rng(1)
X = rng.standard_normal((T, P))
WW = expm(-0.05j * O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0}))
eng = EnergyEngine(D, T * (2 * P + 1))
eng.overlap_set_refs_params(L.ANSATZ_SHALLOW_CNOT, X, WW)
Z = X + 0.02 * rng.standard_normal((T, P))          # iterates near their references, as inside a time step
f0, g0, st = eng.overlap_gradient(L.ANSATZ_SHALLOW_CNOT, Z, tol=1e-14, max_rounds=100000)
for tol in (1e-13, 1e-12, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
    eng.overlap_stats(reset=True)
    f, g, st = eng.overlap_gradient(L.ANSATZ_SHALLOW_CNOT, Z, tol=tol, max_rounds=100000)
    f2, g2, st2 = eng.overlap_gradient(L.ANSATZ_SHALLOW_CNOT, Z, tol=tol, max_rounds=100000, two_sided_f=True)
    s = eng.overlap_stats()
    print('tol %.0e: mean rounds %.1f max %d | max |f - f0| %.2e | max |g - g0| %.2e (|g| ~ %.2e)' % (tol, s['rounds_sum'] / max(s['evaluations'], 1), s['rounds_max'], np.abs(f - f0).max(), np.abs(g - g0).max(), np.abs(g0).max()), '| two-sided f: %.2e' % np.abs(f2 - f0).max())
