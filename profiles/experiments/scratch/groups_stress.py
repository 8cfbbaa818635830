"""Stress: the grouped qmps_evolve_bfgs (host threads inside the call) against the one lock-step, many calls, bit for bit."""
import os, sys
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from qmps_amd import EnergyEngine, _lib
WW = expm(-1j * 0.05 * bench.tfim_h(1.0))
bad = 0
SC = int(sys.argv[1]) if len(sys.argv) > 1 else 1
worst = [0.0, 0.0]
for D, P, T, K, calls in ((8, 6, 600, 4, 25 * SC), (16, 8, 520, 3, 15 * SC), (16, 8, 1024, 4, 10 * SC), (4, 4, 700, 5, 10 * SC)):
    bad = 0
    X0 = np.random.default_rng(D).standard_normal((T, P))
    with EnergyEngine(D, T * (2 * P + 1)) as many, EnergyEngine(D, T * (2 * P + 1)) as one:
        many.set_evolve_groups(K)
        one.set_evolve_groups(1)
        a = many.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, n_steps=2, maxiter=30, carry_hessian=True)
        b = one.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, n_steps=2, maxiter=30, carry_hessian=True)
        for c in range(calls):
            a = many.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, a['x'], WW, n_steps=2, maxiter=30, carry_hessian=True, warm=True, hess_inv=a['hess_inv'])
            b = one.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, b['x'], WW, n_steps=2, maxiter=30, carry_hessian=True, warm=True, hess_inv=b['hess_inv'])
            same = np.array_equal(a['x'], b['x']) and np.array_equal(a['fun'], b['fun']) and np.array_equal(a['nit'], b['nit']) and np.array_equal(a['hess_inv'], b['hess_inv'])
            bad += 0 if same else 1
            worst[0] = max(worst[0], np.abs(a['fun'] - b['fun']).max()); worst[1] = max(worst[1], np.abs(a['x'] - b['x']).max())
        print('D %d T %d groups %d: %d calls, mismatches so far %d, final objective %.6f' % (D, T, K, calls, bad, a['fun'][-1].mean()))
print('STRESS', 'OK' if bad == 0 else 'FAILED', 'worst |dfun| %.2e |dx| %.2e' % tuple(worst))
