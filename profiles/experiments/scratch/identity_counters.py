"""Batch counts of the identity-start (scipy-like) time step at D = 16, 256 trajectories: gradient batches, ladder batches, time."""
import os, sys, time
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from qmps_amd.new_time_evolve import LockstepEvolver
from qmps_amd.represent import ShallowCNOTStateTensor
T, P, D = 256, 8, 16
WW = expm(-1j * 0.05 * bench.tfim_h(1.0))
X = np.random.default_rng(20241022).standard_normal((T, P))
for carry in (True, False):
    ev = LockstepEvolver(D, T, P, ShallowCNOTStateTensor, tol=1e-12, maxiter=30, carry_hessian=carry, speculative=True)
    Y = ev.steps(X, WW, 3)['x']
    t0 = time.perf_counter()
    r = ev.steps(Y, WW, 6)
    dt = time.perf_counter() - t0
    st = ev.fg.eng.overlap_stats()
    print('carry' if carry else 'identity', 'ms/step %.3f' % (dt / 6 * 1e3), 'nit', list(r['nit']), 'gradient batches/step %.1f ladder batches/step %.1f gradient ms/step %.3f' %
          (r['gradient_batches'] / 6, r['ladder_batches'] / 6, r['gradient_ms'] / 6), 'rounds mean %.1f max %d' % (st['rounds_sum'] / max(1, st['evaluations']), st['rounds_max']))
    ev.close()
