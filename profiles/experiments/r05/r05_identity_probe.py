import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from scipy.linalg import expm
from qmps_amd import EnergyEngine
import bench
T, P, D = 256, 8, 16
WW = expm(-0.05j * bench.tfim_h(1.0))
X0 = np.random.default_rng(20241022).standard_normal((T, P))
for env in (None, 'QMPS_EVOLVE_HOST_ALGEBRA'):
    if env: os.environ[env] = '1'
    with EnergyEngine(D, T * (2 * P + 1)) as eng:
        a = eng.evolve_bfgs(0, X0, WW, n_steps=3, maxiter=30, tol=1e-12, carry_hessian=False, counters=True)
        for cnt in (True, False):
            t0 = time.perf_counter()
            b = eng.evolve_bfgs(0, a['x'], WW, n_steps=6, maxiter=30, tol=1e-12, carry_hessian=False, warm=True, counters=cnt)
            dt = time.perf_counter() - t0
            print(env, 'counters', cnt, 'identity start: ms/step %.3f' % (dt / 6 * 1e3), 'nit', b['nit'], 'grad batches', b['gradient_batches'], 'ladder batches', b['ladder_batches'], 'grad_ms %.2f' % b['gradient_ms'])
    if env: del os.environ[env]
