import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from scipy.linalg import expm
from qmps_amd import EnergyEngine
import bench
P, D = 8, 16
WW = expm(-0.05j * bench.tfim_h(1.0))
for T in (256, 1024):
    X0 = np.random.default_rng(20241022).standard_normal((T, P))
    for K in (1, 2, 4):
        with EnergyEngine(D, T * (2 * P + 1)) as eng:
            eng.set_evolve_groups(K)
            a = eng.evolve_bfgs(0, X0, WW, n_steps=3, maxiter=30, tol=1e-12, carry_hessian=True, counters=False)
            ts = []
            for rep in range(3):
                t0 = time.perf_counter()
                a = eng.evolve_bfgs(0, a['x'], WW, n_steps=10, maxiter=30, tol=1e-12, carry_hessian=True, hess_inv=a['hess_inv'], warm=True, counters=False)
                ts.append((time.perf_counter() - t0) / 10 * 1e3)
            print('T', T, 'groups', K, 'carried: ms/step', ['%.3f' % t for t in ts], 'nit', a['nit'])
