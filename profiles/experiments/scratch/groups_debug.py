import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qmps_amd import _lib
from qmps_amd.engine import EnergyEngine as Engine
from scipy.linalg import expm
import bench
D, P, T, K = [int(a) for a in sys.argv[1:5]]
rng = np.random.default_rng(77 + D)
WW = expm(-1j * 0.05 * bench.tfim_h(1.0))
X0 = rng.standard_normal((T, P))
kind = _lib.ANSATZ_SHALLOW_CNOT
one, many = Engine(D, T * (2 * P + 1)), Engine(D, T * (2 * P + 1))
one.set_evolve_groups(1); many.set_evolve_groups(K)
kw = dict(n_steps=3, maxiter=30, tol=1e-13, carry_hessian=True)
a = one.evolve_bfgs(kind, X0, WW, **kw); b = many.evolve_bfgs(kind, X0, WW, **kw)
print('call 1 equal', np.array_equal(a['params_hist'], b['params_hist']), a['nit'], b['nit'])
for n in (1, 1, 1):
    a2 = one.evolve_bfgs(kind, a['x'], WW, n_steps=n, maxiter=30, tol=1e-13, carry_hessian=True, warm=True, hess_inv=a['hess_inv'])
    b2 = many.evolve_bfgs(kind, b['x'], WW, n_steps=n, maxiter=30, tol=1e-13, carry_hessian=True, warm=True, hess_inv=b['hess_inv'])
    print('nit', a2['nit'], b2['nit'], 'fun_start diff per traj', np.abs(a2['fun_start'] - b2['fun_start']).max(axis=0), 'x diff per traj', np.abs(a2['x'] - b2['x']).max(axis=1),
          'batches', a2['gradient_batches'], b2['gradient_batches'], 'ladder', a2['ladder_batches'], b2['ladder_batches'])
    a, b = a2, b2
