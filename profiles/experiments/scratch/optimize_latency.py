"""wall time of the reference-API optimiser run: SparseFullEnergyOptimizer(...).optimize() with method 'Rotosolve' (= one C call)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from qmps_amd.ground_state import Hamiltonian, SparseFullEnergyOptimizer
H = Hamiltonian({'ZZ': -1, 'X': 1}).to_matrix()
for D, depth in ((2, 2), (4, 2), (8, 3)):
    rng = np.random.default_rng(D)
    for maxiter in (10, 100):
        opt = SparseFullEnergyOptimizer(H, D=D, depth=depth, initial_guess=rng.standard_normal(2 * depth), settings={'verbose': False})
        opt.change_settings({'method': 'Rotosolve', 'maxiter': maxiter})
        opt.optimize()                      # warm-up (graph capture, allocations)
        opt = SparseFullEnergyOptimizer(H, D=D, depth=depth, initial_guess=rng.standard_normal(2 * depth), settings={'verbose': False})
        opt.change_settings({'method': 'Rotosolve', 'maxiter': maxiter})
        t = time.perf_counter(); res = opt.optimize(); dt = time.perf_counter() - t
        print(f'D={D} depth={depth} P={2*depth} sweeps={maxiter}: {dt*1e3:.2f} ms, {dt / (maxiter * 2 * depth) * 1e6:.1f} us per parameter update, E = {opt.optimized_result.fun:.6f}', flush=True)

# the reference's DEFAULT optimisers (Nelder-Mead; BFGS / L-BFGS-B) with batched evaluations against the scalar path
for D, depth in ((2, 2), (4, 2), (8, 3)):
    for method in ('Nelder-Mead', 'BFGS', 'L-BFGS-B'):
        for batched in (True, False):
            rng = np.random.default_rng(D)
            x0 = rng.standard_normal(2 * depth)
            for rep in range(2):            # first run: warm-up
                opt = SparseFullEnergyOptimizer(H, D=D, depth=depth, initial_guess=x0.copy(),
                                                settings={'verbose': False, 'store_values': False, 'method': method, 'tol': 1e-8, 'maxiter': 2000, 'batched': batched})
                t = time.perf_counter(); res = opt.optimize(); dt = time.perf_counter() - t
            print(f'D={D} depth={depth} {method:11s} {"batched" if batched else "scalar ":7s}: {dt*1e3:8.2f} ms, nit={res.nit}, nfev={res.nfev}, '
                  f'launches={getattr(res, "n_batches", "-")}, E = {res.fun:.8f}', flush=True)
