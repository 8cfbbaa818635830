import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from qmps_amd.ground_state import Hamiltonian, SparseFullEnergyOptimizer
H = Hamiltonian({'ZZ': -1, 'X': 1}).to_matrix()
for D, depth in ((2, 1), (4, 2), (8, 3)):
    opt = SparseFullEnergyOptimizer(H, D=D, depth=depth, settings={'verbose': False})
    p = np.random.default_rng(0).standard_normal(2 * depth)
    for _ in range(20): opt.objective_function(p)
    t = time.perf_counter()
    for _ in range(500): opt.objective_function(p + 1e-3)
    print(f'D={D}: {(time.perf_counter() - t) / 500 * 1e6:.1f} us per objective_function call (one evaluation, host in / host out)')
