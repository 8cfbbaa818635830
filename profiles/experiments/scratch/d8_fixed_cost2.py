"""the ~8 ms one-off in the first long rotosolve call of a process: is it the first LARGE pageable device-to-host copy?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qmps_amd import EnergyEngine, _lib as L
import bench
variant = sys.argv[1]
eng = EnergyEngine(8, 768)
eng.set_hamiltonian(bench.xxz_h(0.5))
p0 = np.random.default_rng(0).standard_normal((256, 6))
eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, p0, 8)
if variant == 'big_d2h':
    eng.tensors(256)
    t0 = time.perf_counter(); eng.tensors(256); print('second tensors() read-back: %.0f us' % ((time.perf_counter() - t0) * 1e6))
eng.sync()
out = []
for _ in range(3):
    t0 = time.perf_counter()
    eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, p0, 160)
    eng.sync()
    out.append((time.perf_counter() - t0) * 1e6 / 960)
print(variant, out)
