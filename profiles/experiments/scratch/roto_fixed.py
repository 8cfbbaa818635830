import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from qmps_amd import EnergyEngine, _lib as L
for D, R, P in ((4, 21845, 4), (8, 256, 6)):
    eng = EnergyEngine(D, 3 * R); eng.set_hamiltonian(bench.tfim_h(1.0))
    p0 = np.random.default_rng(0).standard_normal((R, P))
    for _ in range(20): eng.probe_fp64_tflops()
    eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, p0, 8)
    for n in (1, 2, 10, 40, 160):
        t = time.perf_counter(); eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, p0, n); dt = time.perf_counter() - t
        print(f'D={D} sweeps={n}: {dt*1e3:.3f} ms total, {dt/n*1e6:.1f} us per sweep', flush=True)
    eng.close()
