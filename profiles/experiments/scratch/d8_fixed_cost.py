"""where do ~8 ms go in a 160-sweep D = 8 rotosolve call that follows an 8-sweep one?  Variants of what precedes the timed call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qmps_amd import EnergyEngine, _lib as L
import bench
def trial(name, pre):
    eng = EnergyEngine(8, 4096)
    eng.set_hamiltonian(bench.xxz_h(0.5))
    p0 = np.random.default_rng(20241022).standard_normal((256, 6))
    pre(eng, p0)
    eng.sync()
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, p0, 160)
        eng.sync()
        out.append((time.perf_counter() - t0) * 1e6 / 960)
    print('%-58s us per update of three consecutive 160-sweep calls: %s' % (name, ' '.join('%.1f' % o for o in out)))
    eng.close()
trial('nothing before', lambda e, p: None)
trial('8 sweeps before', lambda e, p: e.rotosolve(L.ANSATZ_SHALLOW_CNOT, p, 8))
trial('8 sweeps twice before', lambda e, p: (e.rotosolve(L.ANSATZ_SHALLOW_CNOT, p, 8), e.rotosolve(L.ANSATZ_SHALLOW_CNOT, p, 8)))
trial('8 sweeps, 50 ms sleep', lambda e, p: (e.rotosolve(L.ANSATZ_SHALLOW_CNOT, p, 8), time.sleep(0.05)))
trial('64 sweeps before', lambda e, p: e.rotosolve(L.ANSATZ_SHALLOW_CNOT, p, 64))
def probe(e, p):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.06: e.probe_fp64_tflops()
trial('60 ms FP64 probe', probe)
trial('60 ms FP64 probe + 8 sweeps', lambda e, p: (probe(e, p), e.rotosolve(L.ANSATZ_SHALLOW_CNOT, p, 8)))
