"""phase clocks of the D = 8 whole-run rotosolve kernel (library built with -DQMPS_D8_PROFILE into qmps_amd/lib/libqmps_hip_prof.so):
wall_clock64 ticks (100 MHz) of restart 0, sweep 1, parameter 1: ansatz build | solve + energies | wait for the other waves | update"""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('QMPS_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'qmps_amd', 'lib', 'libqmps_hip_prof.so'))
from qmps_amd import EnergyEngine, _lib as L
from qmps_amd.engine import _f64
eng = EnergyEngine(8, 4096)
X = np.array([[0, 1], [1, 0]], dtype=complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.diag([1.0, -1.0]).astype(complex)
eng.set_hamiltonian(np.kron(X, X) + np.kron(Y, Y) + 0.5 * np.kron(Z, Z))
P, sweeps = 6, int(os.environ.get("SWEEPS", "4"))
for R, sweeps in ((256, 8), (256, sweeps), (256, sweeps)):
    P0 = np.ascontiguousarray(np.random.default_rng(1).standard_normal((R, P)))
    hist = np.zeros(sweeps * R + 16 + 3 * 4096)
    Pc = P0.copy()
    import time; t0 = time.perf_counter()
    L.check(eng._lib.qmps_rotosolve(eng._ctx, R, 0, P, _f64(Pc), sweeps, 10000, 1e-13, _f64(hist)))
    dt = time.perf_counter() - t0
    print('host: %.1f us per update; in-kernel whole-run ticks / update of restart 0, R/2, R-1:' % (dt * 1e6 / (sweeps * P)), hist[sweeps * R + 5:sweeps * R + 8] / (sweeps * P))
    print('   power steps: max, total', hist[sweeps * R + 15:sweeps * R + 16].view(np.int32), 'of', sweeps * P * 3 * R + 3 * R, 'evaluations')
    per = hist[sweeps * R + 16:sweeps * R + 16 + 3 * R].reshape(R, 3)
    tk = per[:, 0] / (sweeps * P); hw = per[:, 1].astype(np.int64); xcc = per[:, 2].astype(np.int64) & 0xf
    order = np.argsort(tk)[::-1][:6]
    print('   slowest restarts (ticks / update, restart, xcc, se, sh, cu, simd):', [(round(float(tk[o]), 1), int(o), int(xcc[o]), int((hw[o] >> 13) & 7), int((hw[o] >> 12) & 1), int((hw[o] >> 8) & 15), int((hw[o] >> 4) & 3)) for o in order], 'median', float(np.median(tk)))
    print('restarts', R, 'ticks (10 ns): ansatz | solve + energies | wait | update | total:', hist[sweeps * R:sweeps * R + 5])
    print('   fine (matrix build | layout | elimination | solve tail | acceptance | LDL | energies):', hist[sweeps * R + 8:sweeps * R + 15])
