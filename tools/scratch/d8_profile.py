"""phase clocks of the D = 8 whole-run rotosolve kernel (library built with -DQMPS_D8_PROFILE into qmps_amd/lib/libqmps_hip_prof.so):
wall_clock64 ticks (100 MHz) of restart 0, sweep 1, parameter 1: ansatz build | solve + energies | wait for the other waves | update"""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['QMPS_HIP_LIB'] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'qmps_amd', 'lib', 'libqmps_hip_prof.so')
from qmps_amd import EnergyEngine, _lib as L
from qmps_amd.engine import _f64
eng = EnergyEngine(8, 4096)
X = np.array([[0, 1], [1, 0]], dtype=complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.diag([1.0, -1.0]).astype(complex)
eng.set_hamiltonian(np.kron(X, X) + np.kron(Y, Y) + 0.5 * np.kron(Z, Z))
R, P, sweeps = 256, 6, 4
P0 = np.ascontiguousarray(np.random.default_rng(1).standard_normal((R, P)))
hist = np.zeros(sweeps * R + 8)
for nsh, fn in ((3, eng._lib.qmps_rotosolve), (6, eng._lib.qmps_double_rotosolve)):
    Pc = P0.copy()
    L.check(fn(eng._ctx, R, 0, P, _f64(Pc), sweeps, 10000, 1e-13, _f64(hist)))
    print('nsh', nsh, 'ticks (10 ns):', hist[sweeps * R:sweeps * R + 5])
