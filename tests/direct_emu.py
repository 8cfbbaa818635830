"""ctypes loader of tests/csrc/libdirect_emu.so: the mathematics of energy_direct_d4_kernel (qmps_direct_core.h)
run on the CPU with the four lanes of a DPP quad in lock-step.  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(['make', '-C', _HERE, '--no-print-directory'], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(os.path.join(_HERE, 'libdirect_emu.so'))
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        L.direct_emu_d4.argtypes = [ctypes.c_long, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, dp, dp, ip, ip, dp, dp]
        L.direct_emu_d4.restype = ctypes.c_int
        _LIB = L
    return _LIB


def energies_d4(A, h, max_iter=10000, tol=1e-13):
    """A (B,2,4,4), h (nt,4,4) -> dict(E (B,nt), r (B,4,4), rho (B,4,4), iters, status, resid)."""
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    A = np.ascontiguousarray(A, dtype=np.complex128)
    h = np.ascontiguousarray(np.asarray(h, dtype=np.complex128).reshape(-1, 4, 4))
    B, nt = len(A), len(h)
    E = np.empty((B, nt))
    r = np.empty((B, 4, 4), dtype=np.complex128)
    rho = np.empty((B, 4, 4), dtype=np.complex128)
    it = np.empty(B, dtype=np.int32)
    st = np.empty(B, dtype=np.int32)
    res = np.empty(B)
    Elean = np.empty((B, nt))
    lib().direct_emu_d4(B, A.ctypes.data_as(dp), h.ctypes.data_as(dp), nt, int(max_iter), float(tol), E.ctypes.data_as(dp),
                        r.ctypes.data_as(dp), rho.ctypes.data_as(dp), it.ctypes.data_as(ip), st.ctypes.data_as(ip),
                        res.ctypes.data_as(dp), Elean.ctypes.data_as(dp))
    return {'E': E, 'r': r, 'rho': rho, 'iters': it, 'status': st, 'resid': res, 'E_lean': Elean}
