"""ctypes loader for oracle/libqmps_oracle.so (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-C', _HERE, '--no-print-directory'], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libqmps_oracle.so')
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        L.qmps_oracle_energy_batch.argtypes = [ctypes.c_int, ctypes.c_long, dp, dp, ctypes.c_int, dp, ctypes.c_int,
                                               ctypes.c_double, dp, ip, ip, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.qmps_oracle_energy_batch.restype = ctypes.c_int
        L.qmps_oracle_unitary_to_tensor.argtypes = [ctypes.c_int, ctypes.c_long, dp, dp]
        L.qmps_oracle_unitary_to_tensor.restype = ctypes.c_int
        L.qmps_oracle_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _dp(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def energy_batch(A, h, r0=None, max_iter=10000, tol=1e-13, threads=1, want_r=False, want_rho=False, handoff=None, skip=0, period=0):
    """A (B,2,D,D) c128, h (nt,4,4) or (4,4) c128 -> dict(E (B,nt), iters, status, r?, rho?)."""
    A = np.ascontiguousarray(A, dtype=np.complex128)
    h = np.ascontiguousarray(np.asarray(h, dtype=np.complex128).reshape(-1, 4, 4))
    B, _, D, _ = A.shape
    nt = h.shape[0]
    E = np.empty((B, nt))
    it = np.empty(B, dtype=np.int32)
    st = np.empty(B, dtype=np.int32)
    r = np.empty((B, D, D), dtype=np.complex128) if want_r else None
    rho = np.empty((B, 4, 4), dtype=np.complex128) if want_rho else None
    r0c = None if r0 is None else np.ascontiguousarray(r0, dtype=np.complex128)
    ip = ctypes.POINTER(ctypes.c_int)
    rc = lib().qmps_oracle_energy_batch(D, B, _dp(A.view(np.float64)), _dp(h.view(np.float64)), nt,
                                        _dp(None if r0c is None else r0c.view(np.float64)), int(max_iter), float(tol),
                                        _dp(E), it.ctypes.data_as(ip), st.ctypes.data_as(ip),
                                        _dp(None if r is None else r.view(np.float64)),
                                        _dp(None if rho is None else rho.view(np.float64)), int(threads), -1 if handoff is None else int(handoff), int(skip), int(period))
    if rc != 0:
        raise ValueError('qmps_oracle_energy_batch: bad arguments')
    return {'E': E, 'iters': it, 'status': st, 'r': r, 'rho': rho}


def unitary_to_tensor(U):
    U = np.ascontiguousarray(U, dtype=np.complex128)
    B, N, _ = U.shape
    D = N // 2
    A = np.empty((B, 2, D, D), dtype=np.complex128)
    lib().qmps_oracle_unitary_to_tensor(D, B, _dp(U.view(np.float64)), _dp(A.view(np.float64)))
    return A


def max_threads():
    return lib().qmps_oracle_max_threads()
