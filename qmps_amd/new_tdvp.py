"""Host-side mirror of the brick-wall classical stack `new_tdvp/ClassicalTDVPStripped.py`,
with the contractions running on the MI355X (`qmps_bw_*` entry points of libqmps_hip).

Same class and method names as the reference; every method also accepts a leading batch axis (one launch for
the whole batch).  Tensors follow the reference's convention: a two-qubit unitary is passed either as a 4 x 4
matrix or as `U.reshape(2, 2, 2, 2)` = [out0, out1, in0, in1].

  bwMPS.state                                :179-191  (data helper, host numpy)
  OverlapCalculator.expectation_value        :442-447  -> qbt2_exp_val :511-544 / qbt4_exp_val :464-496
  OverlapCalculator.mexpectation_value       :449-454  -> mqbt2_exp_val :546-555 / mqbt4_exp_val :498-507
  RightEnvironment / LeftEnvironment         :314-431  circuit, exact_environment_circuit, exact_environment
  ManifoldOverlap.circuit / mcircuit         :239-285
  Represent.exact_env                        :652-655
"""
import ctypes
from ctypes import c_double, c_int32
from functools import reduce

import numpy as np

from . import _lib as L
from . import _runtime

_dp = ctypes.POINTER(c_double)
_ip = ctypes.POINTER(c_int32)


def _f64(a):
    return a.view(np.float64).ctypes.data_as(_dp)


def _mats(U, n=4):
    """(..., n, n) or (..., 2,2,..) tensor form -> contiguous (B, n, n) complex128, and whether a batch axis was given."""
    U = np.asarray(U, dtype=np.complex128)
    k = int(np.log2(n))
    if U.shape[-2:] != (n, n) or (U.ndim >= 2 * k and U.shape[-2 * k:] == (2,) * (2 * k) and n != 2):
        U = U.reshape(U.shape[:-2 * k] + (n, n)) if U.shape[-2 * k:] == (2,) * (2 * k) else U
    batched = U.ndim == 3
    return np.ascontiguousarray(U.reshape(-1, n, n)), batched


def _ctx(B):
    eng = _runtime.engine(2, max(int(B), 1))
    return eng._lib, eng._ctx


def tensor(tensors):
    return reduce(np.kron, tensors)


class bwMPS:
    """Brick-wall state on 2 l qubits: layer 0 = Us[0] on pairs (0,1),(2,3).., layer 1 = Us[1] on (1,2),(3,4).. (:174-191)."""

    def __init__(self, Us, l):
        self.layers = len(Us)
        self.Us = Us
        self.l = l

    tensor = staticmethod(tensor)

    def state(self):
        I = np.eye(2)
        psi = np.zeros(2 ** (2 * self.l), dtype=complex)
        psi[0] = 1
        for i, u in enumerate(self.Us):
            u = np.asarray(u).reshape(4, 4)
            psi = (tensor([u] * self.l) if i % 2 == 0 else tensor([I] + [u] * (self.l - 1) + [I])) @ psi
        return psi


class OverlapCalculator:
    """Expectation values of brick-wall states."""

    def _expval(self, U1, U2, O, sites):
        n = 4 if sites == 2 else 16
        U1, b1 = _mats(U1)
        U2, _ = _mats(U2)
        O, bo = _mats(O, n)
        B = len(U1)
        if len(U2) != B or (bo and len(O) != B):
            raise ValueError('batch sizes differ')
        out = np.empty(B, dtype=np.complex128)
        lib, ctx = _ctx(B)
        L.check(lib.qmps_bw_expval(ctx, B, sites, _f64(U1), _f64(U2), _f64(O), 0 if bo else 1, _f64(out)))
        return out if b1 else out[0]

    def qbt2_exp_val(self, U1, U2, O, path=None):
        return np.real(self._expval(U1, U2, O, 2))

    def qbt4_exp_val(self, U1, U2, O, path=None):
        return np.real(self._expval(U1, U2, O, 4))

    def mqbt2_exp_val(self, U1, U2, O):
        return self._expval(U1, U2, O, 2)            # the reference's matrix form returns the complex value

    def mqbt4_exp_val(self, U1, U2, O):
        return np.real(self._expval(U1, U2, O, 4))

    def expectation_value(self, U1, U2, O, path=None):
        O = np.asarray(O)
        return self.qbt2_exp_val(U1, U2, O) if O.size // max(1, np.asarray(U1).size // 16) == 16 else self.qbt4_exp_val(U1, U2, O)

    def mexpectation_value(self, U1, U2, O):
        O = np.asarray(O)
        return self.mqbt2_exp_val(U1, U2, O) if O.shape[-1] == 4 else self.mqbt4_exp_val(U1, U2, O)

    def path(self, O):
        return None          # einsum contraction paths are a numpy artefact; nothing to precompute here


class _Environment:
    side = 0

    def _env(self, U1, U2, U1_, U2_, want_matrix):
        U1, b1 = _mats(U1)
        U2, _ = _mats(U2)
        U1_, _ = _mats(U1_)
        U2_, _ = _mats(U2_)
        B = len(U1)
        mat = np.empty((B, 4, 4), dtype=np.complex128) if want_matrix else None
        eta = np.empty(B, dtype=np.complex128)
        vec = np.empty((B, 2, 2), dtype=np.complex128)
        st = np.empty(B, dtype=np.int32)
        lib, ctx = _ctx(B)
        L.check(lib.qmps_bw_env(ctx, B, self.side, _f64(U1), _f64(U2), _f64(U1_), _f64(U2_), 40, 1e-13,
                                None if mat is None else _f64(mat), _f64(eta), _f64(vec), st.ctypes.data_as(_ip)))
        return mat, eta, vec, st, b1

    def exact_environment_circuit(self, U1, U2, U1_, U2_):
        mat, _, _, _, b = self._env(U1, U2, U1_, U2_, True)
        return mat if b else mat[0]

    def exact_environment(self, U1, U2, U1_, U2_):
        """(eta, vec): the eigenpair with the largest real part (the reference's `eta[np.argmax(eta)]`); vec has
        unit 2-norm and its largest entry real positive (the reference returns LAPACK's arbitrary phase)."""
        _, eta, vec, st, b = self._env(U1, U2, U1_, U2_, False)
        if np.any(st != 0):
            raise np.linalg.LinAlgError('environment eigenpair did not converge (degenerate leading eigenvalues)')
        return (eta, vec) if b else (eta[0], vec[0])


class RightEnvironment(_Environment):
    side = 0

    def circuit(self, U1, U2, U1_, U2_, M, path=None):
        """result[j][i] = <j,0,0| U2'_bc U1'_ab M_c U1_ab U2_bc |i,0,0> = sum Mmat[(i,j),(k,k')] M[k'][k] (:355-379)."""
        mat = self.exact_environment_circuit(U1, U2, U1_, U2_)
        M = np.asarray(M, dtype=complex)
        m4 = mat.reshape(mat.shape[:-2] + (2, 2, 2, 2))          # (i, i', j, j')
        return np.einsum('...abcd,...dc->...ba', m4, M)

    def path(self):
        return None


class LeftEnvironment(_Environment):
    side = 1


class ManifoldOverlap:
    def circuit(self, U1, U2, U1_, U2_, Mr, Ml, W, path=None):
        U1, b1 = _mats(U1)
        U2, _ = _mats(U2)
        U1_, _ = _mats(U1_)
        U2_, _ = _mats(U2_)
        Mr, bm = _mats(Mr, 2)
        Ml, _ = _mats(Ml, 2)
        W, bw = _mats(W, 16)
        B = len(U1)
        out = np.empty(B, dtype=np.complex128)
        lib, ctx = _ctx(B)
        L.check(lib.qmps_bw_manifold(ctx, B, _f64(U1), _f64(U2), _f64(U1_), _f64(U2_), _f64(Mr), _f64(Ml), 0 if bm else 1,
                                     _f64(W), 0 if bw else 1, _f64(out)))
        return out if b1 else out[0]

    mcircuit = circuit

    def path(self):
        return None


class Represent:
    def __init__(self):
        self.RE = RightEnvironment()
        self.LE = LeftEnvironment()

    def exact_env(self, U1, U2, U1_, U2_):
        _, Mr = self.RE.exact_environment(U1, U2, U1_, U2_)
        _, Ml = self.LE.exact_environment(U1, U2, U1_, U2_)
        return Mr, Ml
