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
  CircuitSolver / Represent / Optimize / Evolve / Optimizer drivers: second half of this file
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

    @staticmethod
    def _sites(O):
        """2 or 4: a two-site operator comes as (4, 4) or (2,)*4, a four-site one as (16, 16) or (2,)*8, each
        optionally behind a batch axis."""
        shp = np.shape(O)
        if shp[-2:] == (16, 16) or (len(shp) >= 8 and shp[-8:] == (2,) * 8):
            return 4
        if shp[-2:] == (4, 4) or (len(shp) in (4, 5) and shp[-4:] == (2,) * 4):
            return 2
        raise ValueError(f'operator of shape {shp} is neither a two- nor a four-site operator')

    def expectation_value(self, U1, U2, O, path=None):
        O = np.asarray(O)
        return self.qbt2_exp_val(U1, U2, O) if self._sites(O) == 2 else self.qbt4_exp_val(U1, U2, O)

    def mexpectation_value(self, U1, U2, O):
        O = np.asarray(O)
        return self.mqbt2_exp_val(U1, U2, O) if self._sites(O) == 2 else self.mqbt4_exp_val(U1, U2, O)

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


# ---------------------------------------------------------------------------------------------------------
# Driver classes (the callers of the contractions above): same names, arguments and result objects as
# `new_tdvp/ClassicalTDVPStripped.py`.  They are host glue - scipy optimisers over device contractions - so that a
# script written against the reference's `Optimizer().optimize / .represent / .evolve` runs unchanged.
#
#   ResultObject, gradient_descent                  :51-91
#   CircuitSolver (D, D2, X, Z, D1, D3, M, paramU)   :94-166, OO_lambdas / OO_unitary :30-48
#   state_from_params, optimize_2layer_bwmps        :193-225
#   Represent :599-655, Optimize :658-723, Evolve :726-925, Optimizer :927-944
# ---------------------------------------------------------------------------------------------------------
from scipy.linalg import expm                       # noqa: E402
from scipy.optimize import approx_fprime, minimize  # noqa: E402

from .ground_state import U4                        # noqa: E402


class ResultObject:
    def __init__(self, x, fun, nfev, message):
        self.x, self.fun, self.nfev, self.message = x, fun, nfev, message


def gradient_descent(cf, gf, init, lr=0.01, tol=1e-8, miter=10000, atol=1e-6):
    """Step-halving gradient descent towards cf == 0 (:51-83): a rejected step halves the learning rate; stops when
    |cf| < atol, when cf stops changing (tol) or after miter steps.  nfev is reported as 16 per step like the reference."""
    theta = np.array(init, dtype=float)
    value = cf(theta)
    step = 0
    while True:
        trial = theta - lr * gf(theta)
        trial_value = cf(trial)
        if abs(value) < atol:
            return ResultObject(theta, value, step * 16, 'Answer Reached')
        if abs(trial_value) < atol:
            return ResultObject(trial, trial_value, step * 16, 'Answer reached')
        if trial_value > value:
            lr /= 2
            step += 1
            continue
        if abs(value - trial_value) < tol:
            return ResultObject(trial, trial_value, step * 16, 'CF stopped changing')
        if step == miter:
            return ResultObject(trial, trial_value, step * 16, 'Max Iter Reached')
        theta, value = trial, trial_value
        step += 1


def OO_lambdas():
    """The seven su(4) generators with a non-zero first column - enough to reach every first column of a two-qubit
    unitary (the only column a circuit acting on |00> touches): the symmetric / antisymmetric pairs (0,1), (0,2),
    (0,3) and diag(1,-1,0,0), in the Gell-Mann order lambda_1,2,3,4,5,9,10 the reference indexes (:30-36)."""
    def unit(a, b, v):
        m = np.zeros((4, 4), dtype=complex)
        m[a, b] = v
        m[b, a] = np.conj(v)
        return m
    return np.stack([unit(0, 1, 1), unit(0, 1, -1j), np.diag([1, -1, 0, 0]).astype(complex),
                     unit(0, 2, 1), unit(0, 2, -1j), unit(0, 3, 1), unit(0, 3, -1j)])


def OO_unitary(p):
    return expm(-1j * np.tensordot(np.asarray(p, dtype=float), OO_lambdas(), axes=1))


class CircuitSolver:
    """Gate matrices of the environment ansatz and the 22-parameter (U1, U2) map (:94-166)."""

    @staticmethod
    def D(theta):
        c, s = np.cos(theta) ** 2, np.sin(theta) ** 2
        return np.array([[c, s], [s, c]])

    @staticmethod
    def D2(theta):
        return np.diag([np.cos(theta) ** 2, np.sin(theta) ** 2])

    @staticmethod
    def X(theta):
        c, s = np.cos(np.pi * theta / 2), np.sin(np.pi * theta / 2)
        return np.array([[c, -1j * s], [-1j * s, c]])

    @staticmethod
    def Z(theta):
        return np.diag([1, np.exp(1j * np.pi * theta)])

    @staticmethod
    def D1(theta):
        return np.diag([np.cos(theta), -1j * np.sin(theta)])

    @staticmethod
    def D3(theta):
        return np.diag([np.cos(theta), np.sin(theta)]).astype(complex)

    def M(self, params):
        a, b, c, d, e, f = params
        return self.Z(b) @ self.X(c) @ self.Z(d) @ self.D3(a) @ self.X(e) @ self.Z(f)

    def paramU(self, params):
        """22 parameters -> (U1, U2): 15 for a full U(4) (`U4`; xmps's convention is unpinned, ours is the Gell-Mann
        exponential of qmps_amd.ground_state) and 7 for the one accessed column of U2 (`OO_unitary`)."""
        params = np.asarray(params, dtype=float)
        return U4(params[7:]), OO_unitary(params[:7])

    def batch_paramU(self, params):
        """(B, 22) -> (U1 (B,4,4), U2 (B,4,4)): parameter sets of a whole population / simplex at once."""
        params = np.atleast_2d(np.asarray(params, dtype=float))
        pairs = [self.paramU(p) for p in params]
        return np.stack([a for a, _ in pairs]), np.stack([b for _, b in pairs])


def state_from_params(p, l):
    U1, U2 = CircuitSolver().paramU(p)
    return bwMPS([U2, U1], l).state()


def optimize_2layer_bwmps(H, initial_params=None, maxiter=10000, verbose=False):
    """Nelder-Mead on the mean of the 4- and 6-qubit finite brick-wall energies (:198-225, host numpy); returns the
    objective history like the reference."""
    H = np.asarray(H).reshape(4, 4)
    H1 = tensor([np.eye(2), H, np.eye(2)])
    H2 = tensor([np.eye(4), H, np.eye(4)])

    def obj(p):
        a, b = state_from_params(p, 2), state_from_params(p, 3)
        return 0.5 * (np.real(a.conj() @ H1 @ a) + np.real(b.conj() @ H2 @ b))
    history = []

    def cb(xk):
        history.append(obj(xk))
        if verbose:
            print(history[-1])
    x0 = np.random.rand(22) if initial_params is None else np.asarray(initial_params, dtype=float)
    minimize(obj, x0, method='Nelder-Mead', options={'maxiter': maxiter}, tol=1e-8, callback=cb)
    return history


class Represent(CircuitSolver):
    """Variational search for the right-environment matrix M (:599-655).  The 4x4 environment map depends only on
    the four unitaries, so it is contracted ONCE on the device per `optimize`; the cost function itself is 16 MACs."""

    def __init__(self):
        self.RE = RightEnvironment()
        self.LE = LeftEnvironment()
        self.right_params = None
        self.path = None
        self.convergence, self.gradients, self.params_updates = [], [], []
        self._map = None

    def _set(self, U1, U2, U1_, U2_):
        self.U1, self.U2, self.U1_, self.U2_ = U1, U2, U1_, U2_
        mat = self.RE.exact_environment_circuit(U1, U2, U1_, U2_)
        self._map = np.asarray(mat).reshape(2, 2, 2, 2)

    def cost_function(self, params):
        eta, *p = params
        M = self.M(p)
        image = np.einsum('abcd,dc->ba', self._map, M)          # == RightEnvironment.circuit(U1, U2, U1_, U2_, M)
        return np.linalg.norm(eta * M - image)

    def optimize(self, U1, U2, U1_, U2_):
        self._set(U1, U2, U1_, U2_)
        res = minimize(self.cost_function, x0=[1.0, np.pi / 4, 0, 0, 0, 0, 0], method='Nelder-Mead',
                       options={'disp': False, 'xatol': 1e-8, 'fatol': 1e-8, 'maxiter': 10000})
        self.right_params = res
        return res

    def grad(self, params):
        return approx_fprime(params, self.cost_function, epsilon=1e-8)

    def optimize_by_hand(self, Us, init_params=np.array([1.0, np.pi / 4, 0, 0, 0, 0, 0]), atol=1e-4, alpha=0.1, tol=1e-6,
                         maxiter=10000):
        self._set(*Us)
        return gradient_descent(self.cost_function, self.grad, init_params)

    def exact_env(self, U1, U2, U1_, U2_):
        _, Mr = self.RE.exact_environment(U1, U2, U1_, U2_)
        _, Ml = self.LE.exact_environment(U1, U2, U1_, U2_)
        return Mr, Ml


class Optimize(CircuitSolver):
    """Variational minimisation of <O> over the brick-wall manifold (:658-723)."""

    def __init__(self):
        self.OC = OverlapCalculator()
        self.RE = Represent()
        self.path = None
        self.energy_opt = []

    def cost_function(self, params):
        U1, U2 = self.paramU(params)
        return self.OC.expectation_value(U1, U2, self.O, self.path)

    def mcost_function(self, params):
        U1, U2 = self.paramU(params)
        return np.real(self.OC.mexpectation_value(U1, U2, self.O))

    def batch_cost_function(self, params):
        """(B, 22) -> (B,) expectation values: ONE launch for a population of parameter sets."""
        U1, U2 = self.batch_paramU(params)
        return np.real(self.OC.expectation_value(U1, U2, self.O, self.path))

    def _run(self, cost, O, initial_params):
        self.O = O
        x0 = np.random.rand(22) if initial_params is None else np.asarray(initial_params, dtype=float)
        return minimize(cost, x0=x0, callback=self.callback, method='Nelder-Mead')

    def optimize(self, O, initial_params=None):
        return self._run(self.cost_function, O, initial_params)

    def moptimize(self, O, initial_params=None):
        return self._run(self.mcost_function, O, initial_params)

    def callback(self, xk):
        self.energy_opt.append(self.cost_function(xk))


class Evolve(CircuitSolver):
    """TDVP projection step: maximise |<psi(U1', U2') | W | psi(U1, U2)>|^2 over the 22 parameters of (U1', U2')
    (:726-925); `cost_function` uses the variational environment, `exact_cost_function` / `mcost_function` the
    eigenvector environment."""

    def __init__(self):
        self.MO = ManifoldOverlap()
        self.RE = Represent()
        self.path = None
        self.cf_convergence = []

    def _primed(self, params):
        U1_, U2_ = self.paramU(params)
        return U1_.conj().T, U2_.conj().T

    def cost_function(self, params):
        U1_, U2_ = self._primed(params)
        env = self.RE.optimize(self.U1, self.U2, U1_, U2_)
        M = self.M(env.x[1:])
        return -np.abs(self.MO.circuit(self.U1, self.U2, U1_, U2_, M, M, self.W)) ** 2

    def exact_cost_function(self, params):
        U1_, U2_ = self._primed(params)
        Mr, _ = self.RE.exact_env(self.U1, self.U2, U1_, U2_)
        return -np.abs(self.MO.circuit(self.U1, self.U2, U1_, U2_, Mr, Mr.conj().T, self.W)) ** 2

    mcost_function = exact_cost_function          # matrix and tensor forms share one device kernel

    def _set(self, W, U1, U2):
        self.W, self.U1, self.U2 = W, U1, U2

    def optimize(self, W, U1, U2, initial_params=None):
        x0 = np.random.rand(22) if initial_params is None else np.asarray(initial_params, dtype=float)
        self._set(W, U1, U2)
        return minimize(self.cost_function, x0=x0, callback=self.callback, method='Nelder-Mead',
                        options={'maxiter': len(x0) * 1000})

    def callback(self, xk):
        self.cf_convergence.append(self.cost_function(xk))

    def exact_callback(self, xk):
        self.cf_convergence.append(self.exact_cost_function(xk))

    def exact_optimize(self, W, U1, U2, initial_params=None, record=False):
        x0 = np.random.rand(22) if initial_params is None else np.asarray(initial_params, dtype=float)
        callback = None
        if record:
            self.cf_convergence = []
            callback = self.exact_callback
        self._set(W, U1, U2)
        return minimize(self.exact_cost_function, x0=x0, callback=callback, options={'ftol': 1e-6, 'xtol': 1e-6},
                        method='Powell')

    mexact_optimize = exact_optimize

    def time_evolve(self, steps, W, init_params=None, show_convergence=False):
        """`steps` projection steps of the propagator W, each warm-started from the previous optimum; the list of
        scipy results (:863-893; the reference's optional convergence plots are left to the caller)."""
        p = np.random.rand(22) if init_params is None else np.asarray(init_params, dtype=float)
        results = []
        for _ in range(steps):
            U1, U2 = self.paramU(p)
            res = self.exact_optimize(W, U1, U2, initial_params=p, record=show_convergence)
            results.append(res)
            p = res.x
        return results

    mtime_evolve = time_evolve


class Optimizer(CircuitSolver):
    """`.optimize`, `.represent`, `.evolve`: the three drivers behind one object (:927-944)."""

    def __init__(self):
        self.optimize = Optimize()
        self.represent = Represent()
        self.evolve = Evolve()
