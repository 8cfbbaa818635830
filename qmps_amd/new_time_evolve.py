"""Host-side mirror of the time-evolution objective (`qmps/new_time_evolve.py:193-221`,
`scripts/loschmidt.py:209-239`), D = 2.

The reference projects  W . |A A>  back onto the manifold of one-site iMPS by maximising the overlap
with |B(p) B(p)>: it builds the mixed transfer map, asks xmps for its right fixed point, embeds it in
two unitaries and simulates a 6-qubit circuit whose amplitude is  psi[0] = eta / 2  (the dominant
eigenvalue of that map; the in-file asserts new_time_evolve.py:100-184 and SURVEY App. B-3), returning
`-sqrt(2 |psi[0]|) = -sqrt(|eta|)`.  Here eta comes straight from libqmps_hip (`qmps_overlap_batch`):
one launch evaluates a whole batch of candidate parameter vectors against the current state.
"""
import numpy as np
from scipy.optimize import minimize

from . import _lib as L
from . import _runtime
from .represent import ShallowFullStateTensor, unitary
from .tools import unitary_to_tensor


def gate(v, symbol='U'):
    """The candidate state tensor's gate (new_time_evolve.py:186-187, scripts/loschmidt.py:203-207)."""
    return ShallowFullStateTensor(2, v, symbol)


def state_tensor(p):
    """A(p) = unitary_to_tensor(unitary(gate(p))); already left-canonical (a unitary's first D columns)."""
    return unitary_to_tensor(unitary(gate(p)))


def batch_obj(P, A, WW, return_eta=False):
    """-sqrt(|eta|) for every row of P (B, 15) against the current state A (2,2,2): one kernel launch."""
    P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64)
    eng = _runtime.engine(2, P.shape[0])
    eta, rounds, st = eng.overlaps(A, P[:, :15], WW, kind='params', ansatz=L.ANSATZ_SHALLOW_FULL)
    f = -np.sqrt(np.abs(eta))
    f = np.where(st == L.STATUS_OK, f, np.nan)
    return (f, eta) if return_eta else f


def obj(p, A, WW):
    """Scalar objective with the reference's signature `obj(p, A, WW)` (extra entries of p beyond the 15
    gate angles - the reference's unused `rs` - are ignored)."""
    return float(batch_obj(np.asarray(p, dtype=float)[None, :15], A, WW)[0])


def evolve(params, WW, n_steps, method='Nelder-Mead', options=None, callback=None):
    """The reference's time-evolution loop (new_time_evolve.py:276-292): at each step the current tensor
    A = A(params) is fixed and the next parameters maximise the overlap with W . |A A>."""
    params = np.array(params, dtype=float)
    history = [params.copy()]
    for step in range(n_steps):
        A = state_tensor(params)
        res = minimize(obj, params, (A, WW), method=method, options=options or {})
        params = res.x
        history.append(params.copy())
        if callback is not None:
            callback(step, params, res.fun)
    return np.array(history)


# ---- the rest of the reference module's surface ------------------------------------------------------------
def state_gate(v, symbol='R'):
    """new_time_evolve.py:189-190."""
    from .represent import StateGate
    return StateGate(v, symbol)


def obj_H():
    """Projector observable of the rotosolve variant (new_time_evolve.py:247-248): -|0..0><0..0| on 5 qubits."""
    return -np.diag(np.eye(2 ** 5)[0])


class OverlapOptimizer:
    """`Optimizer` subclass of new_time_evolve.py:18-47 (written there as a `def`, so never instantiable): maximise
    the overlap of |gate(params)> with W |u>.  objective = -2 |psi[0]| of the reference's 6-qubit circuit = -|eta|
    (SURVEY App. B-3), eta from `qmps_overlap_batch`."""

    def __init__(self, u, W, v=None, initial_guess=None, obj_fun=None, args=None, gate=gate):
        from .tools import Optimizer
        self._base = Optimizer(u, v, initial_guess, obj_fun, args)
        self._base.objective_function = self.objective_function
        self.u, self.W, self.gate = u, np.asarray(W, dtype=complex), gate
        self.settings = self._base.settings

    def _current_tensor(self):
        U = self.u if isinstance(self.u, np.ndarray) else unitary(self.u)
        return unitary_to_tensor(U)

    def batch_objective_function(self, P):
        P = np.atleast_2d(np.asarray(P, dtype=float))
        cand = np.stack([unitary_to_tensor(unitary(self.gate(p))) for p in P])
        eng = _runtime.engine(2, len(cand))
        eta, _, st = eng.overlaps(self._current_tensor(), cand, self.W, kind='tensor')
        return np.where(st == L.STATUS_OK, -np.abs(eta), np.nan)

    def objective_function(self, params):
        return float(self.batch_objective_function(params)[0])

    def change_settings(self, new_settings):
        return self._base.change_settings(new_settings)

    def optimize(self):
        self._base.initial_guess = self._base.initial_guess if self._base.initial_guess is not None else np.random.randn(15)
        self._base.optimize()
        self.optimized_result = self._base.optimized_result
        return self


def one_site_expectations(A, ops):
    """<O> for each one-site operator O of an iMPS tensor A (2,2,2) (xmps `iMPS.Es(ops)`, new_time_evolve.py:288):
    two-site energies of O x 1 on the device."""
    h = np.stack([np.kron(np.asarray(O, dtype=complex), np.eye(2)) for O in ops])
    eng = _runtime.engine(2, 1)
    E, _, st = eng.energies(np.asarray(A, dtype=complex)[None], h)
    if st[0] == L.STATUS_NOT_CONVERGED:
        raise np.linalg.LinAlgError('environment did not converge')
    return E[0]


def loschmidt_overlap(A, B):
    """|x|^2 per site between two iMPS tensors (xmps `iMPS.overlap`, new_time_evolve.py:289)."""
    from .time_evolve_tools import overlap_of_tensors
    return overlap_of_tensors(A, B)


def run(params, WW, T, ops=None, method='Nelder-Mead', options=None):
    """The reference's `__main__` loop without the plots (new_time_evolve.py:250-294): evolve over the time grid T,
    recording parameters, one-site expectation values and the Loschmidt echo against the initial state.
    Returns (ps, evs, les)."""
    params = np.array(params, dtype=float)
    A0 = state_tensor(params)
    if ops is None:
        ops = [0.5 * np.array([[0, 1], [1, 0]]), 0.5 * np.array([[0, -1j], [1j, 0]]), 0.5 * np.diag([1.0, -1.0])]
    ps, evs, les = [params.copy()], [], []
    for _ in T[1:]:
        A = state_tensor(params)
        res = minimize(obj, params, (A, WW), method=method, options=options or {})
        params = res.x
        evs.append(one_site_expectations(A, ops))
        les.append(loschmidt_overlap(A, A0))
        ps.append(params.copy())
    return np.array(ps), np.array(evs), np.array(les)
