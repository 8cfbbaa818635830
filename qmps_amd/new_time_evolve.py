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
