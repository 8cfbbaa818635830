"""Host-side mirror of `qmps/rotosolve.py` (the older rotosolve API: state function + H) and the
batched drivers that produce the kernel's batch axis.

  rotosolve          qmps/rotosolve.py:154-181   (3 samples per parameter: theta, theta +- pi/2)
  double_rotosolve   qmps/rotosolve.py:183-241   (6 distinct shifts per parameter)

`batched_rotosolve` / `batched_double_rotosolve` run R independent restarts in lock-step: for
parameter i, all R x (3 | 6) shifted parameter vectors form ONE batch for
`batch_eps(params[B,P]) -> float[B]` (an optimiser's `batch_objective_function`), i.e. one launch
of the MI355X kernel per parameter update instead of 3R (or 10R) scalar evaluations.
"""
import numpy as np

from .tools import ROTO_SHIFTS, _double_sinusoid_shift

π = np.pi


def _wrap(x):
    return np.arctan2(np.sin(x), np.cos(x))


def rotosolve(H, state_function, initial_parameters, args=(), N_iters=10):
    """Single-frequency rotosolve on eps(x) = <psi(x)|H|psi(x)> (rotosolve.py:154-181).
    Updates `initial_parameters` in place; returns (energies, parameter history)."""
    params = initial_parameters
    I = np.eye(len(params))
    es, hist = [], []

    def eps(x):
        psi = state_function(x, *args)
        return np.real(psi.conj().T @ H @ psi)

    for _ in range(N_iters):
        for i in range(len(params)):
            e0, ep, em = eps(params), eps(params + I[i] * π / 2), eps(params - I[i] * π / 2)
            params[i] += _wrap(-π / 2 - np.arctan2(2 * e0 - ep - em, ep - em))
            params[i] = _wrap(params[i])
        es.append(eps(params))
        hist.append(params.copy())
    return es, hist


def double_rotosolve(H, state_function, initial_parameters, args=(), N_iters=5):
    """Double-frequency rotosolve (rotosolve.py:183-241); returns (energies, params)."""
    params = initial_parameters
    I = np.eye(len(params))
    es = []

    def eps(x):
        psi = state_function(x, *args)
        return np.real(psi.conj().T @ H @ psi)

    for _ in range(N_iters):
        for i in range(len(params)):
            M = [np.sum(eps(params + I[i] * x)) for x in ROTO_SHIFTS]
            params[i] += _double_sinusoid_shift(*M)
        es.append(eps(params))
    return np.array(es), params


def batched_rotosolve(batch_eps, initial_parameters, N_iters=10):
    """R restarts x P parameters, single-frequency update.  initial_parameters: (R, P).
    Returns (energies (N_iters, R), params (R, P))."""
    params = np.array(initial_parameters, dtype=float, copy=True)
    R, P = params.shape
    shifts = np.array([0.0, π / 2, -π / 2])
    es = []
    for _ in range(N_iters):
        for i in range(P):
            batch = np.repeat(params[:, None, :], 3, axis=1)
            batch[:, :, i] += shifts
            e = np.asarray(batch_eps(batch.reshape(R * 3, P))).reshape(R, 3)
            theta = -π / 2 - np.arctan2(2 * e[:, 0] - e[:, 1] - e[:, 2], e[:, 1] - e[:, 2])
            params[:, i] = _wrap(params[:, i] + _wrap(theta))
        es.append(np.asarray(batch_eps(params)))
    return np.array(es), params


def batched_double_rotosolve(batch_eps, initial_parameters, N_iters=5):
    """R restarts x P parameters, six shifts per parameter: batches of 6 R evaluations."""
    params = np.array(initial_parameters, dtype=float, copy=True)
    R, P = params.shape
    es = []
    for _ in range(N_iters):
        for i in range(P):
            batch = np.repeat(params[:, None, :], 6, axis=1)
            batch[:, :, i] += ROTO_SHIFTS
            e = np.asarray(batch_eps(batch.reshape(R * 6, P))).reshape(R, 6)
            for r in range(R):
                if np.all(np.isfinite(e[r])):
                    params[r, i] += _double_sinusoid_shift(*e[r])
        es.append(np.asarray(batch_eps(params)))
    return np.array(es), params


def device_rotosolve(optimizer, initial_parameters, N_iters=10):
    """R restarts x P parameters entirely on the GPU (libqmps_hip `qmps_rotosolve`): `optimizer` is a
    SparseFullEnergyOptimizer whose ansatz class the library can simulate (`device_kind`).  Same update
    rule as `batched_rotosolve`; returns (energies (N_iters, R), params (R, P))."""
    from . import _runtime
    from .ground_state import _as_h
    kind = getattr(optimizer.state_tensor, 'device_kind', None)
    if kind is None:
        raise ValueError(f'{optimizer.state_tensor.__name__} has no device implementation; use batched_rotosolve')
    P = np.atleast_2d(np.asarray(initial_parameters, dtype=float))
    eng = _runtime.engine(optimizer.D, 3 * P.shape[0])
    eng.set_hamiltonian(_as_h(optimizer.H))
    return eng.rotosolve(kind, P, N_iters, max_iter=optimizer.max_iter, tol=optimizer.env_tol)
