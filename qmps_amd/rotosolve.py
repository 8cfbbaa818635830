"""Host-side mirror of `qmps/rotosolve.py` (the older rotosolve API: state function + H) and the
batched drivers that produce the kernel's batch axis.

  rotosolve          qmps/rotosolve.py:154-181   (3 samples per parameter: theta, theta +- pi/2)
  double_rotosolve   qmps/rotosolve.py:183-241   (6 distinct shifts per parameter)
  gate, op_state, op_H, evo_Hs, swapper   qmps/rotosolve.py:15-62, 63-65, 106-110 (state functions of the
                     variational-environment problem; `evo_state` references an undefined name in the reference and the
                     `sinusoids` plots are figure tooling: not mirrored)

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
            ok = np.all(np.isfinite(e), axis=1)      # a sample without a valid environment leaves the restart's parameter untouched
            params[ok, i] = _wrap(params[ok, i] + _wrap(theta[ok]))
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


def device_double_rotosolve(optimizer, initial_parameters, N_iters=5):
    """Double-frequency rotosolve (tools.py:422-457) for R restarts entirely on the GPU (`qmps_double_rotosolve`): six
    shifts per parameter, the P sin(2x+u) + Q sin(x+v) fit and its global argmin in the update kernel; no host round
    trip per parameter.  Returns (energies (N_iters, R), params (R, P))."""
    from . import _runtime
    from .ground_state import _as_h
    kind = getattr(optimizer.state_tensor, 'device_kind', None)
    if kind is None:
        raise ValueError(f'{optimizer.state_tensor.__name__} has no device implementation; use batched_double_rotosolve')
    P = np.atleast_2d(np.asarray(initial_parameters, dtype=float))
    eng = _runtime.engine(optimizer.D, 6 * P.shape[0])
    eng.set_hamiltonian(_as_h(optimizer.H))
    return eng.double_rotosolve(kind, P, N_iters, max_iter=optimizer.max_iter, tol=optimizer.env_tol)


# ---- state functions of the variational-environment problem (rotosolve.py:15-62, 109-113) -----------------
# Host state vectors with the cirq-free circuit model of qmps_amd.represent, so that the reference's own driver call
# `rotosolve(op_H(H), op_state, params)` runs unchanged.  The batched device path for the same four circuits is
# `qmps_opt_env_objective` (EnergyEngine.opt_env_objective, SparseFullEnergyOptimizer(optimize_environment=True)).
def gate(v, symbol='U'):
    from .represent import ShallowFullStateTensor
    return ShallowFullStateTensor(2, v, symbol)


def op_state(params, which='energy'):
    """|psi> of one of the four circuits (30 params = [U angles | V angles]): 'energy' (4 qubits: V, U, U),
    'v_purity' (two copies of V + SWAP), 'u_purity' (two copies of U.V + two SWAPs), 'uv_purity' (5 qubits)."""
    from .represent import SWAP, final_state, line_qubits
    params = np.asarray(params, dtype=float)
    assert len(params) == 30
    p2, p1 = np.split(params, 2)
    if which == 'energy':
        q = line_qubits(4)
        return final_state([gate(p1)(*q[2:]), gate(p2)(*q[1:3]), gate(p2)(*q[:2])], 4)
    if which == 'v_purity':                       # 2 x 2 grid, row-major = qubits 0..3
        q = line_qubits(4)
        return final_state([gate(p1)(*q[0:2]), gate(p1)(*q[2:4]), SWAP(*q[0:2])], 4)
    if which == 'u_purity':                       # 2 x 3 grid, row-major = qubits 0..5
        q = line_qubits(6)
        return final_state([gate(p1)(*q[1:3]), gate(p2)(*q[0:2]), gate(p1)(*q[4:6]), gate(p2)(*q[3:5]),
                            SWAP(*q[0:2]), SWAP(*q[1:3])], 6)
    if which == 'uv_purity':
        q = line_qubits(5)
        return final_state([gate(p1)(*q[3:]), gate(p2)(*q[2:4]), gate(p1)(*q[:2]), SWAP(*q[:2])], 5)
    raise ValueError(which)


def op_H(H):
    """1 x H x 1 on the four qubits of the 'energy' circuit."""
    return np.kron(np.kron(np.eye(2), np.asarray(H)), np.eye(2))


def evo_Hs(D=2):
    return np.diag(np.eye(2 ** 6)[0]), np.diag(np.eye(2 ** 4)[0])


def swapper():
    from .ground_state import swap
    return -np.kron(np.kron(np.eye(2), swap()), np.eye(8))
