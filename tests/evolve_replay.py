"""Host replays of the device-resident time-evolution drivers with the ORACLE as evaluator (test infrastructure).

`replay_rotosolve` restates qmps_evolve_rotosolve step by step - reference tensors per time step
(qmps/new_time_evolve.py:281-283), per parameter the shifted candidates and the closed-form update of
qmps/rotosolve.py:175-177 (nsh = 3) or the six-sample fit of qmps/tools.py:434-452 (nsh = 6), per sweep the objective of the
updated vectors - with `oracle.overlap_eta` (numpy dense eig of the D^2 x D^2 mixed transfer matrix) evaluating every
candidate.  Nothing here touches the GPU."""
import numpy as np

from oracle import qmps_oracle as O

SHIFTS3 = (0.0, np.pi / 2, -np.pi / 2)


def builder(kind):
    return {0: O.shallow_cnot_unitary, 1: O.shallow_qaoa_unitary, 2: lambda D, p: O.shallow_full_unitary(p),
            3: O.shallow_cnot3_unitary, 4: O.shallow_cnot_nonuniform_unitary, 5: O.exact_after4_unitary,
            6: lambda D, p: O.state_gate_unitary(p)}[kind]


def tensor(kind, D, p):
    return O.unitary_to_tensor(builder(kind)(D, p))


def objective(kind, D, A, p, WW, want_gap=False, arpack=False):
    B = tensor(kind, D, p)
    if arpack:      # the reference's own solver (scipy eigs through xmps): ~3 ms instead of ~100 ms at D = 16; nearby candidates only
        return -np.sqrt(abs(O.overlap_eta_arpack(A, B, WW)[0]))
    # the eigenvalues of the matrix oracle.overlap_eta diagonalises (no eigenvectors: half the time at D = 16)
    C = np.tensordot(WW, O.merge(A, A), [1, 0])
    w = np.sort(np.abs(np.linalg.eigvals(O.transfer_matrix(C, O.merge(B, B)))))[::-1]
    return (-np.sqrt(w[0]), w[1] / w[0]) if want_gap else -np.sqrt(w[0])


def replay_rotosolve(kind, D, params, WW, n_steps, n_sweeps, nsh=3, gaps=None, global_argmin=False):
    """params (T, P) -> (params_hist (n_steps, T, P), f_hist (n_steps, n_sweeps, T)).
    gaps (optional list): receives |eta_2/eta_1| of every candidate evaluated (how hard the power method has it)."""
    X = np.array(params, dtype=float)
    T, P = X.shape
    shifts = SHIFTS3 if nsh == 3 else O.ROTO_SHIFTS
    ph, fh = np.empty((n_steps, T, P)), np.empty((n_steps, n_sweeps, T))

    def f(A, p):
        if gaps is None:
            return objective(kind, D, A, p, WW)
        v, g = objective(kind, D, A, p, WW, want_gap=True)
        gaps.append(g)
        return v
    for step in range(n_steps):
        A = [tensor(kind, D, X[t]) for t in range(T)]
        for sw in range(n_sweeps):
            for i in range(P):
                for t in range(T):
                    e = []
                    for s in shifts:
                        q = X[t].copy()
                        q[i] += s
                        e.append(f(A[t], q))
                    if nsh == 3:
                        X[t, i] = np.arctan2(np.sin(X[t, i] + O.rotosolve_update(*e)), np.cos(X[t, i] + O.rotosolve_update(*e)))
                    else:
                        X[t, i] += (O.double_sinusoid_argmin if global_argmin else O.double_sinusoid_fminbound)(*O.double_sinusoid_coefficients(*e))
            for t in range(T):
                fh[step, sw, t] = f(A[t], X[t])
        ph[step] = X
    return ph, fh


def spectral_radius(E, squarings=48):
    """max |eigenvalue| of a small dense matrix by Gelfand's formula on normalised squarings, ||E^(2^k)||^(1/2^k): good to ~1e-13 where an
    eigenvalue solver is not - a MULTIPLE dominant eigenvalue (numpy's eigvals: eps^(1/m) of its modulus on a Jordan block) or a nilpotent
    matrix (collapses to exactly 0).  Test infrastructure for the maps of the special grid."""
    X = np.array(E, dtype=complex)
    log_rho = 0.0
    for k in range(squarings + 1):
        n = np.linalg.norm(X)
        if not n > 1e-280 or (k > 0 and n < 1e-13):      # (the square of a unit-norm matrix at rounding level: a nilpotent matrix's collapse)
            return 0.0
        log_rho += np.log(n) / 2.0 ** k
        X = X / n
        X = X @ X
    return float(np.exp(log_rho))


def objective_gelfand(kind, D, A, p, WW):
    """-sqrt(spectral radius) of the mixed transfer matrix oracle.overlap_eta diagonalises (see `objective`)."""
    B = tensor(kind, D, p)
    C = np.tensordot(WW, O.merge(A, A), [1, 0])
    return -np.sqrt(spectral_radius(O.transfer_matrix(C, O.merge(B, B))))
