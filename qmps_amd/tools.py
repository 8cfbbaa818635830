"""Host-side mirror of the reference's `qmps/tools.py` for the hot path.

Same names, argument meaning and error behaviour as the reference (fergusfinn/qmps), with the
environment solve running on the MI355X:

  unitary_to_tensor        qmps/tools.py:151-154
  tensor_to_unitary        qmps/tools.py:123-148
  unitary_extension        qmps/tools.py:76-94
  environment_to_unitary   qmps/tools.py:97-108
  environment_from_unitary qmps/tools.py:111-120
  get_env_exact            qmps/tools.py:176-182   (eigen-solve -> libqmps_hip power iteration)
  Optimizer                qmps/tools.py:203-284
  double_rotosolve         qmps/tools.py:422-457
  RotosolveResult          qmps/tools.py:459-464

No cirq, no xmps: gates are plain unitary-producing objects (see represent.py).
"""
import numpy as np
from scipy.linalg import cholesky, null_space
from scipy.optimize import minimize, minimize_scalar

from . import _runtime

__all__ = ['unitary_to_tensor', 'tensor_to_unitary', 'unitary_extension', 'environment_to_unitary',
           'environment_from_unitary', 'get_env_exact', 'get_env_exact_alternative', 'right_environment',
           'Optimizer', 'OptimizerCircuit', 'double_rotosolve', 'RotosolveResult', 'random_unitary',
           'haar_unitary', 'cT', 'direct_sum', 'eye_like', 'svals', 'from_real_vector', 'to_real_vector',
           'split_2s', 'split_3s', 'split_ns']


# ---------------------------------------------------------------------------------------------
# small helpers (tools.py:36-73, 159-174)
# ---------------------------------------------------------------------------------------------
def random_unitary(*shape):
    """Q factor of a REAL Gaussian matrix, as in the reference (tools.py:36-37)."""
    return np.linalg.qr(np.random.randn(*shape))[0]


def haar_unitary(n, rng=None):
    """Complex Haar-like draw qr(randn + i randn)[0] (qmps/ansatze.py:30)."""
    rng = np.random.default_rng() if rng is None else rng
    return np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]


def svals(A):
    return np.linalg.svd(A, compute_uv=False)


def from_real_vector(v):
    """[re..., im...] -> complex vector."""
    v = np.asarray(v)
    half = v.shape[0] // 2
    return v[:half] + 1j * v[half:]


def to_real_vector(A):
    A = np.asarray(A)
    return np.concatenate([A.real.reshape(-1), A.imag.reshape(-1)])


def eye_like(A):
    return np.eye(A.shape[0])


def cT(tensor):
    """Hermitian conjugate over the last two indices."""
    return np.conj(np.swapaxes(tensor, -1, -2))


def direct_sum(A, B):
    out = np.zeros((A.shape[0] + B.shape[0], A.shape[1] + B.shape[1]), dtype=np.result_type(A, B))
    out[:A.shape[0], :A.shape[1]] = A
    out[A.shape[0]:, A.shape[1]:] = B
    return out


def _chunks(x, n):
    return [x[i:i + n] for i in range(0, len(x), n)]


def split_2s(x):
    return _chunks(x, 2)


def split_3s(x):
    return _chunks(x, 3)


def split_ns(x, n):
    return _chunks(x, n)


# ---------------------------------------------------------------------------------------------
# tensor <-> unitary embeddings
# ---------------------------------------------------------------------------------------------
def unitary_to_tensor(U):
    """A[s, i, j] = U[2 i + s, j] for j < D: the first input qubit is |0>, the last output qubit
    is the physical index (tools.py:151-154)."""
    U = np.asarray(U)
    N = U.shape[0]
    D = N // 2
    return np.ascontiguousarray(U[:, :D].reshape(D, 2, D).transpose(1, 0, 2))


def unitary_extension(Q, D=None):
    """Complete an isometry to a unitary with null-space columns/rows (tools.py:76-94)."""
    Q = np.asarray(Q)
    rows, cols = Q.shape
    if rows > cols:
        out = np.concatenate([Q, null_space(Q.conj().T)], axis=1)
    elif rows < cols:
        out = np.concatenate([Q.conj().T, null_space(Q)], axis=1).conj().T
    else:
        out = Q
    if D is not None and D > out.shape[0]:
        out = direct_sum(out, np.eye(D - out.shape[0]))
    return out


def tensor_to_unitary(A, testing=False):
    """Embed a left-isometric tensor A[s,i,j] in a unitary whose first D columns are
    iso[(i,s), j] (tools.py:123-148).  With testing=True also returns the reference's five checks."""
    d, D, _ = A.shape
    iso = np.asarray(A).transpose(1, 0, 2).reshape(D * d, D)
    U = unitary_extension(iso)
    if testing:
        n = U.shape[0]
        ok = (np.allclose(iso.conj().T @ iso, np.eye(D)) and np.allclose(U @ U.conj().T, np.eye(n))
              and np.allclose(U.conj().T @ U, np.eye(n)) and np.allclose(U[:, :D], iso)
              and np.allclose(unitary_to_tensor(U), A))
        return U, bool(ok)
    return U


def environment_to_unitary(v):
    """vec(v)/||v|| becomes the first column of a unitary, completed by a null space
    (tools.py:97-108).  Only that first column ever reaches the energy."""
    row = np.asarray(v).reshape(1, -1)
    row = row / np.linalg.norm(row)
    rest = null_space(row).conj().T
    return np.concatenate([row, rest], axis=0).T


def environment_from_unitary(u):
    """First column of u as a D x D matrix (tools.py:111-120; the reference hard-codes D = 2)."""
    u = np.asarray(u)
    D = int(round(np.sqrt(u.shape[0])))
    return u[:, 0].reshape(D, D)


# ---------------------------------------------------------------------------------------------
# exact environment on the GPU
# ---------------------------------------------------------------------------------------------
def right_environment(U, max_iter=10000, tol=1e-13):
    """Dominant right eigen-matrix r (Hermitian, tr r = 1) of the transfer map of
    unitary_to_tensor(U) - what `TransferMatrix(A).eigs()` returns at tools.py:181, up to xmps's
    normalisation - by the libqmps_hip power iteration.  Raises LinAlgError if it did not converge."""
    U = np.asarray(U, dtype=np.complex128)
    D = U.shape[0] // 2
    eng = _runtime.engine(D, 1)
    r, it, st = eng.env_batch(U[None], kind='unitary', max_iter=max_iter, tol=tol)
    if st[0] == 1:
        raise np.linalg.LinAlgError(f'right environment did not converge in {max_iter} power iterations')
    return r[0]


def get_env_exact(U):
    """V = environment_to_unitary(cholesky(r)^dagger) (tools.py:176-182).  Raises
    numpy.linalg.LinAlgError when r is not positive definite, like scipy's cholesky in the reference."""
    r = right_environment(U)
    return environment_to_unitary(cholesky(r).conj().T)


def get_env_exact_alternative(U):
    """tools.py:184-186 builds V from the mixed-canonical centre matrix C (r = C C^dagger); any
    square root of r gives the same energy, so the Hermitian square root is used here."""
    r = right_environment(U)
    w, v = np.linalg.eigh(r)
    if w.min() <= 0:
        raise np.linalg.LinAlgError('environment is not positive definite')
    return environment_to_unitary((v * np.sqrt(w)) @ v.conj().T)


# ---------------------------------------------------------------------------------------------
# Optimizer base class (tools.py:196-284) and rotosolve (tools.py:422-464)
# ---------------------------------------------------------------------------------------------
class OptimizerCircuit:
    def __init__(self, circuit=None, total_qubits=None, aux_qubits=None):
        self.circuit = circuit
        self.total_qubits = total_qubits
        self.aux_qubits = aux_qubits
        self.qubits = None


class RotosolveResult(object):
    def __init__(self, history, fun, x, message):
        self.history = history
        self.fun = fun
        self.x = x
        self.message = message


def _double_sinusoid_shift(M0, Mpi, Mp2, Mm2, Mp4, Mm4):
    """Fit P sin(2x+u) + Q sin(x+v) through the six samples and return its wrapped minimiser
    (tools.py:434-452)."""
    A, Bv = M0 + Mpi, M0 - Mpi
    C, Dv = Mp2 + Mm2, Mp2 - Mm2
    E = Mp4 - Mm4
    a, b = 0.25 * (2 * E - np.sqrt(2) * Dv), 0.25 * (A - C)
    c, d = 0.5 * Dv, 0.5 * Bv
    P, u = np.hypot(a, b), np.arctan2(b, a)
    Q, v = np.hypot(c, d), np.arctan2(d, c)
    th = minimize_scalar(lambda x: P * np.sin(2 * x + u) + Q * np.sin(x + v), bounds=[-np.pi, np.pi]).x
    return np.arctan2(np.sin(th), np.cos(th))


ROTO_SHIFTS = np.array([0.0, np.pi, np.pi / 2, -np.pi / 2, np.pi / 4, -np.pi / 4])


def double_rotosolve(eps, initial_parameters, N_iters=100, disp=True, batch_eps=None):
    """Double-frequency rotosolve (tools.py:422-457): per parameter, sample the objective at the
    shifts {0, pi, +-pi/2, +-pi/4}, fit, move to the minimiser.  Updates `initial_parameters`
    in place like the reference.  If `batch_eps` (params[B,P] -> float[B]) is given, the six
    samples of a parameter are ONE launch on the GPU instead of ten scalar calls."""
    params = initial_parameters
    n = len(params)
    history = []
    for w in range(N_iters):
        if disp:
            print(w, ', ', sep='', end='', flush=True)
        for i in range(n):
            if batch_eps is not None:
                P = np.repeat(np.asarray(params, dtype=float)[None], 6, axis=0)
                P[:, i] += ROTO_SHIFTS
                M = np.asarray(batch_eps(P), dtype=float).reshape(6, -1).sum(1)
            else:
                def one(x):
                    q = np.array(params, dtype=float)
                    q[i] += x
                    return np.sum(eps(q))
                M = [one(x) for x in ROTO_SHIFTS]
            # a sample without a valid environment (NaN from the batched objective: product / GHZ points, which the
            # +-pi/2 shifts do hit) leaves the parameter where it is - what the device kernels do, and the nearest
            # batch analogue of the scalar objective's "return the previous value" (ground_state.py:153-157)
            if np.all(np.isfinite(M)):
                params[i] += _double_sinusoid_shift(*M)
        if disp:
            print('\n', sep='', end='', flush=True)
        history.append(eps(params))
    return RotosolveResult(history, history[-1], params, '')


def batched_fd_gradient(batch_fun, X, h=1e-6, F0=None):
    """Central-difference gradients of T independent objectives in ONE batched evaluation: the "finite-difference
    columns" of SURVEY section 3 as a batch.  batch_fun: (N, P) -> (N,), rows independent; X (T, P).
    Returns (f (T,), g (T, P)); the candidates are trajectory-major, row t (2P + 1) + k."""
    X = np.atleast_2d(np.asarray(X, dtype=float))
    T, P = X.shape
    I = np.eye(P)
    cand = np.concatenate([X[:, None, :], X[:, None, :] + h * I[None], X[:, None, :] - h * I[None]], axis=1)
    F = np.asarray(batch_fun(cand.reshape(-1, P))).reshape(T, 2 * P + 1)
    return F[:, 0], (F[:, 1:P + 1] - F[:, P + 1:]) / (2 * h)


def batched_bfgs(grad_batch, line_batch, X0, maxiter=200, gtol=1e-5, h=1e-6, c1=1e-4,
                 alphas=(1.0, 0.5, 0.25, 0.125, 1 / 16, 1 / 64, 1 / 256, 1 / 4096), value_and_grad=None, first_rungs=None, Hinv0=None, speculative=False, on_active=None):
    """T independent BFGS minimisations in LOCK-STEP (scipy's BFGS is what the reference's time-evolution loop runs per
    step: `minimize(obj, params, (A_, WW))`, new_time_evolve.py:284 / scripts/loschmidt.py:371 - one trajectory, one
    scalar objective call at a time).  Here every iteration is two batched evaluations over all trajectories:
      grad_batch  (T (2P+1), P) -> values: the iterates and their 2P central-difference neighbours (step h),
      line_batch  (T len(alphas), P) -> values: the backtracking ladder x + alpha d; the first alpha with the Armijo
                  decrease f(x + alpha d) <= f + c1 alpha g.d is taken (none: the best of the ladder if it decreases f,
                  else the trajectory stops).
    Rows are trajectory-major (what `qmps_overlap_set_group` expects).  Inverse-Hessian update: the BFGS formula, skipped
    when s.y <= 1e-12 |s||y|.  A trajectory is converged when max|g| < gtol and then stays put (its rows are still
    evaluated: the batch shape never changes, so resident warm starts keep their slots).
    value_and_grad (optional): X (T,P) -> (f (T,), g (T,P)) replaces the central-difference batches (the device computes the same
    central differences from one pair of eigen-solves per iterate: qmps_overlap_gradient).
    first_rungs (optional int n < len(alphas)): the ladder is evaluated in two stages - line_batch receives T n candidates (the first
    n step lengths) and, only if some trajectory finds no acceptable step among them, a second batch of T (len(alphas) - n) - near
    the minimum BFGS accepts alpha = 1 almost always.  line_batch must then accept both group sizes.
    speculative (needs value_and_grad): objective AND gradient are evaluated at the full step x + alphas[0] d straight away;
    if every active trajectory accepts that step (Armijo) - the normal case of a quasi-Newton iteration - the iteration is that
    ONE batch; otherwise the remaining rungs of the ladder are evaluated (line_batch receives T (len(alphas) - 1) candidates) and
    the gradient at the accepted points - both masked (on_active) to the trajectories that rejected the full step.  Same decisions as the plain ladder: the first rung is tested first either way.
    on_active (optional): called as on_active('grad' | 'line', active (T,) bool) right before every batch of the loop - an evaluator
    that can skip trajectories (qmps_overlap_set_active) then spends nothing on the converged ones; rows of skipped trajectories
    may come back with stale values: they are never used.
    Hinv0 (optional, (T,P,P)): initial inverse Hessians instead of the identity (scipy's start) - e.g. the ones the previous time
    step of the same trajectories ended with; the result carries the final ones as 'hess_inv'.
    Returns dict(x (T,P), fun (T,), jac (T,P), nit, nfev, converged (T,), history [fun per iteration], hess_inv (T,P,P))."""
    X = np.array(np.atleast_2d(X0), dtype=float)
    T, P = X.shape
    al = np.asarray(alphas, dtype=float)
    Hinv = np.tile(np.eye(P), (T, 1, 1)) if Hinv0 is None else np.array(Hinv0, dtype=float, copy=True)
    vg = value_and_grad if value_and_grad is not None else (lambda Z: batched_fd_gradient(grad_batch, Z, h))
    f, g = vg(X)
    nfev = T * (2 * P + 1)
    active = np.abs(g).max(axis=1) >= gtol
    history = [f.copy()]
    nit = 0
    while nit < maxiter and active.any():
        d = -np.einsum('tij,tj->ti', Hinv, g)
        slope = np.einsum('ti,ti->t', g, d)
        bad = ~(slope < 0)
        if bad.any():                                   # not a descent direction: restart from steepest descent
            Hinv[bad] = np.eye(P)
            d[bad] = -g[bad]
            slope[bad] = -np.einsum('ti,ti->t', g[bad], g[bad])
        d[~active] = 0.0
        def ladder(a, need=None):
            if on_active is not None:
                on_active('line', active if need is None else need)
            cand = X[:, None, :] + a[None, :, None] * d[:, None, :]
            F = np.asarray(line_batch(cand.reshape(-1, P))).reshape(T, len(a))
            return np.where(np.isfinite(F), F, np.inf)
        fn = gn = None
        if speculative and value_and_grad is not None:
            if on_active is not None:
                on_active('grad', active)
            fs, gs = vg(X + al[0] * d)
            fs, gs = np.where(active, fs, f), np.where(active[:, None], gs, g)      # (rows of skipped trajectories: their last values)
            nfev += T * (2 * P + 1)
            Fc = np.full((T, len(al)), np.inf)
            Fc[:, 0] = np.where(np.isfinite(fs), fs, np.inf)
            took = Fc[:, 0] <= f + c1 * al[0] * slope
            if took[active].all():
                fn, gn = fs, gs                          # every active trajectory takes the full step: nothing else to evaluate
            else:
                # the ladder and the gradient at the accepted point only for the trajectories that rejected the full step (the
                # others keep what the speculative batch gave them: their accepted point IS x + alphas[0] d)
                need = active & ~took
                Fc[:, 1:] = np.where(need[:, None], ladder(al[1:], need), np.inf)
                nfev += T * (len(al) - 1)
        elif first_rungs:
            Fc = np.full((T, len(al)), np.inf)
            Fc[:, :first_rungs] = ladder(al[:first_rungs])
            nfev += T * first_rungs
            if not (Fc <= f[:, None] + c1 * al[None, :] * slope[:, None]).any(axis=1)[active].all():
                Fc[:, first_rungs:] = ladder(al[first_rungs:])
                nfev += T * (len(al) - first_rungs)
        else:
            Fc = ladder(al)
            nfev += T * len(al)
        ok = Fc <= f[:, None] + c1 * al[None, :] * slope[:, None]
        first = np.where(ok.any(axis=1), ok.argmax(axis=1), Fc.argmin(axis=1))
        fa = Fc[np.arange(T), first]
        moved = active & (fa < f)
        a = np.where(moved, al[first], 0.0)
        s = a[:, None] * d
        Xn = X + s
        if fn is None:
            redo = active if not (speculative and value_and_grad is not None) else need
            if on_active is not None:
                on_active('grad', redo)
            fn, gn = vg(Xn)
            if redo is not active:
                fn, gn = np.where(redo, fn, fs), np.where(redo[:, None], gn, gs)
            nfev += T * (2 * P + 1)
        y = gn - g
        sy = np.einsum('ti,ti->t', s, y)
        upd = moved & (sy > 1e-12 * np.sqrt(np.einsum('ti,ti->t', s, s) * np.einsum('ti,ti->t', y, y))) & (sy > 0)
        if upd.any():
            # BFGS update of the (symmetric) inverse Hessians, all trajectories at once, as rank-two corrections
            #   H' = (1 - rho s y^T) H (1 - rho y s^T) + rho s s^T = H - rho (s (Hy)^T + (Hy) s^T) + rho (1 + rho y^T H y) s s^T
            # (einsum outer products: numpy's batched matmul of 8 x 8 matrices costs 100 us per product at T = 256)
            rho = np.where(upd, 1.0 / np.where(upd, sy, 1.0), 0.0)
            Hy = np.einsum('tij,tj->ti', Hinv, y)
            a = rho * (1.0 + rho * np.einsum('ti,ti->t', y, Hy))
            c = np.einsum('ti,tj->tij', rho[:, None] * s, Hy)      # (two-operand einsums: the three-operand form is 5x slower)
            Hinv = Hinv - (c + c.transpose(0, 2, 1)) + np.einsum('ti,tj->tij', a[:, None] * s, s)
        X, f, g = Xn, np.where(moved, fn, f), np.where(moved[:, None], gn, g)
        active = active & moved & (np.abs(g).max(axis=1) >= gtol)
        history.append(f.copy())
        nit += 1
    return {'x': X, 'fun': f, 'jac': g, 'nit': nit, 'nfev': nfev, 'converged': np.abs(g).max(axis=1) < gtol,
            'history': np.array(history), 'hess_inv': Hinv}


def batched_nelder_mead(batch_fun, x0, xatol=1e-4, fatol=1e-4, maxiter=None, maxfev=None, callback=None, speculate=True):
    """scipy's Nelder-Mead (the reference's DEFAULT optimiser: settings['method'], qmps/tools.py:212-219) with its objective
    evaluations issued as batches: the initial simplex (N + 1 points) and every shrink step (N points) are one batched call
    each, and - speculate=True - the four candidate points of an iteration (reflection, expansion, outside and inside
    contraction) are evaluated together, so an iteration is ONE device launch instead of up to three dependent scalar calls.
    The decisions are scipy's (`_minimize_neldermead`, rho = 1, chi = 2, psi = sigma = 1/2, initial simplex x0 (1 + 0.05 e_k),
    same termination test), taken on the same values: the simplex sequence is the one the scalar path produces.
    batch_fun: (n, N) -> (n,); NaN (no valid environment) counts as +inf.  Returns a scipy OptimizeResult; `nfev` counts the
    evaluations scipy would have made, `nfev_batched` the points actually evaluated, `n_batches` the device launches."""
    from scipy.optimize import OptimizeResult
    x0 = np.asarray(x0, dtype=float).ravel()
    N = len(x0)
    rho, chi, psi, sigma = 1.0, 2.0, 0.5, 0.5
    if maxiter is None and maxfev is None:          # scipy's defaults, case by case
        maxiter = maxfev = N * 200
    elif maxiter is None:
        maxiter = N * 200 if maxfev == np.inf else np.inf
    elif maxfev is None:
        maxfev = N * 200 if maxiter == np.inf else np.inf
    stats = {'batches': 0, 'points': 0}

    def F(X):
        stats['batches'] += 1
        stats['points'] += len(X)
        v = np.asarray(batch_fun(np.atleast_2d(X)), dtype=float).ravel()
        return np.where(np.isfinite(v), v, np.inf)
    sim = np.empty((N + 1, N))
    sim[0] = x0
    for k in range(N):
        y = x0.copy()
        y[k] = (1 + 0.05) * y[k] if y[k] != 0 else 0.00025
        sim[k + 1] = y
    fsim = F(sim)
    fcalls = N + 1
    ind = np.argsort(fsim)
    sim, fsim = sim[ind], fsim[ind]
    iterations = 1
    while fcalls < maxfev and iterations < maxiter:
        if np.max(np.abs(sim[1:] - sim[0])) <= xatol and np.max(np.abs(fsim[0] - fsim[1:])) <= fatol:
            break
        xbar = np.add.reduce(sim[:-1], 0) / N
        xr = (1 + rho) * xbar - rho * sim[-1]
        xe = (1 + rho * chi) * xbar - rho * chi * sim[-1]
        xc = (1 + psi * rho) * xbar - psi * rho * sim[-1]
        xcc = (1 - psi) * xbar + psi * sim[-1]
        if speculate:
            fxr, fxe, fxc, fxcc = F(np.stack([xr, xe, xc, xcc]))
        else:
            fxr = F(xr)[0]
        fcalls += 1
        doshrink = False
        if fxr < fsim[0]:
            if not speculate:
                fxe = F(xe)[0]
            fcalls += 1
            if fxe < fxr:
                sim[-1], fsim[-1] = xe, fxe
            else:
                sim[-1], fsim[-1] = xr, fxr
        elif fxr < fsim[-2]:
            sim[-1], fsim[-1] = xr, fxr
        elif fxr < fsim[-1]:
            if not speculate:
                fxc = F(xc)[0]
            fcalls += 1
            if fxc <= fxr:
                sim[-1], fsim[-1] = xc, fxc
            else:
                doshrink = True
        else:
            if not speculate:
                fxcc = F(xcc)[0]
            fcalls += 1
            if fxcc < fsim[-1]:
                sim[-1], fsim[-1] = xcc, fxcc
            else:
                doshrink = True
        if doshrink:
            sim[1:] = sim[0] + sigma * (sim[1:] - sim[0])
            fsim[1:] = F(sim[1:])
            fcalls += N
        iterations += 1
        ind = np.argsort(fsim)
        sim, fsim = sim[ind], fsim[ind]
        if callback is not None:
            callback(sim[0])
    x, fval = sim[0], float(np.min(fsim))
    if fcalls >= maxfev:
        status, msg = 1, 'Maximum number of function evaluations has been exceeded.'
    elif iterations >= maxiter:
        status, msg = 2, 'Maximum number of iterations has been exceeded.'
    else:
        status, msg = 0, 'Optimization terminated successfully.'
    return OptimizeResult(fun=fval, nit=iterations, nfev=fcalls, status=status, success=status == 0, message=msg, x=x,
                          final_simplex=(sim, fsim), nfev_batched=stats['points'], n_batches=stats['batches'])


class Optimizer:
    """Same contract as tools.py:203-284: subclasses bind/override `objective_function(params) ->
    float`; `optimize()` dispatches on settings['method'] ('Rotosolve' -> double_rotosolve,
    otherwise scipy.optimize.minimize) and then calls `update_state()`."""

    def __init__(self, u=None, v=None, initial_guess=None, obj_fun=None, args=None):
        self.u = u
        self.v = v
        self.initial_guess = initial_guess
        self.iters = 0
        self.optimized_result = None
        self.obj_fun_values = []
        self.settings = {'maxiter': 10000, 'verbose': True, 'method': 'Nelder-Mead', 'tol': 1e-8,
                         'store_values': True, 'bayesian': False}
        self.is_verbose = self.settings['verbose']
        self.obj_fun = obj_fun
        self.args = args
        self.circuit = OptimizerCircuit()

    def change_settings(self, new_settings):
        return self.settings.update(new_settings)

    def gate_from_params(self, params):
        pass

    def update_state(self):
        pass

    def callback_store_values(self, xk):
        val = self.objective_function(xk)
        self.obj_fun_values.append(val)
        if self.settings['verbose']:
            print(f'{self.iters}:{val}')
        self.iters += 1

    def objective_function(self, params):
        if self.obj_fun is not None:
            return self.obj_fun(params, *(self.args or ()))

    def batch_objective_function(self, params_batch):
        """Batched objective; subclasses with a GPU path override it."""
        return np.array([self.objective_function(p) for p in params_batch])

    def optimize_restarts(self, initial_guesses, method=None, maxiter=None, tol=None):
        """R restarts of this optimisation in lock-step over device batches: `tools.optimize_restarts`."""
        return optimize_restarts(self, initial_guesses, method=method, maxiter=maxiter, tol=tol)

    def optimize(self):
        s = self.settings
        verbose = s['verbose']
        if s['bayesian']:
            raise NotImplementedError('bayesian (skopt) optimisation is outside the hot path')
        if s['method'] == 'Rotosolve':
            device = getattr(self, '_device_double_rotosolve', None)
            res = device(s['maxiter']) if device is not None else None
            if res is None:
                batch = self.batch_objective_function if type(self).batch_objective_function \
                    is not Optimizer.batch_objective_function else None
                res = double_rotosolve(self.objective_function, self.initial_guess, s['maxiter'], verbose, batch_eps=batch)
            self.optimized_result = res
        else:
            batch = self.batch_objective_function if type(self).batch_objective_function \
                is not Optimizer.batch_objective_function else None
            cb = self.callback_store_values if s['store_values'] else None
            if batch is not None and s.get('batched', True) and s['method'] == 'Nelder-Mead':
                # the reference's default method with its simplex points evaluated as device batches (same decisions, same simplex)
                self.optimized_result = batched_nelder_mead(batch, self.initial_guess, xatol=s['tol'], fatol=s['tol'],
                                                            maxiter=s['maxiter'], callback=cb)
                if verbose:
                    print(self.optimized_result.message)
            elif batch is not None and s.get('batched', True) and s['method'] in ('BFGS', 'L-BFGS-B', 'CG', 'SLSQP', 'TNC'):
                # gradient methods: scipy differentiates by one scalar call per column; here the 2 P central-difference
                # neighbours of an iterate are ONE batch (the "finite-difference columns" of SURVEY section 3)
                def jac(x):
                    return batched_fd_gradient(batch, np.asarray(x, dtype=float)[None], h=1e-6)[1][0]
                self.optimized_result = minimize(fun=self.objective_function, x0=self.initial_guess, method=s['method'], jac=jac,
                                                 tol=s['tol'], options={'maxiter': s['maxiter'], 'disp': verbose}, callback=cb)
            else:
                self.optimized_result = minimize(fun=self.objective_function, x0=self.initial_guess, method=s['method'],
                                                 tol=s['tol'], options={'maxiter': s['maxiter'], 'disp': verbose}, callback=cb)
        self.update_state()
        if verbose:
            print(f'Reason for termination is {self.optimized_result.message} ' +
                  f'\nObjective Function Value is {self.optimized_result.fun}')
        return self.optimized_result


def _restart_result(x, fun, nit, nfev, converged, history):
    from scipy.optimize import OptimizeResult
    return OptimizeResult(x=np.array(x), fun=float(fun), nit=int(nit), nfev=int(nfev), success=bool(converged), status=0 if converged else 1,
                          message='converged' if converged else 'stopped', history=np.asarray(history))


def optimize_restarts(optimizer, initial_guesses, method=None, maxiter=None, tol=None):
    """R independent optimisations of `optimizer`'s objective from R starting points, as ONE lock-step over device batches.

    The reference's drivers loop over random restarts and run one scalar optimisation after the other, every evaluation a cirq
    simulation (`scripts/ground_state_finding.py:137-154, 173-195`: `minimize(eps, randn(n), method='BFGS')` until the energy stops
    improving; `scripts/noisy_optimization.py:46-72`: `opt.optimize()` again from a fresh `randn`).  The restarts are independent, so
    here they advance together: method 'BFGS' = `batched_bfgs` (every iteration two batches over all restarts: the iterates with their
    2 P central-difference neighbours, the backtracking ladder), 'Rotosolve' = the device driver with R restarts in one C call
    (`qmps_double_rotosolve`; gate classes without a device kernel: the host driver per restart), anything else: `optimizer.optimize()`
    per restart (Nelder-Mead takes sequential decisions per simplex; its candidate points are already batched).
    method / maxiter / tol default to `optimizer.settings`.  Returns the list of R results (scipy `OptimizeResult`s: x, fun, nit,
    success, history) in the order of the guesses; `optimizer.optimized_result` is the best of them (lowest finite fun) and
    `optimizer.restart_results` the list - then `update_state()` as after `optimize()`."""
    s = optimizer.settings
    method = s['method'] if method is None else method
    maxiter = s['maxiter'] if maxiter is None else maxiter
    tol = s['tol'] if tol is None else tol
    X0 = np.array(np.atleast_2d(initial_guesses), dtype=float)
    R = X0.shape[0]
    batch = optimizer.batch_objective_function
    results = None
    if method == 'BFGS':
        res = batched_bfgs(batch, batch, X0, maxiter=maxiter, gtol=tol)
        hist = res['history']
        results = [_restart_result(res['x'][r], res['fun'][r], res['nit'], res['nfev'] // R, res['converged'][r], hist[:, r]) for r in range(R)]
    elif method == 'Rotosolve' and getattr(optimizer, '_device_double_rotosolve', None) is not None:
        kind = getattr(getattr(optimizer, 'state_tensor', None), 'device_kind', None)
        if not getattr(optimizer, 'optimize_environment', False) and kind is not None and kind <= 3 and not (kind == 2 and optimizer.D != 2):
            from .rotosolve import device_double_rotosolve
            es, P = device_double_rotosolve(optimizer, X0, maxiter)
            results = [_restart_result(P[r], es[-1, r], maxiter, 6 * maxiter * X0.shape[1], True, es[:, r]) for r in range(R)]
    if results is None:
        results = []
        keep = optimizer.initial_guess
        for r in range(R):
            optimizer.initial_guess = X0[r].copy()
            old = dict(optimizer.settings)
            optimizer.settings.update({'method': method, 'maxiter': maxiter, 'tol': tol})
            try:
                results.append(optimizer.optimize())
            finally:
                optimizer.settings.clear()
                optimizer.settings.update(old)
        optimizer.initial_guess = keep
    funs = np.array([r_.fun if np.isfinite(r_.fun) else np.inf for r_ in results])
    optimizer.restart_results = results
    optimizer.optimized_result = results[int(np.argmin(funs))]
    optimizer.update_state()
    return results


class GuessInitialFullParameterOptimizer(Optimizer):
    """Find U4 parameters reproducing a given two-qubit unitary `u` (tools.py:287-305).  The reference's 4-qubit
    circuit (two Bell pairs, u on one half, conj(U4(params)) on the other, un-Bell) has |0000>-amplitude
    tr(U4(params)^+ u) / 4, so the objective 1 - |amplitude|^2 is evaluated in closed form."""

    def objective_function(self, params):
        from .ground_state import U4
        from .represent import unitary
        u = self.u if isinstance(self.u, np.ndarray) else unitary(self.u)
        amp = np.trace(U4(params).conj().T @ u) / u.shape[0]
        return float(1 - abs(amp) ** 2)
