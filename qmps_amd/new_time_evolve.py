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
from .represent import ShallowFullStateTensor, build_gate, unitary
from .tools import unitary_to_tensor


def gate(v, symbol='U'):
    """The candidate state tensor's gate (new_time_evolve.py:186-187; scripts/loschmidt.py:203-207 uses
    ShallowCNOTStateTensor(2, v) instead - pass `state_tensor=` to the functions below for that)."""
    return ShallowFullStateTensor(2, v, symbol)


def _default_class(D):
    """new_time_evolve.py evolves ShallowFullStateTensor(2, .) (15 angles); scripts/loschmidt.py ShallowCNOTStateTensor(2, .),
    the only family with members at every bond dimension."""
    from .represent import ShallowCNOTStateTensor
    return ShallowFullStateTensor if D == 2 else ShallowCNOTStateTensor


def _n_angles(cls, p):
    return 15 if cls is ShallowFullStateTensor else len(p)


def state_tensor(p, D=2, state_tensor=None):
    """A(p) = unitary_to_tensor(unitary(gate(p))); already left-canonical (a unitary's first D columns)."""
    cls = state_tensor or _default_class(D)
    return unitary_to_tensor(unitary(build_gate(cls, D, np.asarray(p, dtype=float)[:_n_angles(cls, p)])))


def batch_obj(P, A, WW, return_eta=False, D=2, state_tensor=None, max_rounds=None, tol=1e-13):
    """-sqrt(|eta|) for every row of P (B, n_angles) against the current state A (2,D,D) [or one state per row,
    (B,2,D,D)]: one kernel launch.  Gate classes the library simulates (`device_kind`) are built on the device."""
    cls = state_tensor or _default_class(D)
    P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64)
    P = P[:, :_n_angles(cls, P[0])]
    eng = _runtime.engine(D, P.shape[0])
    kind = getattr(cls, 'device_kind', None)
    if kind is None or (kind in (L.ANSATZ_SHALLOW_FULL, L.ANSATZ_STATE_GATE) and D != 2):
        cand = np.stack([unitary_to_tensor(unitary(build_gate(cls, D, p))) for p in P])
        eta, rounds, st = eng.overlaps(A, cand, WW, kind='tensor', max_rounds=max_rounds, tol=tol)
    else:
        eta, rounds, st = eng.overlaps(A, P, WW, kind='params', ansatz=kind, max_rounds=max_rounds, tol=tol)
    f = -np.sqrt(np.abs(eta))
    f = np.where(L.overlap_usable(st), f, np.nan)          # (D = 2: a tie's common modulus, QMPS_STATUS_TIED, is a valid objective)
    return (f, eta) if return_eta else f


def obj(p, A, WW, D=None, state_tensor=None):
    """Scalar objective with the reference's signature `obj(p, A, WW)` (extra entries of p beyond the gate angles -
    the reference's unused `rs` - are ignored).  The bond dimension is read off A."""
    D = np.asarray(A).shape[-1] if D is None else D
    return float(batch_obj(np.asarray(p, dtype=float)[None], A, WW, D=D, state_tensor=state_tensor)[0])


class _GroupedObjective:
    """Batched objective of T trajectories against their own reference states, trajectory-major candidate batches of a
    FIXED group size G (candidate t G + k belongs to trajectory t): parameters -> tensors, overlap objective and
    -sqrt|eta| all on the device; the fixed points stay resident in the candidates' slots, so the next batch of the
    same shape - the next iteration's neighbours of a slightly moved iterate - starts warm (D = 8, 16)."""

    def __init__(self, D, kind, T, G, max_rounds, tol, device=0):
        from .engine import EnergyEngine
        self.eng = EnergyEngine(D, T * max(np.atleast_1d(G)), device=device)
        self.kind, self.T, self.G, self.max_rounds, self.tol = kind, T, G, max_rounds, tol
        self.warm = False
        self.grad_warm = False
        self.kernel_ms = None          # a list: receives the HIP-event duration of every launch's overlap kernel (bench.py)
        self._timing = False

    def set_reference(self, ref_params, WW):
        self.eng.overlap_set_refs_params(self.kind, ref_params, WW)

    def __call__(self, cand):
        """cand (T G, P), trajectory-major; G = the group size given to the constructor (or any of them, if several were given)."""
        cand = np.ascontiguousarray(cand, dtype=np.float64)
        self._want_timing()
        G = cand.shape[0] // self.T
        assert cand.shape[0] == self.T * G and G in np.atleast_1d(self.G)
        self.eng.overlap_set_group(G)
        # power-method bond dimensions: the fixed points stay in the candidates' slots (one slot layout: only while G is fixed)
        # the ladder's FIRST stage (or the only group size) owns the slots; a second-stage batch neither reads nor overwrites them
        keep = self.eng.D >= 8 and G == np.atleast_1d(self.G)[0]
        f, st = self.eng.overlap_eval_params(self.kind, cand, max_rounds=self.max_rounds, tol=self.tol, want_r=keep, warm=self.warm and keep)
        self.warm = self.warm or keep
        if self.kernel_ms is not None:
            self.kernel_ms.append(self.eng.kernel_time(1)[0])
        return np.where(L.overlap_usable(st), f, np.nan)

    def value_and_grad(self, X, h=1e-6):
        """(f (T,), g (T, P)) of the iterates X from one right + one left eigen-solve each (qmps_overlap_gradient), warm-started
        from the previous call's fixed points."""
        # (tight_gradient False: objective by the two-sided quotient, the two solves stop at 1e-8 - what qmps_evolve_bfgs does)
        self._want_timing()
        tight = getattr(self, 'tight_gradient', True)
        f, g, st = self.eng.overlap_gradient(self.kind, X, h=h, max_rounds=max(self.max_rounds, 100000), tol=self.tol if tight else max(self.tol, 1e-8),
                                             warm=self.grad_warm, two_sided_f=not tight)
        self.grad_warm = True
        if self.kernel_ms is not None:
            self.kernel_ms.append(self.eng.kernel_time(1)[0])
        bad = ~L.overlap_usable(st)
        return np.where(bad, np.nan, f), np.where(bad[:, None], np.nan, g)

    def _want_timing(self):
        if self.kernel_ms is not None and not self._timing:      # (event records are off by default: they cost the stream several us per launch)
            self.eng.set_kernel_timing_period(1)
            self._timing = True

    def close(self):
        self.eng.close()


def evolve(params, WW, n_steps, method='Nelder-Mead', options=None, callback=None, D=2, state_tensor=None,
           n_sweeps=4, max_rounds=None, tol=1e-12, return_info=False):
    """The reference's time-evolution loop (new_time_evolve.py:276-292, scripts/loschmidt.py:367-375): at each step the
    current tensor A = A(params) is fixed and the next parameters maximise the overlap with W . |A A>, starting from the
    current ones.  `params` (P,) - one trajectory, the reference's shape - or (T, P): T independent trajectories
    (BASELINE.json configs[4]) evolved in lock-step.  Any bond dimension D in {2, 4, 8, 16}; `state_tensor` = gate class
    (default: ShallowFullStateTensor at D = 2, ShallowCNOTStateTensor otherwise).
      method 'Rotosolve' / 'DoubleRotosolve': the WHOLE evolution - every step, sweep, parameter, trajectory - is one C
          call (`qmps_evolve_rotosolve`, n_sweeps sweeps per step), no host round trip.  HEURISTIC: -sqrt|eta| with the exact
          environment is not a sinusoid of a gate angle (the reference's rotosolve evolution, scripts/rotosolve.py:270-294, fits a
          variational environment for that reason), so a closed-form update can RAISE the objective and the trajectory may stop
          tracking W|AA>; a RuntimeWarning says so once per call when a step ends above where it started.  Use 'BFGS' for physics;
      method 'BFGS': lock-step batched BFGS - by default the whole evolution in ONE C call (`qmps_evolve_bfgs`: objective and
          gradient at the full quasi-Newton step first, the backtracking ladder only for trajectories that reject it; the same
          decisions as a plain ladder; at D = 2 the optimiser itself runs on the device, one wave per trajectory, every trajectory at
          its own pace: `qmps_evolve_bfgs_device`, options {'device_driver': False} for the host loop; {'device_driver': 'trajectory'} at D = 16: the per-trajectory kernel of qmps_evolve_d16.hip, an option measured slower than the lock-step); options {'native': False} runs the same loop from numpy (`tools.batched_bfgs`, per-iteration
          objective history), {'speculative': False} the plain two-batch iteration (gradient columns, then the ladder);
          'carry_hessian', 'tight_gradient', 'adaptive_gradient', 'gradient', 'first_rungs', 'maxiter', 'gtol', 'eps', 'alphas' as in LockstepEvolver;
      anything else: scipy.optimize.minimize on the scalar `obj` per trajectory (the reference's own call).
    Returns the parameter history (n_steps + 1, [T,] P) [and an info dict with the objective history]."""
    cls = state_tensor or _default_class(D)
    single = np.ndim(params) == 1
    X = np.array(np.atleast_2d(params), dtype=float)
    X = X[:, :_n_angles(cls, X[0])]
    T, P = X.shape
    kind = getattr(cls, 'device_kind', None)
    on_device = kind is not None and not (kind in (L.ANSATZ_SHALLOW_FULL, L.ANSATZ_STATE_GATE) and D != 2)
    history, info = [X.copy()], {'fun': []}
    m = method.lower() if isinstance(method, str) else method
    if m in ('rotosolve', 'doublerotosolve') and on_device:
        nsh = 6 if m == 'doublerotosolve' else 3
        eng = _runtime.engine(D, nsh * T)
        Xf, ph, fh = eng.evolve_rotosolve(kind, X, WW, n_steps=n_steps, n_sweeps=n_sweeps, double_frequency=nsh == 6,
                                          max_rounds=max_rounds, tol=tol)
        history += [ph[k] for k in range(n_steps)]
        info['fun'] = fh
        info['solver'] = eng.overlap_stats()
        worse = int(np.sum(fh[:, -1] > fh[:, 0] + 1e-12))
        if worse:
            import warnings
            warnings.warn(f"evolve(method='{method}'): {worse} of {fh[:, -1].size} (step, trajectory) pairs ended a time step with a HIGHER "
                          "objective than their first sweep reached - rotosolve is a heuristic for this objective (see the docstring); "
                          "method='BFGS' minimises it", RuntimeWarning, stacklevel=2)
        if callback is not None:
            for k in range(n_steps):
                callback(k, ph[k] if not single else ph[k][0], fh[k, -1] if not single else fh[k, -1, 0])
    elif m == 'bfgs' and on_device:
        opts = dict(options or {})
        ladder = opts.pop('alphas', (1.0, 0.5, 0.25, 0.125, 1 / 16, 1 / 64, 1 / 256, 1 / 4096))
        mr = max_rounds if max_rounds is not None else (60 if D in (2, 4) else 100000)
        ev = LockstepEvolver(D, T, P, cls, mr, tol, opts.get('maxiter', 200), opts.get('gtol', 1e-5), opts.get('eps', 1e-6), ladder,
                             gradient=opts.get('gradient', 'auto'), first_rungs=opts.get('first_rungs'),
                             carry_hessian=opts.get('carry_hessian', False), speculative=opts.get('speculative', True), native=opts.get('native', True),
                             tight_gradient=opts.get('tight_gradient', False), device_driver=opts.get('device_driver', True),
                             adaptive_gradient=opts.get('adaptive_gradient', True))
        fg, fl = ev.fg, ev.fl
        try:
            for step in range(n_steps):
                res = ev.step(X, WW)
                X = res['x']
                history.append(X.copy())
                info['fun'].append(res['history'])
                info.setdefault('nit', []).append(res['nit'])
                info.setdefault('nfev', []).append(res['nfev'])
                if callback is not None:
                    callback(step, X if not single else X[0], res['fun'] if not single else res['fun'][0])
            info['solver'] = {'gradient_batches': fg.eng.overlap_stats(), 'line_search_batches': fl.eng.overlap_stats()}
        finally:
            ev.close()
        lost = np.array([np.isnan(np.asarray(f)[-1]) for f in info['fun']])        # (n_steps, T)
        if lost.any():
            # a fixed-point solve ended with a status the objective cannot use (include/qmps_hip.h "status"): at D >= 4 SEVERAL dominant
            # eigenvalues of equal modulus (the transfer map of a product state, special angles of the ansatz: no unique fixed point, the
            # two-sided gradient has nothing to stand on, and the library says so instead of returning one of the eigenvectors as ARPACK
            # would), or a solve that exhausted max_rounds.  The trajectory is reported where it stands when the objective is lost - normally
            # the last iterate with a finite objective; when the solve fails at an ACCEPTED backtracking point the parameters have moved there
            import warnings
            first = {int(t): int(np.argmax(lost[:, t])) for t in np.where(lost.any(axis=0))[0]}
            warnings.warn(f"evolve(method='BFGS'): {len(first)} of {T} trajectories have no objective (NaN) from time step "
                          f"{min(first.values())} on: a fixed-point solve of their mixed transfer map did not converge (status != 0: dominant eigenvalues "
                          f"tied in modulus - a product state / special angles of the ansatz - or max_rounds exhausted); the optimiser no longer "
                          f"moves them: trajectories {sorted(first)[:8]}", RuntimeWarning, stacklevel=2)
            info['no_unique_fixed_point'] = first
    else:
        for step in range(n_steps):
            Xn, fs = np.empty_like(X), np.empty(T)
            for t in range(T):
                A = state_tensor_of(cls, D, X[t])
                res = minimize(obj, X[t], (A, WW, D, cls), method=method, options=options or {})
                Xn[t], fs[t] = res.x, res.fun
            X = Xn
            history.append(X.copy())
            info['fun'].append(fs)
            if callback is not None:
                callback(step, X if not single else X[0], fs if not single else fs[0])
    H = np.array(history)
    H = H[:, 0] if single else H
    return (H, info) if return_info else H


# D = 16: the per-trajectory device-resident optimiser (qmps_evolve_bfgs_device, qmps_evolve_d16.hip) is built, tested and NOT the
# default: measured slower than the lock-step at every size (config 4, 256 trajectories: 0.84-1.00 against 0.82-0.96 ms per time
# step; 32: 0.43 / 0.29; 2 048: 3.3 / 2.6; identity start 5-15 / 4.4-5.0) - one trajectory with |eta_2 / eta_1| = 0.82 needs 4.5 x the
# mean's power steps in EVERY evaluation and holds its compute unit (and with it the launch) three times as long as the others,
# while the lock-step gives that straggler's two solves a compute unit each and builds its neighbours elsewhere
# (profiles/EXPERIMENTS.md).  options {'device_driver': 'trajectory'} selects it.
D16_TRAJECTORY_DRIVER = False


class LockstepEvolver:
    """The 'BFGS' branch of `evolve` as an object that keeps its two device contexts (gradient batches, ladder batches) and
    their resident fixed points across time steps: `step(X)` = one time step of all trajectories."""

    def __init__(self, D, T, P, cls=None, max_rounds=None, tol=1e-12, maxiter=200, gtol=1e-5, eps=1e-6,
                 alphas=(1.0, 0.5, 0.25, 0.125, 1 / 16, 1 / 64, 1 / 256, 1 / 4096), device=0, gradient='auto', first_rungs=None,
                 carry_hessian=False, speculative=False, native=True, tight_gradient=False, device_driver=True, adaptive_gradient=True):
        """native (with speculative and the two-sided gradient): the whole time step - every BFGS iteration of every trajectory - is
        ONE C call (qmps_evolve_bfgs: the loop of tools.batched_bfgs with its host arithmetic in C++ inside the library); False: the
        same loop in numpy, one ctypes call per batch.
        tight_gradient=False (two-sided gradient): the objective of an iterate comes from the two-sided quotient <y, T(r)>/<y, r> - its
        error is the product of the residuals of y and r - so the two eigen-solves of a gradient batch stop at max(tol, 1e-8): eta
        to ~1e-16, gradient to ~2e-8 (gtol = 1e-5; scipy's own forward differences carry ~1e-8), ~16 power steps fewer per solve.
        adaptive_gradient (native driver, D = 8, 16, not with tight_gradient): QMPS_BFGS_ADAPTIVE_GRADIENT - a trajectory's gradient solves
        stop at clamp(1e-3 max|g|, max(tol, 1e-8), 1e-6): same minima, a third fewer power steps (False: every batch to max(tol, 1e-8),
        the numbers of the numpy loop).
        gradient: 'fd' = the 2P + 1 central-difference candidates are eigen-solved one by one (any D); 'two-sided' = one right
        and one left eigen-solve per iterate, the neighbours by the second-order formula eta' = <y, T'(r)>/<y, r> (D >= 4);
        'auto' = 'two-sided' where the library has it.  first_rungs: two-stage ladder (tools.batched_bfgs).
        speculative: gradient batches at the full quasi-Newton step first (tools.batched_bfgs): one device batch per iteration
        whenever every trajectory accepts it.
        carry_hessian: every time step starts BFGS from the inverse Hessians the previous step ended with instead of the identity
        (scipy - and the reference - start every minimisation afresh; consecutive time steps minimise nearly the same function)."""
        cls = cls or _default_class(D)
        self.kind = getattr(cls, 'device_kind')
        mr = max_rounds if max_rounds is not None else (60 if D in (2, 4) else 100000)
        self.alphas, self.maxiter, self.gtol, self.eps = tuple(alphas), maxiter, gtol, eps
        self.D = D
        self.two_sided = (gradient == 'two-sided') or (gradient == 'auto' and D >= 4)
        self.first_rungs = first_rungs
        self.speculative = bool(speculative)      # (D = 2: with the central-difference candidates eigen-solved one by one)
        self.carry_hessian, self._hinv = carry_hessian, None
        # the C driver IS the speculative iteration with the library's gradient (two-sided at D >= 4, eigen-solved neighbours at D = 2):
        # any other combination of options (gradient='fd' at D >= 4, speculative=False) runs the numpy loop, which implements them all
        # (first_rungs only shapes the NON-speculative ladder, so it does not matter here)
        self.native = bool(native) and self.speculative and (self.two_sided or D == 2)
        # D = 2 (the reference's own bond dimension): the native driver is the DEVICE-resident one - a wave per trajectory runs every
        # BFGS iteration of every time step without returning to the host (device=False: the host loop of qmps_evolve_bfgs)
        # D = 16 (round 5): a workgroup of eight waves per trajectory (qmps_evolve_d16.hip); device_driver='lockstep' keeps the lock-step
        self.device = (self.native and bool(device_driver) and P <= 16 and 2 * P + len(self.alphas) <= 64 and
                       ((D == 2 and not self.two_sided and self.kind in L.EVOLVE_DEVICE_KINDS_D2) or
                        (D == 4 and gradient == 'auto' and len(self.alphas) <= 9 and self.kind in (L.ANSATZ_SHALLOW_CNOT, 1, 3)) or
                        (D == 16 and self.two_sided and device_driver != 'lockstep' and (device_driver == 'trajectory' or D16_TRAJECTORY_DRIVER)
                         and self.kind in (L.ANSATZ_SHALLOW_CNOT, 3))))
        self.mr, self.tol = mr, tol
        self.tight_gradient = bool(tight_gradient) or not self.two_sided
        self.adaptive_gradient = bool(adaptive_gradient) and not self.tight_gradient and D in (8, 16)
        if self.native:
            # one context serves the gradient batches and the (rare, cold-started) ladder batches
            self.fg = _GroupedObjective(D, self.kind, T, max(2 * P + 1, len(self.alphas) - 1), mr, tol, device=device)
            self.fl = self.fg
            self.fg.tight_gradient = self.tight_gradient
            self._continued = False
            return
        self.fg = _GroupedObjective(D, self.kind, T, 2 * P + 1, mr, tol, device=device)
        rungs = len(self.alphas) if not first_rungs else (first_rungs, len(self.alphas) - first_rungs)
        if self.speculative:
            rungs = len(self.alphas) - 1
        self.fl = _GroupedObjective(D, self.kind, T, rungs, mr, tol, device=device)
        self.fg.tight_gradient = self.tight_gradient

    def steps(self, X, WW, n_steps, counters=True, time_steps=False):
        """n_steps time steps in one C call (native driver): dict(x, params_hist, fun (n_steps, T), nit (n_steps,), ...).
        time_steps (D = 8, 16, with counters): res['device_ms'] = device time of the call (QMPS_BFGS_TIME_STEPS), the run un-instrumented."""
        if not self.native:
            raise RuntimeError('LockstepEvolver.steps needs the native driver (speculative=True, two-sided gradient)')
        if self.device:
            # D = 2, 4: the optimiser itself on the device (a wave / a workgroup per trajectory), the whole call in one launch (qmps_evolve_bfgs_device)
            res = self.fg.eng.evolve_bfgs_device(self.kind, X, WW, n_steps=n_steps, maxiter=self.maxiter, gtol=self.gtol, h=self.eps, alphas=self.alphas,
                                                 carry_hessian=self.carry_hessian, hess_inv=self._hinv if (self.carry_hessian and self._continued) else None,
                                                 max_rounds=self.mr if self.D == 16 else min(self.mr, 60), tol=self.tol, counters=counters,
                                                 tight_gradient=self.tight_gradient and self.D == 16, adaptive_gradient=self.adaptive_gradient and self.D == 16)
            self._continued = True
            self._hinv = res['hess_inv']
            res['nit_per_trajectory'] = res['nit']
            res['nit'] = res['nit'].max(axis=1)                 # (the lock-step drivers' count: the slowest trajectory's)
            res['gradient_batches'], res['ladder_batches'], res['gradient_ms'] = 0, 0, res['kernel_ms']
            if self.fg.kernel_ms is not None and counters:
                self.fg.kernel_ms += [res['kernel_ms']]
            return res
        res = self.fg.eng.evolve_bfgs(self.kind, X, WW, n_steps=n_steps, maxiter=self.maxiter, gtol=self.gtol, h=self.eps, alphas=self.alphas,
                                      carry_hessian=self.carry_hessian, hess_inv=self._hinv if (self.carry_hessian and self._continued) else None,
                                      warm=self._continued, max_rounds=self.mr, tol=self.tol, tight_gradient=self.tight_gradient, counters=counters,
                                      adaptive_gradient=self.adaptive_gradient, time_steps=bool(time_steps) and self.D in (8, 16))
        if time_steps and self.D in (8, 16):
            res['device_ms'] = res['gradient_ms']
            self._continued = True
            self._hinv = res['hess_inv']
            return res
        self._continued = True
        self._hinv = res['hess_inv']
        if self.fg.kernel_ms is not None and res['gradient_batches']:
            self.fg.kernel_ms += [res['gradient_ms'] / res['gradient_batches']] * res['gradient_batches']
        return res

    def step(self, X, WW):
        if self.native:
            res = self.steps(X, WW, 1)
            # history: the objective at the start and at the end of the time step (the numpy loop records every iteration)
            return {'x': res['x'], 'fun': res['fun'][0], 'nit': int(res['nit'][0]), 'nfev': res['nfev'], 'history': np.stack([res['fun_start'][0], res['fun'][0]]),
                    'hess_inv': res['hess_inv'], 'converged': None}
        from .tools import batched_bfgs
        self.fg.set_reference(X, WW)
        self.fl.set_reference(X, WW)
        vg = (lambda Z: self.fg.value_and_grad(Z, self.eps)) if self.two_sided else None
        if vg is None and self.speculative:
            from .tools import batched_fd_gradient
            vg = lambda Z: batched_fd_gradient(self.fg, Z, self.eps)
        def on_active(kind, mask):          # converged trajectories cost nothing in the batches that follow
            (self.fg if kind == 'grad' else self.fl).eng.overlap_set_active(mask)
        res = batched_bfgs(self.fg, self.fl, X, maxiter=self.maxiter, gtol=self.gtol, h=self.eps, alphas=self.alphas, on_active=on_active,
                           value_and_grad=vg, first_rungs=None if self.speculative else self.first_rungs,
                           Hinv0=self._hinv if self.carry_hessian else None, speculative=self.speculative)
        self._hinv = res['hess_inv']
        return res

    def close(self):
        self.fg.close()
        if self.fl is not self.fg:
            self.fl.close()


def state_tensor_of(cls, D, p):
    return unitary_to_tensor(unitary(build_gate(cls, D, p)))


# ---- the rest of the reference module's surface ------------------------------------------------------------
def state_gate(v, symbol='R'):
    """new_time_evolve.py:189-190."""
    from .represent import StateGate
    return StateGate(v, symbol)


def run_tests(N, rng=None):
    """The reference's in-file assertion suite (new_time_evolve.py:50-184) through this package: random left-canonical tensors A, B; the right
    fixed point r and the left fixed point l of `Map(A, B)` FROM THE DEVICE (`qmps_overlap_batch`: l is the right fixed point of the map of
    the daggered tensors, eigenvalue conj(x) - the convention the reference's own asserts pin: tests/golden/cirq_shim.py:_dominant); the
    circuits on the host's gate objects.  Raises AssertionError on the first identity that fails:
        embeddings round-trip and are unitary;  2 psi[0] = tr(g r), x tr(g r), x^2 tr(g r);  = tr(g l*), x tr(g l*), x^2 tr(g l*);
        the 6-qubit overlap circuit: 2 psi[0] = x^2 tr(l^+ r)."""
    from .represent import CNOT, H, Environment, MatrixGate, final_state, line_qubits
    from .time_evolve_tools import (get_env_off_left_site, get_env_off_right_site, overlap_of_tensors, put_env_on_left_site,
                                    put_env_on_right_site)
    from .tools import tensor_to_unitary
    rng = np.random.default_rng() if rng is None else rng
    paulis = [np.eye(2, dtype=complex), np.array([[0, 1], [1, 0]], dtype=complex), np.array([[0, -1j], [1j, 0]]), np.diag([1.0 + 0j, -1.0])]

    def random_tensor():
        z = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
        return unitary_to_tensor(np.linalg.qr(z)[0])

    def fixed_points(A, B):
        _, r = overlap_of_tensors(A, B, want_r=True)
        Ad, Bd = A.conj().transpose(0, 2, 1), B.conj().transpose(0, 2, 1)
        _, l = overlap_of_tensors(Ad, Bd, want_r=True)
        x = np.vdot(r, sum(A[s] @ r @ B[s].conj().T for s in range(2)))         # Rayleigh quotient: the eigenvalue with its phase
        assert np.abs(sum(A[s] @ r @ B[s].conj().T for s in range(2)) - x * r).max() < 1e-10
        assert np.abs(sum(Ad[s] @ l @ Bd[s].conj().T for s in range(2)) - np.conj(x) * l).max() < 1e-10
        return x, r, l

    def amp(ops, n):
        return 2 * final_state(ops, n)[0]

    for _ in range(N):
        q = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
        for put, off in ((put_env_on_left_site, get_env_off_left_site), (put_env_on_right_site, get_env_off_right_site)):
            E, n = put(q, ret_n=True)
            assert np.allclose(off(E * n), q) and np.allclose(E.conj().T @ E, np.eye(4))
        A, B = random_tensor(), random_tensor()
        x, r, l = fixed_points(A, B)
        U, U_ = Environment(tensor_to_unitary(A), 'U'), Environment(tensor_to_unitary(B), "U'")
        R = Environment(put_env_on_left_site(r), 'R')
        Lg = Environment(put_env_on_right_site(l.conj().T), 'L')
        for g in paulis:
            G = MatrixGate(g)
            qb = line_qubits(4)
            assert abs(amp([H(qb[1]), CNOT(*qb[1:3]), R(*qb[2:]), G(qb[1]), CNOT(*qb[1:3]), H(qb[1])], 4) - np.trace(g @ r)) < 1e-9
            assert abs(amp([H(qb[1]), CNOT(*qb[1:3]), U(*qb[0:2]), R(*qb[2:]), G(qb[0]), (U_ ** -1)(*qb[0:2]), CNOT(*qb[1:3]), H(qb[1])], 4)
                       - x * np.trace(g @ r)) < 1e-9
            qb = line_qubits(5)
            assert abs(amp([H(qb[2]), CNOT(*qb[2:4]), U(*qb[1:3]), U(*qb[0:2]), R(*qb[3:]), G(qb[0]), (U_ ** -1)(*qb[0:2]), (U_ ** -1)(*qb[1:3]),
                            CNOT(*qb[2:4]), H(qb[2])], 5) - x ** 2 * np.trace(g @ r)) < 1e-9
            qb = line_qubits(3)
            assert abs(amp([H(qb[1]), CNOT(*qb[1:3]), Lg(*qb[:2]), G(qb[2]), CNOT(*qb[1:3]), H(qb[1])], 3) - np.trace(g @ l.conj())) < 1e-9
            qb = line_qubits(4)
            assert abs(amp([H(qb[2]), CNOT(*qb[2:4]), U(*qb[1:3]), Lg(*qb[:2]), G(qb[3]), (U_ ** -1)(*qb[1:3]), CNOT(*qb[2:4]), H(qb[2])], 4)
                       - x * np.trace(g @ l.conj())) < 1e-9
            qb = line_qubits(5)
            assert abs(amp([H(qb[3]), CNOT(*qb[3:5]), U(*qb[2:4]), U(*qb[1:3]), Lg(*qb[0:2]), G(qb[4]), (U_ ** -1)(*qb[1:3]), (U_ ** -1)(*qb[2:4]),
                            CNOT(*qb[3:5]), H(qb[3])], 5) - x ** 2 * np.trace(g @ l.conj())) < 1e-9
        qb = line_qubits(6)
        assert abs(amp([H(qb[3]), CNOT(*qb[3:5]), U(*qb[2:4]), U(*qb[1:3]), Lg(*qb[0:2]), R(*qb[4:]), (U_ ** -1)(*qb[1:3]), (U_ ** -1)(*qb[2:4]),
                        CNOT(*qb[3:5]), H(qb[3])], 6) - x ** 2 * np.trace(l.conj().T @ r)) < 1e-9


def obj_state(p_, A, WW):
    """new_time_evolve.py:223-247: the 5-qubit register after R = StateGate(p_[15:]) on qubits (3,4), U U W, L on (0,1), U'^+ U'^+,
    CNOT, H - the state function the reference's rotosolve variant pairs with `obj_H` (energy = -|psi[0]|^2).  Host state-vector
    pass over the package's gate objects (represent.final_state), like the reference's simulator call; psi[0] on its own -
    <r^, T(r^)>_F / sqrt(2), r = environment_from_unitary(R) - is `obj_state_amplitudes` on the device."""
    from .represent import CNOT, H, Environment, final_state, line_qubits
    from .time_evolve_tools import put_env_on_right_site
    from .tools import environment_from_unitary, tensor_to_unitary
    p_ = np.asarray(p_, dtype=float)
    p, rs = p_[:15], p_[15:]
    B = unitary_to_tensor(unitary(gate(p)))
    U = Environment(tensor_to_unitary(np.asarray(A, dtype=complex)), 'U')
    U_ = Environment(tensor_to_unitary(B), "U'")
    R = state_gate(rs)
    Lg = Environment(put_env_on_right_site(environment_from_unitary(unitary(R)).conj().T), 'L')
    W = Environment(np.asarray(WW, dtype=complex), 'W')
    q = line_qubits(5)
    return final_state([R(*q[3:5]), U(*q[2:4]), U(*q[1:3]), W(*q[2:4]), Lg(*q[0:2]), (U_ ** -1)(*q[1:3]), (U_ ** -1)(*q[2:4]),
                        CNOT(*q[3:5]), H(q[3])], 5)


def obj_state_amplitudes(P_, A, WW):
    """psi[0] of `obj_state` for a batch of parameter vectors P_ (n, 15 + 6) against one state tensor A, on the device."""
    from .tools import environment_from_unitary
    P_ = np.atleast_2d(np.asarray(P_, dtype=float))
    cand = np.stack([unitary_to_tensor(unitary(gate(p[:15]))) for p in P_])
    q = np.stack([environment_from_unitary(unitary(state_gate(p[15:]))) for p in P_])
    eng = _runtime.engine(2, len(cand))
    eng.set_tensors(cand)
    eng.overlap_set(np.asarray(A, dtype=complex), WW)
    return eng.overlap_amplitudes(q) * np.sqrt(2.0)


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
        return np.where(L.overlap_usable(st), -np.abs(eta), np.nan)

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
    params = np.array(params, dtype=float)[:15]      # (the reference carries an unused tail `rs` behind the 15 gate angles)
    A0 = state_tensor(params)
    if ops is None:
        ops = [0.5 * np.array([[0, 1], [1, 0]]), 0.5 * np.array([[0, -1j], [1j, 0]]), 0.5 * np.diag([1.0, -1.0])]
    ps, evs, les = [params.copy()], [], []
    for _ in T[1:]:
        A = state_tensor(params)
        res = minimize(obj, params, (A, WW), method=method, options=options or {})
        params = res.x[:15]
        evs.append(one_site_expectations(A, ops))
        les.append(loschmidt_overlap(A, A0))
        ps.append(params.copy())
    return np.array(ps), np.array(evs), np.array(les)
