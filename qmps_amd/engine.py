"""Batched engine: the Python face of one libqmps_hip context (one GPU, one stream).

`EnergyEngine.energies(...)` is the batched counterpart of the reference's scalar objective
`objective_function(params) -> float` (qmps/ground_state.py:150-168, 251-266): it evaluates B
independent (state tensor, Hamiltonian) pairs per launch on the MI355X.
"""
import ctypes
from ctypes import byref, c_double, c_float, c_int32, c_void_p

import numpy as np

from . import _lib as L

_dp = ctypes.POINTER(c_double)
_ip = ctypes.POINTER(c_int32)


def _f64(a):
    return a.ctypes.data_as(_dp)


def _i32(a):
    return a.ctypes.data_as(_ip)


def _c128(a, shape_tail, name):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    if a.shape[1:] != tuple(shape_tail):
        raise ValueError(f'{name}: expected shape (B,{",".join(map(str, shape_tail))}), got {a.shape}')
    return a


class EnergyEngine:
    """Owns one `qmps_ctx`.  Not thread-safe; one engine per thread / device."""

    def __init__(self, D, max_batch, device=0):
        self._lib = L.load()
        self.D = int(D)
        self.max_batch = int(max_batch)
        self.device = int(device)
        self._ctx = c_void_p()
        L.check(self._lib.qmps_create(self.device, self.D, self.max_batch, byref(self._ctx)))
        self.n_terms = 0
        self.B = 0

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            self._lib.qmps_destroy(self._ctx)
            self._ctx = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- inputs -----------------------------------------------------------------------------
    def set_tensors(self, A):
        """A: (B, 2, D, D) complex128 state tensors, A[b, s, i, j]."""
        A = _c128(A, (2, self.D, self.D), 'A')
        L.check(self._lib.qmps_set_states(self._ctx, A.shape[0], _f64(A.view(np.float64)), L.INPUT_TENSOR))
        self.B = A.shape[0]

    def set_unitaries(self, U):
        """U: (B, 2D, 2D) complex128 state unitaries; unitary_to_tensor runs on the device."""
        U = _c128(U, (2 * self.D, 2 * self.D), 'U')
        L.check(self._lib.qmps_set_states(self._ctx, U.shape[0], _f64(U.view(np.float64)), L.INPUT_UNITARY))
        self.B = U.shape[0]

    def set_ansatz_params(self, kind, params):
        """params: (B, P) float64 ansatz angles; the state tensors are built on the device.
        kind: L.ANSATZ_* (0 ShallowCNOT, 1 QAOA, 2 ShallowFull [D = 2], 3 ShallowCNOT3)."""
        params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        L.check(self._lib.qmps_set_states_ansatz(self._ctx, params.shape[0], int(kind), params.shape[1], _f64(params)))
        self.B = params.shape[0]

    def rotosolve(self, kind, params, n_sweeps=1, max_iter=10000, tol=1e-13):
        """Device-resident rotosolve: params (R, P) -> (energies (n_sweeps, R), params (R, P)).
        The Hamiltonian must be resident (`set_hamiltonian`)."""
        P = np.array(np.atleast_2d(params), dtype=np.float64, order='C', copy=True)
        hist = np.empty((int(n_sweeps), P.shape[0]))
        L.check(self._lib.qmps_rotosolve(self._ctx, P.shape[0], int(kind), P.shape[1], _f64(P), int(n_sweeps),
                                         int(max_iter), float(tol), _f64(hist)))
        self.B = P.shape[0]         # the library leaves the R final vectors (tensors, energies, statuses) resident
        return hist, P

    def set_roto_rule(self, rule):
        """Update rule of the double-frequency drivers: L.ROTO_REFERENCE (scipy's bounded search, what tools.py:451 runs) or
        L.ROTO_GLOBAL_ARGMIN (global minimiser of the fit: departs from the reference's trajectory)."""
        L.check(self._lib.qmps_set_roto_rule(self._ctx, int(rule)))

    def roto_rule_probe(self, abcd, rule=L.ROTO_REFERENCE):
        """The device's update step for fits a sin 2x + b cos 2x + c sin x + d cos x, abcd (n, 4) -> theta (n,)."""
        abcd = np.ascontiguousarray(abcd, dtype=np.float64).reshape(-1, 4)
        out = np.empty(len(abcd))
        L.check(self._lib.qmps_roto_rule_probe(self._ctx, len(abcd), _f64(abcd), int(rule), _f64(out)))
        return out

    def double_rotosolve(self, kind, params, n_sweeps=1, max_iter=10000, tol=1e-13, rule=L.ROTO_REFERENCE):
        """Device-resident DOUBLE-frequency rotosolve (qmps/tools.py:422-457): params (R, P) ->
        (energies (n_sweeps, R), params (R, P)); needs 6 R <= max_batch and a resident Hamiltonian.
        rule: see `set_roto_rule` (set on every call: engines are cached and shared)."""
        P = np.array(np.atleast_2d(params), dtype=np.float64, order='C', copy=True)
        hist = np.empty((int(n_sweeps), P.shape[0]))
        self.set_roto_rule(rule)
        L.check(self._lib.qmps_double_rotosolve(self._ctx, P.shape[0], int(kind), P.shape[1], _f64(P), int(n_sweeps),
                                                int(max_iter), float(tol), _f64(hist)))
        self.B = P.shape[0]
        return hist, P

    def tensors(self, B=None):
        """Read back the resident state tensors (B, 2, D, D)."""
        B = self.B if B is None else B
        A = np.empty((B, 2, self.D, self.D), dtype=np.complex128)
        L.check(self._lib.qmps_get_states(self._ctx, B, _f64(A.view(np.float64))))
        return A

    def set_hamiltonian(self, h):
        """h: (4, 4) or (n_terms, 4, 4) complex; index 2*s1+s2 with s1 the left site."""
        h = np.ascontiguousarray(np.asarray(h, dtype=np.complex128).reshape(-1, 4, 4))
        if getattr(self, '_h_resident', None) is not None and self._h_resident.shape == h.shape and np.array_equal(self._h_resident, h):
            return                  # already resident (an optimiser passes the same Hamiltonian with every objective call)
        self._h_resident = None
        L.check(self._lib.qmps_set_hamiltonian(self._ctx, h.shape[0], _f64(h.view(np.float64))))
        self.n_terms = h.shape[0]
        self._h_resident = h.copy()

    def set_window(self, first):
        """Several resident batches side by side: after set_window(first) the launch / read-back calls address
        evaluations [first, first + B) of the resident arrays (`set_*` calls reset the window to 0)."""
        L.check(self._lib.qmps_set_window(self._ctx, int(first)))

    def set_env_guess(self, r0):
        if r0 is None:
            L.check(self._lib.qmps_set_env_guess(self._ctx, 0, None))
            return
        r0 = _c128(r0, (self.D, self.D), 'r0')
        L.check(self._lib.qmps_set_env_guess(self._ctx, r0.shape[0], _f64(r0.view(np.float64))))

    # -- hot path ---------------------------------------------------------------------------
    _SOLVERS = {'plain': L.ENV_POWER, 'squaring': L.ENV_POWER_SQUARING, 'direct': L.ENV_DIRECT}

    def launch(self, B=None, max_iter=10000, tol=1e-13, solver='direct', store_env=True, accumulate_cost=False, warm_start=False, krylov_fallback=False):
        """Asynchronous: right environment + energies for the resident batch.
        solver: 'direct' (exact fixed-point solve accepted by one power step; D = 4, other bond dimensions run
        'squaring'), 'squaring' (power iteration 2^m steps at a time) or 'plain' (power iteration).
        store_env=False ('direct' at D = 4 only): the environments are not written to HBM.
        accumulate_cost=True (same): the kernel also sums the energies (exact fixed-point accumulation); the
        `cost_launch` that must follow then launches no reduction kernel.
        warm_start=True: start from the RESIDENT environments (an earlier launch with store_env, or `set_env_guess`); with
        'direct' at D = 4 an evaluation whose environment passes the acceptance test as it is skips the solve.
        krylov_fallback=True (D = 8; the one-shot calls `energies*` always do): long tails of the power iteration go to the Arnoldi kernel."""
        flag = self._SOLVERS[solver] | (0 if store_env else L.FLAG_NO_ENV_OUT) | (L.FLAG_ACCUMULATE_COST if accumulate_cost else 0)
        flag |= L.FLAG_WARM_RESIDENT if warm_start else 0
        flag |= L.FLAG_KRYLOV_FALLBACK if krylov_fallback else 0
        L.check(self._lib.qmps_energy_launch(self._ctx, self.B if B is None else B, int(max_iter), float(tol), flag))

    def set_solver(self, solver='direct', handoff=None):
        """Solver of the one-shot calls (`energies`, `env_batch`, `rotosolve`) and the hand-off point of the tail."""
        flag = self._SOLVERS[solver]
        L.check(self._lib.qmps_set_default_solver(self._ctx, flag))
        if handoff is not None:
            L.check(self._lib.qmps_set_handoff(self._ctx, int(handoff)))

    @property
    def handoff(self):
        v = ctypes.c_int(0)
        L.check(self._lib.qmps_get_handoff(self._ctx, byref(v)))
        return v.value

    @property
    def squaring_schedule(self):
        """(untracked squarings, D = 4 mat-vecs between further squarings) in force for this context."""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        L.check(self._lib.qmps_get_squaring_schedule(self._ctx, byref(a), byref(b)))
        return a.value, b.value

    def set_exchange_period(self, steps):
        """Summed costs of `steps` consecutive cost_launch calls travel in one all-reduce (1..16, default 1)."""
        L.check(self._lib.qmps_set_exchange_period(self._ctx, int(steps)))

    def exchange_stats(self, reset=False):
        """(slot-guard checks, times the previous exchange of the slot was still in flight, host milliseconds spent waiting)."""
        import ctypes
        a, b, ms = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_double(0.0)
        L.check(self._lib.qmps_exchange_stats(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(ms), 1 if reset else 0))
        return int(a.value), int(b.value), float(ms.value)

    def set_kernel_timing_period(self, period):
        """HIP events around the dominant kernel on every `period`-th launch only (they cost ~3 us each on the
        stream); 1 = every launch (default), 0 = never."""
        L.check(self._lib.qmps_set_kernel_timing_period(self._ctx, int(period)))

    def launch_energy_only(self, B=None):
        L.check(self._lib.qmps_energy_only_launch(self._ctx, self.B if B is None else B))

    def sync(self):
        """Waits for both streams.  COLLECTIVE when a communicator exists and a group of costs is partly filled (it is
        exchanged now): call it on every rank, like `cost_launch`, `get_cost`, `allreduce_cost`, `allreduce_sum` and
        `set_exchange_period`."""
        L.check(self._lib.qmps_sync(self._ctx))

    def results(self, B=None):
        B = self.B if B is None else B
        E = np.empty((B, self.n_terms))
        it = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        L.check(self._lib.qmps_get_energies(self._ctx, B, _f64(E), _i32(it), _i32(st)))
        return E, it, st

    def results_status(self, B=None):
        """Per-evaluation status of the last launch (energy or overlap) only."""
        B = self.B if B is None else B
        st = np.empty(B, dtype=np.int32)
        L.check(self._lib.qmps_get_status(self._ctx, B, _i32(st)))
        return st

    def environments(self, B=None):
        B = self.B if B is None else B
        r = np.empty((B, self.D, self.D), dtype=np.complex128)
        L.check(self._lib.qmps_get_env(self._ctx, B, _f64(r.view(np.float64))))
        return r

    def rdm(self, B=None):
        B = self.B if B is None else B
        rho = np.empty((B, 4, 4), dtype=np.complex128)
        L.check(self._lib.qmps_get_rdm(self._ctx, B, _f64(rho.view(np.float64))))
        return rho

    def summed_cost(self, B=None):
        cost = np.empty(max(self.n_terms, 1))
        L.check(self._lib.qmps_sum_energies(self._ctx, self.B if B is None else B, _f64(cost)))
        return cost[:self.n_terms]

    # -- one-shot (host buffers in, host buffers out) -------------------------------------------
    def energies(self, states, h, kind='tensor', r0=None, max_iter=10000, tol=1e-13):
        """states: A (B,2,D,D) [kind='tensor'] or U (B,2D,2D) [kind='unitary'] -> (E (B,n_terms), iters, status)."""
        tail = (2, self.D, self.D) if kind == 'tensor' else (2 * self.D, 2 * self.D)
        states = _c128(states, tail, 'states')
        h = np.ascontiguousarray(np.asarray(h, dtype=np.complex128).reshape(-1, 4, 4))
        B, nt = states.shape[0], h.shape[0]
        r0c = None if r0 is None else _c128(r0, (self.D, self.D), 'r0')
        E = np.empty((B, nt))
        it = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        self._h_resident = None     # the call uploads h before it can fail: the cache is valid again only after success
        L.check(self._lib.qmps_energy_batch(
            self._ctx, B, _f64(states.view(np.float64)), L.INPUT_TENSOR if kind == 'tensor' else L.INPUT_UNITARY,
            _f64(h.view(np.float64)), nt, None if r0c is None else _f64(r0c.view(np.float64)), int(max_iter),
            float(tol), _f64(E), _i32(it), _i32(st)))
        self.B, self.n_terms = B, nt
        self._h_resident = h.copy()
        return E, it, st

    def energies_from_params(self, kind, params, h, max_iter=10000, tol=1e-13):
        """Ansatz parameters (B, P) -> (E (B, n_terms), iters, status) in ONE round trip (the optimisers' call shape)."""
        P = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        h = np.ascontiguousarray(np.asarray(h, dtype=np.complex128).reshape(-1, 4, 4))
        B, nt = P.shape[0], h.shape[0]
        E = np.empty((B, nt))
        it = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        self._h_resident = None
        L.check(self._lib.qmps_energy_batch_ansatz(self._ctx, B, int(kind), P.shape[1], _f64(P), _f64(h.view(np.float64)), nt,
                                                   int(max_iter), float(tol), _f64(E), _i32(it), _i32(st)))
        self.B, self.n_terms = B, nt
        self._h_resident = h.copy()
        return E, it, st

    def energies_from_su(self, params, h, max_iter=10000, tol=1e-13):
        """SU(2D) parameters (B, (2D)^2 - 1) -> (E (B, n_terms), iters, status): matrix exponential, unitary_to_tensor,
        environment and energies on the device, one round trip (NonSparseFullEnergyOptimizer's call shape)."""
        P = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        if P.shape[1] != (2 * self.D) ** 2 - 1:
            raise ValueError(f'SU({2 * self.D}) takes {(2 * self.D) ** 2 - 1} parameters, got {P.shape[1]}')
        h = np.ascontiguousarray(np.asarray(h, dtype=np.complex128).reshape(-1, 4, 4))
        B, nt = P.shape[0], h.shape[0]
        E = np.empty((B, nt))
        it = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        self._h_resident = None
        L.check(self._lib.qmps_energy_batch_su(self._ctx, B, _f64(P), _f64(h.view(np.float64)), nt, int(max_iter), float(tol),
                                               _f64(E), _i32(it), _i32(st)))
        self.B, self.n_terms = B, nt
        self._h_resident = h.copy()
        return E, it, st

    def su_unitaries(self, params, N):
        """SU(N) parameters (B, N^2 - 1) -> unitaries (B, N, N) built on the device (N in 4, 8, 16, 32)."""
        P = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        if P.shape[1] != N * N - 1:
            raise ValueError(f'SU({N}) takes {N * N - 1} parameters, got {P.shape[1]}')
        U = np.empty((P.shape[0], N, N), dtype=np.complex128)
        L.check(self._lib.qmps_su_unitaries(self._ctx, P.shape[0], int(N), _f64(P), _f64(U.view(np.float64))))
        return U

    def cell2_energies_su(self, params, h, max_iter=10000, tol=1e-13):
        """Two-site unit cell from the optimiser's 30 parameters per row: U1 = U4(p[:15]), U2 = U4(p[15:]) on the device."""
        P = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        if P.shape[1] != 30:
            raise ValueError('the two-site unit cell takes 30 parameters')
        h = np.ascontiguousarray(np.asarray(h, dtype=np.complex128).reshape(-1, 4, 4))
        B, nt = P.shape[0], h.shape[0]
        E = np.empty((B, nt))
        it = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        self._h_resident = None
        L.check(self._lib.qmps_cell2_energy_batch_su(self._ctx, B, _f64(P), _f64(h.view(np.float64)), nt, int(max_iter), float(tol),
                                                     _f64(E), _i32(it), _i32(st)))
        self.n_terms = nt
        self._h_resident = h.copy()
        return E, it, st

    def env_batch(self, states, kind='tensor', r0=None, max_iter=10000, tol=1e-13):
        tail = (2, self.D, self.D) if kind == 'tensor' else (2 * self.D, 2 * self.D)
        states = _c128(states, tail, 'states')
        B = states.shape[0]
        r0c = None if r0 is None else _c128(r0, (self.D, self.D), 'r0')
        r = np.empty((B, self.D, self.D), dtype=np.complex128)
        it = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        L.check(self._lib.qmps_env_batch(
            self._ctx, B, _f64(states.view(np.float64)), L.INPUT_TENSOR if kind == 'tensor' else L.INPUT_UNITARY,
            None if r0c is None else _f64(r0c.view(np.float64)), int(max_iter), float(tol),
            _f64(r.view(np.float64)), _i32(it), _i32(st)))
        self.B = B
        if self.n_terms == 0:
            self.n_terms = 1
            self._h_resident = None       # the library made h = 0 resident
        return r, it, st

    def cell2_energies(self, U1, U2, h, max_iter=10000, tol=1e-13):
        """Two-site unit cell (D = 2): U1, U2 (B,4,4) -> (E (B,n_terms) = (E1+E2)/2, iters, status)."""
        U1 = _c128(U1, (2 * self.D, 2 * self.D), 'U1')
        U2 = _c128(U2, (2 * self.D, 2 * self.D), 'U2')
        if U1.shape != U2.shape:
            raise ValueError('U1 and U2 must have the same shape')
        h = np.ascontiguousarray(np.asarray(h, dtype=np.complex128).reshape(-1, 4, 4))
        B, nt = U1.shape[0], h.shape[0]
        E = np.empty((B, nt))
        it = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        self._h_resident = None
        L.check(self._lib.qmps_cell2_energy_batch(self._ctx, B, _f64(U1.view(np.float64)), _f64(U2.view(np.float64)),
                                                  _f64(h.view(np.float64)), nt, int(max_iter), float(tol), _f64(E),
                                                  _i32(it), _i32(st)))
        self.n_terms = nt
        self._h_resident = h.copy()
        return E, it, st

    def overlaps(self, A, candidates, WW, kind='tensor', ansatz=None, max_rounds=None, tol=1e-13, want_r=False):
        """Time-evolution overlap at the engine's bond dimension D: dominant eigenvalue eta of the mixed two-site
        transfer map between WW . merge(A, A) and merge(B, B) for every candidate.  A: (2,D,D) shared or (B,2,D,D);
        candidates: tensors (B,2,D,D) [kind='tensor'], unitaries (B,2D,2D) ['unitary'] or parameters (B,P) ['params' with
        ansatz = L.ANSATZ_*].  max_rounds: D = 2, 4 squarings of the matrix of the map (default 40, <= 60); D = 8, 16 cap on
        power steps (default 20000).
        Returns (eta complex (B,), rounds, status[, r (B,D,D)])."""
        D = self.D
        if max_rounds is None:
            max_rounds = 40 if D in (2, 4) else 20000
        A = np.ascontiguousarray(A, dtype=np.complex128)
        shared = A.ndim == 3
        if A.shape[-3:] != (2, D, D):
            raise ValueError(f'A: expected (2,{D},{D}) or (B,2,{D},{D}), got {A.shape}')
        WW = np.ascontiguousarray(WW, dtype=np.complex128).reshape(4, 4)
        if kind == 'params':
            cand = np.ascontiguousarray(np.atleast_2d(candidates), dtype=np.float64)
            code, npar, ptr = L.INPUT_ANSATZ_BASE + int(ansatz), cand.shape[1], _f64(cand)
        else:
            tail = (2, D, D) if kind == 'tensor' else (2 * D, 2 * D)
            cand = _c128(candidates, tail, 'candidates')
            code, npar, ptr = (L.INPUT_TENSOR if kind == 'tensor' else L.INPUT_UNITARY), 0, _f64(cand.view(np.float64))
        B = cand.shape[0]
        if not shared and A.shape[0] != B:
            raise ValueError('A must be (2,D,D) or (B,2,D,D)')
        eta = np.empty(B, dtype=np.complex128)
        r = np.empty((B, D, D), dtype=np.complex128) if want_r else None
        rounds = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        L.check(self._lib.qmps_overlap_batch(self._ctx, B, _f64(A.view(np.float64)), 1 if shared else 0, ptr, code, npar,
                                             _f64(WW.view(np.float64)), int(max_rounds), float(tol),
                                             _f64(eta.view(np.float64)), None if r is None else _f64(r.view(np.float64)),
                                             _i32(rounds), _i32(st)))
        self.B = B
        return (eta, rounds, st, r) if want_r else (eta, rounds, st)

    def overlap_set(self, A, WW):
        """Resident form of `overlaps`: reference tensor(s) A (2,D,D) or (n,2,D,D) and the two-site operator WW."""
        A = np.ascontiguousarray(A, dtype=np.complex128)
        if A.shape[-3:] != (2, self.D, self.D):
            raise ValueError(f'A: expected (2,{self.D},{self.D}) or (n,2,{self.D},{self.D}), got {A.shape}')
        WW = np.ascontiguousarray(WW, dtype=np.complex128).reshape(4, 4)
        L.check(self._lib.qmps_overlap_set(self._ctx, 1 if A.ndim == 3 else A.shape[0], _f64(A.view(np.float64)),
                                           _f64(WW.view(np.float64))))

    def overlap_set_refs_params(self, kind, ref_params, WW):
        """Reference states as ansatz parameters (n, P): the tensors A_t = tensor(params_t) are built on the device
        (what the reference does at the top of every time step, new_time_evolve.py:281-283)."""
        P = np.ascontiguousarray(np.atleast_2d(ref_params), dtype=np.float64)
        WW = np.ascontiguousarray(WW, dtype=np.complex128).reshape(4, 4)
        L.check(self._lib.qmps_overlap_set_refs_ansatz(self._ctx, P.shape[0], int(kind), P.shape[1], _f64(P), _f64(WW.view(np.float64))))

    def overlap_set_group(self, group):
        """Trajectory-major batches: candidate b is compared with reference b // group (0 = shared / one per candidate)."""
        L.check(self._lib.qmps_overlap_set_group(self._ctx, int(group)))

    def overlap_set_active(self, active):
        """One-shot mask for the next overlap launch / evaluation / gradient: trajectories with active[t] == False are skipped
        (their outputs keep the previous launch's values).  None disarms."""
        if active is None:
            L.check(self._lib.qmps_overlap_set_active(self._ctx, 0, None))
            return
        m = np.ascontiguousarray(np.asarray(active, dtype=bool).astype(np.uint8))
        L.check(self._lib.qmps_overlap_set_active(self._ctx, m.size, m.ctypes.data_as(ctypes.c_char_p)))

    def overlap_launch(self, B=None, max_rounds=None, tol=1e-13, want_r=False, warm=False):
        """Asynchronous: overlaps of the resident candidates [window, window + B) with the resident reference(s).
        warm=True (D = 8, 16): start every candidate from the fixed point its slot holds from the previous launch
        (implies want_r: the new fixed points stay resident)."""
        if max_rounds is None:
            max_rounds = 40 if self.D in (2, 4) else 20000
        flags = (L.OVERLAP_WANT_R if (want_r or warm) else 0) | (L.OVERLAP_WARM if warm else 0)
        L.check(self._lib.qmps_overlap_launch(self._ctx, self.B if B is None else B, int(max_rounds), float(tol), flags))

    def overlap_eval_params(self, kind, cand, max_rounds=None, tol=1e-12, want_r=False, warm=False):
        """Candidate parameters (B, P) -> (f = -sqrt|eta| (B,), status (B,)) against the resident references, one round trip."""
        cand = np.ascontiguousarray(np.atleast_2d(cand), dtype=np.float64)
        if max_rounds is None:
            max_rounds = 40 if self.D in (2, 4) else 20000
        B = cand.shape[0]
        f = np.empty(B)
        st = np.empty(B, dtype=np.int32)
        flags = (L.OVERLAP_WANT_R if (want_r or warm) else 0) | (L.OVERLAP_WARM if warm else 0)
        L.check(self._lib.qmps_overlap_eval_ansatz(self._ctx, B, int(kind), cand.shape[1], _f64(cand), int(max_rounds), float(tol), flags,
                                                   _f64(f), _i32(st)))
        self.B = B
        return f, st

    def overlap_gradient(self, kind, X, h=1e-6, max_rounds=None, tol=1e-12, warm=False, two_sided_f=False):
        """Iterates X (T, P) -> (f (T,), g (T, P), status (T,)): objective and its central-difference gradient against the
        resident references, from one right + one left eigen-solve per iterate (qmps_overlap_gradient; D = 4, 8, 16).
        two_sided_f: f from the two-sided quotient as well (error ~ tol^2: the solves may stop at tol ~ 1e-8)."""
        X = np.ascontiguousarray(np.atleast_2d(X), dtype=np.float64)
        if max_rounds is None:
            max_rounds = 100000
        T, P = X.shape
        f, g, st = np.empty(T), np.empty((T, P)), np.empty(T, dtype=np.int32)
        L.check(self._lib.qmps_overlap_gradient(self._ctx, T, int(kind), P, _f64(X), float(h), int(max_rounds), float(tol),
                                                (L.OVERLAP_WARM if warm else 0) | (L.OVERLAP_TWO_SIDED_F if two_sided_f else 0), _f64(f), _f64(g), _i32(st)))
        self.B = T
        return f, g, st

    def overlap_objective(self, B=None):
        """f_b = -sqrt(|eta_b|) of the last overlap launch (new_time_evolve.py:221), computed on the device."""
        B = self.B if B is None else B
        f = np.empty(B)
        L.check(self._lib.qmps_overlap_get_objective(self._ctx, B, _f64(f)))
        return f

    def overlap_amplitudes(self, q, B=None):
        """psi[0] of the reference's overlap circuit for GIVEN environments q (B,D,D) on its outer qubits (the variational route:
        `get_overlap`, qmps/time_evolve_tools.py:95-131; `obj_state`, qmps/new_time_evolve.py:223-247) = 1/2 <q^, T(q^)>_F with
        q^ = q/||q||_F, for the resident candidates of the window against the references of `overlap_set*`.  Returns complex (B,)."""
        q = _c128(q, (self.D, self.D), 'q')
        B = q.shape[0] if B is None else int(B)
        if q.shape[0] != B:
            raise ValueError('one environment per candidate')
        amp = np.empty(B, dtype=np.complex128)
        L.check(self._lib.qmps_overlap_amplitude(self._ctx, B, _f64(q.view(np.float64)), _f64(amp.view(np.float64))))
        return amp

    def overlap_stats(self, reset=False):
        """dict(evaluations, rounds_sum, rounds_max, not_converged) accumulated by the overlap kernels since the last reset."""
        v = [ctypes.c_int64(0) for _ in range(4)]
        L.check(self._lib.qmps_overlap_stats(self._ctx, byref(v[0]), byref(v[1]), byref(v[2]), byref(v[3]), 1 if reset else 0))
        return {'evaluations': v[0].value, 'rounds_sum': v[1].value, 'rounds_max': v[2].value, 'not_converged': v[3].value}

    def evolve_rotosolve(self, kind, params, WW, n_steps=1, n_sweeps=1, double_frequency=False, max_rounds=None, tol=1e-12, rule=L.ROTO_REFERENCE):
        """Device-resident time evolution (qmps_evolve_rotosolve): params (T, P) -> (params after the last step (T, P),
        params_hist (n_steps, T, P), f_hist (n_steps, n_sweeps, T))."""
        P = np.array(np.atleast_2d(params), dtype=np.float64, order='C', copy=True)
        WW = np.ascontiguousarray(WW, dtype=np.complex128).reshape(4, 4)
        if max_rounds is None:
            max_rounds = 60 if self.D in (2, 4) else 100000
        T, npar = P.shape
        ph = np.empty((int(n_steps), T, npar))
        fh = np.empty((int(n_steps), int(n_sweeps), T))
        self.set_roto_rule(rule)
        L.check(self._lib.qmps_evolve_rotosolve(self._ctx, T, int(kind), npar, _f64(P), _f64(WW.view(np.float64)), int(n_steps),
                                                int(n_sweeps), 6 if double_frequency else 3, int(max_rounds), float(tol),
                                                _f64(ph), _f64(fh)))
        self.B = T
        return P, ph, fh

    def set_evolve_groups(self, groups=0):
        """Lock-step groups of `evolve_bfgs` (qmps_set_evolve_groups): 0 = automatic, 1 = one lock-step over all trajectories, K = K groups."""
        L.check(self._lib.qmps_set_evolve_groups(self._ctx, int(groups)))

    def evolve_groups(self, T):
        """The number of lock-step groups an `evolve_bfgs` call with T trajectories uses (qmps_get_evolve_groups)."""
        k = ctypes.c_int(0)
        L.check(self._lib.qmps_get_evolve_groups(self._ctx, int(T), ctypes.byref(k)))
        return k.value

    def evolve_bfgs(self, kind, params, WW, n_steps=1, maxiter=200, gtol=1e-5, h=1e-6, c1=1e-4,
                    alphas=(1.0, 0.5, 0.25, 0.125, 1 / 16, 1 / 64, 1 / 256, 1 / 4096), carry_hessian=False, hess_inv=None, warm=False,
                    max_rounds=None, tol=1e-12, tight_gradient=False, counters=True, adaptive_gradient=False, time_steps=False):
        """Time evolution by lock-step BFGS, every time step of every trajectory in ONE C call (qmps_evolve_bfgs): params (T, P) ->
        dict(x (T, P), params_hist (n_steps, T, P), fun (n_steps, T) [and fun_start: at the start of each time step], nit (n_steps,), hess_inv (T, P, P), gradient_batches,
        ladder_batches, nfev, gradient_ms).  warm=True continues a previous call on this engine (resident fixed points; with
        carry_hessian also `hess_inv`).  tight_gradient: the eigen-solves of the gradient batches iterate to tol instead of
        max(tol, 1e-8) (their objective comes from the two-sided quotient either way).  counters=False: no batch counts and no
        HIP events around the gradient batches (a pair of event records costs the stream ~12 us per batch).
        adaptive_gradient (D = 8, 16): QMPS_BFGS_ADAPTIVE_GRADIENT - a trajectory's gradient solves stop at clamp(1e-3 max|g|, max(tol, 1e-8), 1e-6).
        time_steps (with counters; D = 8, 16): 'gradient_ms' is the DEVICE time of the call, one event pair per time step, the run un-instrumented."""
        P = np.array(np.atleast_2d(params), dtype=np.float64, order='C', copy=True)
        WW = np.ascontiguousarray(WW, dtype=np.complex128).reshape(4, 4)
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        if max_rounds is None:
            max_rounds = 60 if self.D in (2, 4) else 100000
        T, npar = P.shape
        ph = np.empty((int(n_steps), T, npar))
        fh = np.empty((int(n_steps), 2, T))
        nit = np.zeros(int(n_steps), dtype=np.int32)
        cnt = np.zeros(4)
        if warm and carry_hessian and hess_inv is None:
            raise ValueError('a continued evolution with carried inverse Hessians needs hess_inv (the previous call\'s)')
        Hinv = np.zeros((T, npar, npar))
        if hess_inv is not None:
            Hinv[...] = hess_inv
        flags = ((L.BFGS_CARRY_HESSIAN if carry_hessian else 0) | (L.BFGS_WARM if warm else 0) | (L.BFGS_TIGHT_GRADIENT if tight_gradient else 0) |
                 (L.BFGS_ADAPTIVE_GRADIENT if adaptive_gradient else 0) | (L.BFGS_TIME_STEPS if time_steps else 0))
        # (through the versioned option structs: the positional qmps_evolve_bfgs is the same code)
        opts = L.EvolveOpts()
        L.check(self._lib.qmps_evolve_opts_init(byref(opts)))
        opts.n_steps, opts.maxiter, opts.n_alphas, opts.flags, opts.max_rounds = int(n_steps), int(maxiter), len(al), flags, int(max_rounds)
        opts.gtol, opts.h, opts.c1, opts.tol, opts.alphas = float(gtol), float(h), float(c1), float(tol), _f64(al)
        out = L.EvolveOut(size=ctypes.sizeof(L.EvolveOut), hinv=_f64(Hinv), params_hist=_f64(ph), f_hist=_f64(fh), nit=_i32(nit),
                          counters=_f64(cnt) if counters else None)
        L.check(self._lib.qmps_evolve_bfgs_opts(self._ctx, T, int(kind), npar, _f64(P), _f64(WW.view(np.float64)), byref(opts), byref(out)))
        self.B = T
        return {'x': P, 'params_hist': ph, 'fun': fh[:, 1], 'fun_start': fh[:, 0], 'nit': nit, 'hess_inv': Hinv, 'gradient_batches': int(cnt[0]),
                'ladder_batches': int(cnt[1]), 'nfev': int(cnt[2]), 'gradient_ms': float(cnt[3])}

    def evolve_bfgs_device(self, kind, params, WW, n_steps=1, maxiter=200, gtol=1e-5, h=1e-6, c1=1e-4,
                           alphas=(1.0, 0.5, 0.25, 0.125, 1 / 16, 1 / 64, 1 / 256, 1 / 4096), carry_hessian=False, hess_inv=None,
                           max_rounds=None, tol=1e-12, counters=True, tight_gradient=False, adaptive_gradient=False):
        """D = 2, 4 (and, as an option measured slower than `evolve_bfgs`, D = 16): the whole BFGS time evolution in ONE LAUNCH, one wave (D = 2) /
        one workgroup (D = 4: eight waves; D = 16: eight waves on a compute unit of their own) per trajectory, the optimiser on the device
        (qmps_evolve_bfgs_device): every trajectory advances at its own pace, no host round trip per iteration.  Same iteration
        as `evolve_bfgs`.  tight_gradient / adaptive_gradient: D = 16 only (the tolerance of a gradient's two solves, as `evolve_bfgs`).  Returns dict(x (T, P), params_hist (n_steps, T, P), fun / fun_start (n_steps, T), nit (n_steps, T) per
        trajectory, hess_inv, nfev, failed_evaluations, kernel_ms)."""
        P = np.array(np.atleast_2d(params), dtype=np.float64, order='C', copy=True)
        WW = np.ascontiguousarray(WW, dtype=np.complex128).reshape(4, 4)
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        if max_rounds is None:
            max_rounds = 100000 if self.D == 16 else 60
        T, npar = P.shape
        ph = np.empty((int(n_steps), T, npar))
        fh = np.empty((int(n_steps), 2, T))
        nit = np.zeros((int(n_steps), T), dtype=np.int32)
        cnt = np.zeros(4)
        Hinv = np.zeros((T, npar, npar))
        if hess_inv is not None:
            Hinv[...] = hess_inv
        flags = (L.BFGS_CARRY_HESSIAN if carry_hessian else 0) | (L.BFGS_WARM if (carry_hessian and hess_inv is not None) else 0)
        if self.D == 16:      # (the gradient's two solves: qmps_hip.h)
            flags |= (L.BFGS_TIGHT_GRADIENT if tight_gradient else 0) | (L.BFGS_ADAPTIVE_GRADIENT if (adaptive_gradient and not tight_gradient) else 0)
        L.check(self._lib.qmps_evolve_bfgs_device(self._ctx, T, int(kind), npar, _f64(P), _f64(WW.view(np.float64)), int(n_steps), int(maxiter),
                                                  float(gtol), float(h), float(c1), len(al), _f64(al), flags, int(max_rounds), float(tol),
                                                  _f64(Hinv), _f64(ph), _f64(fh), _i32(nit), _f64(cnt) if counters else None))
        self.B = T
        return {'x': P, 'params_hist': ph, 'fun': fh[:, 1], 'fun_start': fh[:, 0], 'nit': nit, 'hess_inv': Hinv, 'nfev': int(cnt[0]),
                'failed_evaluations': int(cnt[1]), 'kernel_ms': float(cnt[2]), 'squarings': int(cnt[3])}

    def overlap_results(self, B=None, want_r=False):
        B = self.B if B is None else B
        eta = np.empty(B, dtype=np.complex128)
        r = np.empty((B, self.D, self.D), dtype=np.complex128) if want_r else None
        rounds = np.empty(B, dtype=np.int32)
        st = np.empty(B, dtype=np.int32)
        L.check(self._lib.qmps_overlap_get(self._ctx, B, _f64(eta.view(np.float64)), None if r is None else _f64(r.view(np.float64)),
                                           _i32(rounds), _i32(st)))
        return (eta, rounds, st, r) if want_r else (eta, rounds, st)

    def opt_env_objective(self, params, h, k=1.0, want_parts=False):
        """Variational-environment objective (D = 2, 30 angles per row): f (B,) [and parts (B,4)]."""
        P = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        if P.shape[1] != 30:
            raise ValueError('the variational-environment objective takes 30 parameters')
        h = np.ascontiguousarray(h, dtype=np.complex128).reshape(4, 4)
        f = np.empty(P.shape[0])
        parts = np.empty((P.shape[0], 4)) if want_parts else None
        L.check(self._lib.qmps_opt_env_objective(self._ctx, P.shape[0], _f64(P), _f64(h.view(np.float64)), float(k), _f64(f),
                                                 None if parts is None else _f64(parts)))
        return (f, parts) if want_parts else f

    # -- timing / probes ----------------------------------------------------------------------
    def timer_begin(self):
        L.check(self._lib.qmps_timer_begin(self._ctx))

    def timer_end(self):
        ms = c_float(0)
        L.check(self._lib.qmps_timer_end(self._ctx, byref(ms)))
        return ms.value

    def kernel_time(self, n_last=1):
        """(average ms, kernel name) of the dominant kernel over the last n_last launches."""
        ms = c_float(0)
        name = ctypes.create_string_buffer(128)
        L.check(self._lib.qmps_kernel_time(self._ctx, int(n_last), byref(ms), name, 128))
        return ms.value, name.value.decode()

    def probe_fp64_tflops(self):
        v = c_double(0)
        L.check(self._lib.qmps_probe_fp64_peak(self._ctx, byref(v)))
        return v.value

    def probe_fp64_mfma_tflops(self, waves_per_simd=1):
        v = c_double(0)
        L.check(self._lib.qmps_probe_fp64_mfma_peak(self._ctx, int(waves_per_simd), byref(v)))
        return v.value

    def probe_hbm_gbps(self):
        v = c_double(0)
        L.check(self._lib.qmps_probe_hbm_peak(self._ctx, byref(v)))
        return v.value

    # -- multi-GPU ----------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(L.UNIQUE_ID_BYTES)
        L.check(L.load().qmps_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        L.check(self._lib.qmps_comm_init(self._ctx, unique_id, int(rank), int(nranks)))

    def comm_count(self):
        """Ranks that joined the communicator (1 without one)."""
        n = ctypes.c_int(0)
        L.check(self._lib.qmps_comm_count(self._ctx, byref(n)))
        return n.value

    def comm_destroy(self):
        L.check(self._lib.qmps_comm_destroy(self._ctx))

    def allreduce_sum(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        L.check(self._lib.qmps_allreduce_sum(self._ctx, _f64(v), v.size))
        return v

    def allreduce_min(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        L.check(self._lib.qmps_allreduce_min(self._ctx, _f64(v), v.size))
        return v

    def cost_launch(self, B=None):
        """Asynchronous: device-side sum over the batch (+ one RCCL all-reduce when a communicator exists)."""
        L.check(self._lib.qmps_cost_launch(self._ctx, self.B if B is None else B))

    def get_cost(self):
        cost = np.empty(max(self.n_terms, 1))
        L.check(self._lib.qmps_get_cost(self._ctx, _f64(cost)))
        return cost[:self.n_terms]

    def allreduce_cost(self, B=None):
        cost = np.empty(max(self.n_terms, 1))
        L.check(self._lib.qmps_allreduce_cost(self._ctx, self.B if B is None else B, _f64(cost)))
        return cost[:self.n_terms]
