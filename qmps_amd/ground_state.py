"""Host-side mirror of `qmps/ground_state.py` for the noiseless energy path.

  Hamiltonian                              qmps/ground_state.py:66-118
  SparseFullEnergyOptimizer                qmps/ground_state.py:120-168  (exact-environment objective)
  NonSparseFullEnergyOptimizer             qmps/ground_state.py:230-269
  NonSparseFullTwoSiteEnergyOptimizer      qmps/ground_state.py:271-335
  ground_state_sweep                       the drivers' loops over couplings x ansatz sizes x restarts (scripts/ground_state_finding.py:166-200)

Every objective keeps the reference's contract - `objective_function(params) -> float`, previous
value returned when the environment is not positive definite (the reference's caught
`LinAlgError`, :153-157) - and gains `batch_objective_function(params[B,P]) -> float[B]`, one
launch of the MI355X kernel for the whole batch (rotosolve shifts x restarts).

Energies are computed in fp64.  (The reference simulates this path in cirq's default single
precision, ground_state.py:160,165, so its own values carry ~1e-7 noise.)
"""
from functools import reduce
from itertools import product

import numpy as np
from scipy.linalg import expm

from . import _runtime
from ._lib import STATUS_NOT_PD, STATUS_OK
from .represent import FullStateTensor, ShallowCNOTStateTensor, build_gate, final_state, unitary
from .tools import Optimizer, get_env_exact

π = np.pi

Sx = np.array([[0, 1], [1, 0]], dtype=complex)
Sy = np.array([[0, -1j], [1j, 0]], dtype=complex)
Sz = np.array([[1, 0], [0, -1]], dtype=complex)
S = {'I': np.eye(2), 'X': Sx, 'Y': Sy, 'Z': Sz}


def paulis(_spin=0.5):
    """Pauli matrices with +-1 eigenvalues - what `xmps.spin.paulis(0.5)` yields at
    ground_state.py:29 (pinned by the 4x4 TFIM known answer, tests/test_ground_state.py:26-38)."""
    return Sx, Sy, Sz


def swap():
    return np.eye(4)[[0, 2, 1, 3]].astype(complex)


def _su_generators(N):
    """Generalised Gell-Mann basis of su(N): N^2 - 1 traceless Hermitian matrices."""
    gens = []
    for a in range(N):
        for b in range(a + 1, N):
            m = np.zeros((N, N), dtype=complex)
            m[a, b] = m[b, a] = 1
            gens.append(m)
            m = np.zeros((N, N), dtype=complex)
            m[a, b], m[b, a] = -1j, 1j
            gens.append(m)
    for k in range(1, N):
        m = np.zeros((N, N), dtype=complex)
        m[np.arange(k), np.arange(k)] = 1
        m[k, k] = -k
        gens.append(m * np.sqrt(2.0 / (k * (k + 1))))
    return gens


_GEN_CACHE = {}


def SU(params, N):
    """exp(-i sum_k p_k G_k / 2) with G_k the generalised Gell-Mann matrices: a smooth
    parameterisation of SU(N) by N^2 - 1 reals.  The reference takes `SU` from xmps.spin, whose
    convention is not visible in the reference tree ("unpinned"): any onto smooth map
    gives the same optimisation problem, and the batched entry points accept unitaries directly."""
    G = _su_basis(N)
    params = np.asarray(params, dtype=float)
    if params.shape[0] != N * N - 1:
        raise ValueError(f'SU({N}) takes {N * N - 1} parameters, got {params.shape[0]}')
    return expm(-0.5j * np.tensordot(params, G, axes=1))


def U4(params):
    return SU(params, 4)


def _su_basis(N):
    if N not in _GEN_CACHE:
        _GEN_CACHE[N] = np.stack(_su_generators(N))
    return _GEN_CACHE[N]


def extractv(U):
    """The su(N) coordinates of a unitary: SU(extractv(U), N) == U up to a global phase (principal matrix logarithm; the generators of this
    module's `SU`, tr G_a G_b = 2 delta_ab).  Stand-in for `xmps.spin.extractv` as `scripts/bond_dimension.py:35` uses it."""
    from scipy.linalg import logm
    U = np.asarray(U, dtype=complex)
    N = U.shape[0]
    U = U / np.linalg.det(U) ** (1.0 / N)                 # special unitary: the phase never reaches an energy
    Hm = 2j * logm(U)                                    # U = exp(-i H / 2)
    Hm = 0.5 * (Hm + Hm.conj().T)
    return np.real(np.einsum('kab,ba->k', _su_basis(N), Hm)) / 2.0


def insu2N(v):
    """A vector of su(n) as the vector of su(2n) that generates U x 1 - the same unitary beside a new, untouched LAST qubit
    (`xmps.spin.insu2N`, scripts/bond_dimension.py:10-12: "returns the same vector in su(2n)"): SU(insu2N(v), 2n) == kron(SU(v, n), 1_2)."""
    v = np.asarray(v, dtype=float)
    n = int(round(np.sqrt(len(v) + 1)))
    if n * n - 1 != len(v):
        raise ValueError(f'{len(v)} parameters are not su(n)')
    Hn = np.tensordot(v, _su_basis(n), axes=1)
    return np.real(np.einsum('kab,ba->k', _su_basis(2 * n), np.kron(Hn, np.eye(2)))) / 2.0


def embed_bond_dimension(v, eps=4e-2):
    """The optimum of bond dimension D as a starting point at 2 D - the `fixindices(insu2N(v))` of `scripts/bond_dimension.py:21-35, 50`:
    v in su(2D) -> su(4D) (a new qubit beside U: `insu2N`), perturbed by eps in every coordinate ("gets away from singular points": the
    embedded tensor A x 1 has a degenerate transfer spectrum), the new qubit swapped with the physical one so that it becomes the last BOND
    qubit.  In THIS module's register layout (rows of U: left bond | physical, `unitary_to_tensor`) that swap acts on the OUTPUT side,
    U' = S12 (U x 1): A'[s, (i,a), (j,b)] = A[s,i,j] delta_ab - the same iMPS, the same energy per site at eps = 0 (tests/test_host_api.py).
    (The reference multiplies from the right, `U @ S12`, in xmps's layout of `SU` / `insu2N`, which is not in its tree; the intent - "swap makes
    i, j have same tensor product structure" - is this embedding.)"""
    w = insu2N(v)
    N = int(round(np.sqrt(len(w) + 1)))
    U = SU(w + eps, N)
    S12 = np.kron(np.eye(N // 4), swap())
    return extractv(S12 @ U)


class Hamiltonian:
    """Two-site Hamiltonian from Pauli strings; a one-letter key 'X': g means g/2 (IX + XI)
    (ground_state.py:66-88)."""

    def __init__(self, strings=None):
        self.strings = strings
        if strings is not None:
            for key, val in list(self.strings.items()):
                if len(key) == 1:
                    self.strings['I' + key] = val / 2
                    self.strings[key + 'I'] = val / 2
                    self.strings.pop(key)

    def to_matrix(self):
        assert self.strings is not None
        h = np.zeros((4, 4), dtype=complex)
        for js, J in self.strings.items():
            h += J * reduce(np.kron, [S[j] for j in js])
        self._matrix = h
        return h

    def term_matrices(self):
        """One 4x4 matrix per Pauli string (J included): the per-term batch axis of the kernel."""
        return {js: J * reduce(np.kron, [S[j] for j in js]) for js, J in self.strings.items()}

    def from_matrix(self, mat):
        keys = list(S.keys())
        self.strings = {a + b: np.trace(np.kron(S[a], S[b]) @ mat) / 4 for a, b in product(keys, keys)}
        del self.strings['II']
        return self

    def calculate_energy(self, ops, n_qubits, loc=0):
        """<psi| 1^loc x H x 1^rest |psi> for the state prepared by `ops` on |0..0>
        (ground_state.py:110-118, with this package's circuit model instead of cirq).  API helper for
        arbitrary user circuits (host numpy); no optimiser of the hot path calls it."""
        psi = final_state(ops, n_qubits)
        H = reduce(np.kron, [np.eye(2)] * loc + [self.to_matrix()] + [np.eye(2)] * (n_qubits - loc - 2))
        return float(np.real(psi.conj() @ H @ psi))


def _as_h(H):
    return H.to_matrix() if isinstance(H, Hamiltonian) else np.asarray(H, dtype=complex)


class _GpuEnergyMixin:
    """Shared plumbing: unitaries -> libqmps_hip -> energies, with the reference's error behaviour."""
    max_iter = 10000
    env_tol = 1e-13

    def _energies_from_unitaries(self, U):
        U = np.ascontiguousarray(U, dtype=np.complex128)
        eng = _runtime.engine(self.D, U.shape[0])
        E, it, st = eng.energies(U, _as_h(self.H), kind='unitary', max_iter=self.max_iter, tol=self.env_tol)
        return E[:, 0], it, st


class SparseFullEnergyOptimizer(_GpuEnergyMixin, Optimizer):
    """Shallow-ansatz energy optimiser (ground_state.py:120-168).  `H` is the 4x4 matrix (or a
    Hamiltonian); `state_tensor(D, params)` any gate class of represent.py."""

    def __init__(self, H, D=2, depth=2, state_tensor=ShallowCNOTStateTensor, optimize_environment=False,
                 env_depth=4, initial_guess=None, settings=None):
        self.optimize_environment = optimize_environment
        self.env_depth = env_depth
        self.state_tensor = state_tensor
        self.H = H
        self.D = D
        self.d = 2
        if optimize_environment:
            if D != 2:
                raise ValueError('the variational-environment objective is D = 2 only (ground_state.py:176-177)')
            initial_guess = np.random.randn(30) if initial_guess is None else initial_guess
            self.objective_function = self.objective_function_opt_environment
        else:
            initial_guess = np.array([np.random.randn(), np.random.randn()] * depth) if initial_guess is None \
                else initial_guess
            self.objective_function = self.objective_function_exact_environment
        self.p = len(initial_guess)
        self.f = 0
        super().__init__(build_gate(self.state_tensor, D, initial_guess), None, initial_guess)
        if settings:
            self.change_settings(settings)

    def unitaries(self, params_batch):
        return np.stack([unitary(build_gate(self.state_tensor, self.D, p)) for p in np.atleast_2d(params_batch)])

    def _energies_from_params(self, params_batch):
        """params -> energies.  For the ansatz classes libqmps_hip knows (`device_kind`), the circuit is
        simulated on the GPU: 8 P bytes per evaluation cross PCIe instead of a 2D x 2D
        unitary built gate by gate on the host; any other gate class goes through `unitary()`."""
        P = np.ascontiguousarray(np.atleast_2d(params_batch), dtype=np.float64)
        kind = getattr(self.state_tensor, 'device_kind', None)
        if kind is None or (kind in (2, 6) and self.D != 2):
            return self._energies_from_unitaries(self.unitaries(P))
        eng = _runtime.engine(self.D, P.shape[0])
        E, it, st = eng.energies_from_params(kind, P, _as_h(self.H), max_iter=self.max_iter, tol=self.env_tol)
        return E[:, 0], it, st

    def objective_function_exact_environment(self, u_params):
        E, _, st = self._energies_from_params(u_params)
        if st[0] != STATUS_OK:
            # the reference catches cholesky's LinAlgError, prints and returns the previous value
            print('LinAlgError')
            return self.f
        self.f = float(E[0])
        return self.f

    def objective_function_opt_environment(self, params):
        """No eigen-solve: U and V are both variational; energy + k (tr rho_u^2 + tr rho_v^2 - 2 tr rho_u rho_v),
        k = 1 (ground_state.py:170-228).  Four circuits, simulated on the device."""
        assert len(params) == 30
        return float(self.batch_objective_function(np.asarray(params, dtype=float)[None])[0])

    def batch_objective_function(self, params_batch):
        """One kernel launch for B parameter vectors.  Entries whose environment is not positive
        definite / not converged are NaN (there is no 'previous value' in a batch)."""
        if self.optimize_environment:
            P = np.atleast_2d(np.asarray(params_batch, dtype=float))
            return _runtime.engine(2, P.shape[0]).opt_env_objective(P, _as_h(self.H), k=1.0)
        E, _, st = self._energies_from_params(params_batch)
        return np.where(st == STATUS_OK, E, np.nan)

    def _device_double_rotosolve(self, n_sweeps):
        """`Optimizer.optimize()` with settings['method'] == 'Rotosolve' (tools.py:248-270 -> double_rotosolve,
        tools.py:422-457): when libqmps_hip can simulate the ansatz, the whole run - six shifted evaluations per
        parameter, sinusoid fit, argmin, update, sweep energies - happens in one C call, without a host round trip per
        parameter.  Returns None (host driver takes over) for gate classes without a device implementation."""
        from .tools import RotosolveResult
        kind = getattr(self.state_tensor, 'device_kind', None)
        if self.optimize_environment or kind is None or kind > 3 or (kind == 2 and self.D != 2):
            return None          # (the device rotosolve drivers know the layered families 0 .. 3; other gate classes: host driver, batched objective)
        from .rotosolve import device_double_rotosolve
        es, P = device_double_rotosolve(self, np.asarray(self.initial_guess, dtype=float)[None], n_sweeps)
        # the reference updates the caller's parameter vector in place, element by element (tools.py:453): that works for
        # lists as well as arrays (a tuple cannot be updated in place there either)
        if isinstance(self.initial_guess, np.ndarray):
            self.initial_guess[...] = P[0]
        elif isinstance(self.initial_guess, list):
            self.initial_guess[:] = [float(v) for v in P[0]]
        else:
            self.initial_guess = P[0].copy()
        hist = [float(e) for e in es[:, 0]]
        self.f = hist[-1]
        return RotosolveResult(hist, hist[-1], self.initial_guess, '')

    def update_state(self):
        self.u = build_gate(self.state_tensor, self.D, self.optimized_result.x)
        self.U = unitary(self.u)


class NonSparseFullEnergyOptimizer(_GpuEnergyMixin, Optimizer):
    """Full SU(2D) parameterisation (ground_state.py:230-269).  `get_env_function` is kept for API
    compatibility: when it is the default the environment is solved inside the energy kernel; a
    user-supplied function U -> V supplies the environment and the energy is still evaluated on the device."""

    def __init__(self, H, D=2, get_env_function=get_env_exact, initial_guess=None, settings=None):
        self.env_function = get_env_function
        self.H = H
        self.D = D
        self.d = 2
        initial_guess = np.random.randn((2 * D) ** 2 - 1) if initial_guess is None else initial_guess
        super().__init__(FullStateTensor(SU(initial_guess, 2 * D)), None, initial_guess)
        if settings:
            self.change_settings(settings)

    def objective_function(self, u_params):
        self.U = U = SU(u_params, 2 * self.D)
        if self.env_function is not get_env_exact:
            # injected environment (ground_state.py:238,254): V = env_function(U) is the caller's business; its
            # first column is vec(L)/||L|| (tools.py:97-108), so r = L L^+ goes to the device as the resident
            # environment and the energy is evaluated there (energy-only kernel) - no host-side simulation
            V = np.asarray(self.env_function(U))
            Lm = V[:, 0].reshape(self.D, self.D)
            eng = _runtime.engine(self.D, 1)
            eng.set_unitaries(U[None])
            eng.set_hamiltonian(_as_h(self.H))
            eng.set_env_guess((Lm @ Lm.conj().T)[None])
            eng.launch_energy_only(1)
            return float(eng.results(1)[0][0, 0])
        E, _, st = self._energies_from_unitaries(U[None])
        if st[0] == STATUS_NOT_PD:
            raise np.linalg.LinAlgError('environment is not positive definite')  # uncaught in the reference too
        return float(E[0])

    def batch_objective_function(self, params_batch):
        """B parameter vectors -> energies; the unitaries SU(p, 2D) are built ON THE DEVICE (qmps_set_states_su: scaling and
        squaring, one workgroup per evaluation) - no host matrix exponential per row, 8 ((2D)^2 - 1) bytes per evaluation over PCIe."""
        P = np.ascontiguousarray(np.atleast_2d(params_batch), dtype=np.float64)
        eng = _runtime.engine(self.D, P.shape[0])
        E, _, st = eng.energies_from_su(P, _as_h(self.H), max_iter=self.max_iter, tol=self.env_tol)
        return np.where(st == STATUS_OK, E[:, 0], np.nan)

    def update_state(self):
        self.U = SU(self.optimized_result.x, 2 * self.D)


class NonSparseFullTwoSiteEnergyOptimizer(Optimizer):
    """Two-site unit cell, D = 2 only (ground_state.py:271-335): 30 parameters, U1 = SU(p[:15], 4),
    U2 = SU(p[15:], 4), f = (E1 + E2)/2."""
    D = 2

    def __init__(self, H, initial_guess=None):
        self.H = H
        super().__init__(initial_guess=np.random.randn(30) if initial_guess is None else initial_guess)

    def _cell(self, U1, U2):
        eng = _runtime.engine(2, U1.shape[0])
        E, it, st = eng.cell2_energies(U1, U2, _as_h(self.H))
        return E[:, 0], it, st

    def objective_function(self, u_params):
        self.U1 = U1 = SU(u_params[:15], 4)
        self.U2 = U2 = SU(u_params[15:], 4)
        E, _, st = self._cell(U1[None], U2[None])
        if st[0] == STATUS_NOT_PD:
            raise np.linalg.LinAlgError('environment is not positive definite')
        return float(E[0])

    def batch_objective_function(self, params_batch):
        """U1 = U4(p[:15]), U2 = U4(p[15:]) for every row ON THE DEVICE (qmps_cell2_energy_batch_su)."""
        P = np.ascontiguousarray(np.atleast_2d(params_batch), dtype=np.float64)
        eng = _runtime.engine(2, P.shape[0])
        E, _, st = eng.cell2_energies_su(P, _as_h(self.H))
        return np.where(st == STATUS_OK, E[:, 0], np.nan)

    def update_state(self):
        self.u1 = SU(self.optimized_result.x[:15], 4)
        self.u2 = SU(self.optimized_result.x[15:], 4)


def ground_state_sweep(terms, coefficients, D=2, depth=2, state_tensor=ShallowCNOTStateTensor, restarts=8, initial_guesses=None, rng=None,
                       maxiter=200, gtol=1e-6, return_all=False):
    """Variational ground states of K Hamiltonians  H_k = sum_q coefficients[k, q] terms[q]  from R random restarts each - K R BFGS
    minimisations advancing in ONE lock-step over device batches.

    This is the reference's phase-diagram driver (`scripts/ground_state_finding.py:166-200`: for 21 couplings lambda and three ansatz sizes,
    `minimize(lambda p: eps(p, lambda), randn(n), method='BFGS', tol=1e-10)`, again from a fresh start while the energy does not improve) and
    its error-against-depth driver (`scripts/noisy_optimization.py:30-72`) with the loops turned into the batch axis: a launch returns the
    energies of every TERM for every candidate (`qmps_energy_batch_ansatz` with n_terms Hamiltonians: one environment solve per candidate,
    the terms share it), and trajectory (k, r) combines them with its own coefficients - the couplings cost nothing extra.
      terms         (Q, 4, 4) two-site operators (or Hamiltonian objects), e.g. [-ZZ, (XI + IX)/2] for the TFIM
      coefficients  (K, Q)
      initial_guesses (K, R, P) or None: `rng.standard_normal` (the reference's `np.random.randn`), P = 2 depth angles (3 depth for ShallowCNOTStateTensor3)
    Returns dict(energy (K,) best of the restarts, params (K, P) its parameters, energies (K, R), nit, nfev[, all_params (K, R, P)])."""
    from .tools import batched_bfgs
    T = np.stack([_as_h(t_) for t_ in terms]).astype(np.complex128)
    C = np.atleast_2d(np.asarray(coefficients, dtype=float))
    K, Q = C.shape
    if T.shape[0] != Q:
        raise ValueError(f'{Q} coefficients per Hamiltonian for {T.shape[0]} terms')
    kind = getattr(state_tensor, 'device_kind', None)
    on_device = kind is not None and not (kind in (2, 6) and D != 2)
    if initial_guesses is None:
        rng = np.random.default_rng() if rng is None else rng
        per_layer = 3 if state_tensor.__name__ == 'ShallowCNOTStateTensor3' else 2
        n_par = 15 if state_tensor.__name__ == 'ShallowFullStateTensor' else per_layer * depth
        X0 = rng.standard_normal((K, restarts, n_par))
    else:
        X0 = np.array(initial_guesses, dtype=float)
        if X0.ndim != 3 or X0.shape[0] != K:
            raise ValueError('initial_guesses: expected (K, R, P)')
    _, R, P = X0.shape
    n_traj = K * R
    coef_traj = np.repeat(C, R, axis=0)                      # trajectory k R + r -> the coefficients of H_k

    def batch(cand):
        cand = np.ascontiguousarray(cand, dtype=np.float64)
        per = cand.shape[0] // n_traj                        # rows are trajectory-major (batched_bfgs)
        eng = _runtime.engine(D, cand.shape[0])
        if on_device:
            E, _, st = eng.energies_from_params(kind, cand, T, max_iter=_GpuEnergyMixin.max_iter, tol=_GpuEnergyMixin.env_tol)
        else:
            U = np.stack([unitary(build_gate(state_tensor, D, p_)) for p_ in cand])
            E, _, st = eng.energies(U, T, kind='unitary', max_iter=_GpuEnergyMixin.max_iter, tol=_GpuEnergyMixin.env_tol)
        f = (E * np.repeat(coef_traj, per, axis=0)).sum(axis=1)
        return np.where(st == STATUS_OK, f, np.nan)

    res = batched_bfgs(batch, batch, X0.reshape(n_traj, P), maxiter=maxiter, gtol=gtol)
    fun = np.where(np.isfinite(res['fun']), res['fun'], np.inf).reshape(K, R)
    Xf = res['x'].reshape(K, R, P)
    best = fun.argmin(axis=1)
    out = {'energy': fun[np.arange(K), best], 'params': Xf[np.arange(K), best], 'energies': fun, 'nit': res['nit'], 'nfev': res['nfev'],
           'converged': res['converged'].reshape(K, R)}
    if return_all:
        out['all_params'] = Xf
    return out
