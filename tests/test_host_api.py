"""CPU: the host-side mirror of the reference API (qmps_amd.tools / represent / ground_state /
rotosolve / time_evolve_tools) against the reference-generated golden vectors and the oracle."""
import numpy as np
import pytest

from oracle import qmps_oracle as O
from qmps_amd import ground_state as G
from qmps_amd import represent as R
from qmps_amd import rotosolve as RS
from qmps_amd import time_evolve_tools as TT
from qmps_amd import tools as T


@pytest.mark.parametrize('D', [2, 4, 8, 16])
def test_unitary_to_tensor(D, golden):
    for U, A in zip(golden[f'U_D{D}'], golden[f'ref_A_D{D}']):
        assert np.array_equal(T.unitary_to_tensor(U), A)


def test_tensor_to_unitary_roundtrip(golden):
    """tests/test_tools.py:15-20."""
    for A in golden['ref_A_D2']:
        U, passed = T.tensor_to_unitary(A, testing=True)
        assert passed and np.allclose(T.unitary_to_tensor(U), A)
    for A in golden['ref_A_D4']:
        U, passed = T.tensor_to_unitary(A, testing=True)
        assert passed


def test_unitary_to_tensor_shape_from_four_qubit_unitary():
    """tests/test_tools.py:22-31: a 4-qubit unitary gives a (2, 8, 8) tensor."""
    U = T.haar_unitary(16, np.random.default_rng(0))
    assert T.unitary_to_tensor(U).shape == (2, 8, 8)


@pytest.mark.parametrize('D', [2, 4, 8])
def test_environment_embeddings(D, golden):
    for L, V_ref in zip(golden[f'oracle_L_D{D}'], golden[f'ref_V_D{D}']):
        V = T.environment_to_unitary(L)
        assert np.allclose(V.conj().T @ V, np.eye(D * D)) and np.allclose(V[:, 0], V_ref[:, 0])
        assert np.allclose(T.environment_from_unitary(V), L / np.linalg.norm(L))
    assert np.allclose(T.environment_from_unitary(golden['ref_V_D2'][0]), golden['ref_env_from_unitary'])


def test_small_helpers(golden):
    v = golden['realvec_in']
    assert np.allclose(T.from_real_vector(v), golden['ref_from_real_vector'])
    assert np.allclose(T.to_real_vector(golden['ref_V_D2'][0]), golden['ref_to_real_vector'])
    assert T.split_2s([1, 2, 3, 4]) == [[1, 2], [3, 4]] and T.split_3s(list(range(6))) == [[0, 1, 2], [3, 4, 5]]
    A, B = np.ones((2, 2)), 2 * np.ones((1, 1))
    assert np.array_equal(T.direct_sum(A, B), np.array([[1, 1, 0], [1, 1, 0], [0, 0, 2]]))
    Q = T.random_unitary(4, 4)
    assert np.isrealobj(Q) and np.allclose(Q.T @ Q, np.eye(4))          # tools.py:36-37 is a REAL QR
    iso = T.haar_unitary(4, np.random.default_rng(1))[:, :2]
    Uext = T.unitary_extension(iso)
    assert np.allclose(Uext.conj().T @ Uext, np.eye(4)) and np.allclose(Uext[:, :2], iso)
    assert T.unitary_extension(iso, D=6).shape == (6, 6)
    assert np.allclose(T.cT(iso), iso.conj().T)


def test_merge(golden):
    A2 = golden['ref_A_D2']
    for i in range(len(A2)):
        assert np.allclose(TT.merge(A2[i], A2[(i + 1) % len(A2)]), golden['ref_merge_D2'][i])
    A4 = golden['ref_A_D4']
    assert TT.merge(A4[0], A4[1]).shape == (4, 4, 4)


def test_env_site_embeddings():
    """qmps/time_evolve_tools.py:133-166 / new_time_evolve.py:53-100 self-tests: round trips + unitarity."""
    rng = np.random.default_rng(2)
    for _ in range(10):
        q = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
        A, n = TT.put_env_on_left_site(q, ret_n=True)
        assert np.allclose(TT.get_env_off_left_site(A) * n, q) and np.allclose(A.conj().T @ A, np.eye(4))
        A, n = TT.put_env_on_right_site(q, ret_n=True)
        assert np.allclose(TT.get_env_off_right_site(A) * n, q) and np.allclose(A.conj().T @ A, np.eye(4))


# ---- gates / ansatz -> unitary ---------------------------------------------------------------
def test_ansatz_unitaries_match_golden(golden):
    for p, U in zip(golden['cnot_params_D2'], golden['oracle_cnot_U_D2']):
        assert np.allclose(R.unitary(R.ShallowCNOTStateTensor(2, p)), U)
    for p, U in zip(golden['cnot_params_D4'], golden['oracle_cnot_U_D4']):
        assert np.allclose(R.unitary(R.ShallowCNOTStateTensor(4, p)), U)
    for p, U in zip(golden['full_params'], golden['oracle_full_U']):
        assert np.allclose(R.unitary(R.ShallowFullStateTensor(2, p)), U)


@pytest.mark.parametrize('cls,npar', [(R.ShallowQAOAStateTensor, 4), (R.ShallowCNOTStateTensor, 4),
                                      (R.ShallowCNOTStateTensor3, 6), (R.ExactAfter4, 6)])
@pytest.mark.parametrize('D', [2, 4])
def test_every_ansatz_is_unitary(cls, npar, D):
    p = np.random.default_rng(4).standard_normal(npar)
    U = R.unitary(cls(D, p))
    assert U.shape == (2 * D, 2 * D) and np.allclose(U.conj().T @ U, np.eye(2 * D))


def test_other_gates():
    rng = np.random.default_rng(6)
    U = R.unitary(R.ShallowCNOTStateTensor_nonuniform(4, rng.standard_normal(12)))
    assert np.allclose(U.conj().T @ U, np.eye(8))
    assert R.ShallowCNOTStateTensor_nonuniform.params_per_iter(4) == 6
    U = R.unitary(R.StateGate(rng.standard_normal(6)))
    assert np.allclose(U.conj().T @ U, np.eye(4))
    U = R.unitary(R.ShallowEnvironment(4, rng.standard_normal(4)))
    assert U.shape == (16, 16) and np.allclose(U.conj().T @ U, np.eye(16))
    # cirq's X**t: eigenvalue 1 on |+>, e^{i pi t} on |->
    Xt = R.unitary(R.x_pow(0.3))
    plus, minus = np.array([1, 1]) / np.sqrt(2), np.array([1, -1]) / np.sqrt(2)
    assert np.allclose(Xt @ plus, plus) and np.allclose(Xt @ minus, np.exp(0.3j * np.pi) * minus)
    assert np.allclose(R.unitary(R.zz_pow(0.5)), np.diag([1, 1j, 1j, 1]))
    T2 = R.FullStateTensor(T.haar_unitary(4, rng))
    assert np.allclose(R.unitary(T2 ** -1), T2.U.conj().T) and np.allclose((T2 ** 2).U, T2.U @ T2.U)


@pytest.mark.parametrize('D', [2, 4])
def test_state_wiring_matches_golden_statevector(D, golden):
    """represent.py:258-262 register layout: State(U, V, 2) on |0..0> equals the golden psi."""
    n = 2 + 2 * int(np.log2(D))
    for U, V, psi in zip(golden[f'U_D{D}'], golden[f'ref_V_D{D}'], golden[f'oracle_psi_D{D}']):
        st = R.State(R.FullStateTensor(U), R.FullEnvironment(V), 2)
        assert st.num_qubits() == n and st.bond_dim == D
        assert np.allclose(R.final_state(st(*R.line_qubits(n)), n), psi)
        # Hamiltonian.calculate_energy on that circuit (ground_state.py:110-118) with loc = log2 D
        H = G.Hamiltonian({'ZZ': -1, 'X': 1})
        e = H.calculate_energy(st(*R.line_qubits(n)), n, loc=int(np.log2(D)))
        assert abs(e - O.energy_statevector(U, golden['ref_h_tfim'], V)) < 1e-13


def test_power_circuit():
    """represent.py:235-248: K staggered copies of U on n + K - 1 qubits."""
    U = T.haar_unitary(4, np.random.default_rng(8))
    pc = R.FullStateTensor(U).raise_power(3)
    assert pc.num_qubits() == 4
    W = R.unitary(pc)
    I = np.eye(2)
    expect = np.kron(U, np.eye(4)) @ np.kron(I, np.kron(U, I)) @ np.kron(np.eye(4), U)
    assert np.allclose(W, expect)


# ---- Hamiltonian -----------------------------------------------------------------------------
def test_hamiltonian(golden):
    assert np.array_equal(G.Hamiltonian({'ZZ': -1, 'X': 1}).to_matrix(), golden['ref_h_tfim'])
    assert np.array_equal(G.Hamiltonian({'ZZ': -1, 'IX': 0.5, 'XI': 0.5}).to_matrix(), golden['ref_h_tfim'])
    assert np.allclose(G.Hamiltonian({'XX': 1, 'YY': 1, 'ZZ': 0.5}).to_matrix(), golden['ref_h_xxz'])
    H = G.Hamiltonian({'ZZ': -1, 'X': 0.7})
    assert set(H.strings) == {'ZZ', 'IX', 'XI'} and H.strings['IX'] == 0.35
    terms = H.term_matrices()
    assert np.allclose(sum(terms.values()), golden['ref_h_tfim_g07'])
    back = G.Hamiltonian().from_matrix(golden['ref_h_xxz'])
    assert np.allclose(back.to_matrix(), golden['ref_h_xxz'])
    assert np.allclose(G.paulis(0.5)[2], np.diag([1, -1]))


def test_SU_parameterisation():
    rng = np.random.default_rng(9)
    for N in (4, 8):
        U = G.SU(rng.standard_normal(N * N - 1), N)
        assert np.allclose(U.conj().T @ U, np.eye(N)) and abs(np.linalg.det(U) - 1) < 1e-10
    assert np.allclose(G.SU(np.zeros(15), 4), np.eye(4))
    with pytest.raises(ValueError):
        G.SU(np.zeros(3), 4)


# ---- optimiser plumbing (no GPU: objective injected) ---------------------------------------------
def test_optimizer_base_contract():
    opt = T.Optimizer(initial_guess=np.array([1.0, -2.0]), obj_fun=lambda p, a: float(np.sum((p - a) ** 2)),
                      args=(np.array([0.5, 0.25]),))
    opt.change_settings({'verbose': False, 'tol': 1e-12})
    res = opt.optimize()
    assert np.allclose(res.x, [0.5, 0.25], atol=1e-4) and len(opt.obj_fun_values) == opt.iters > 0


def test_double_rotosolve_scalar_and_batched_agree():
    """tools.py:422-457 on an exactly double-sinusoidal objective: one sweep reaches the coordinate
    minimum; the batched sampler gives the same trajectory as the ten scalar calls."""
    def eps(p):
        return float(np.sin(2 * p[0] + 0.3) + 0.5 * np.sin(p[0] - 1) + 0.7 * np.sin(p[1] + 0.2) * np.cos(p[0]))

    calls = []

    def batch(P):
        calls.append(len(P))
        return np.array([eps(p) for p in P])

    x1 = np.array([0.1, 0.2])
    x2 = x1.copy()
    r1 = T.double_rotosolve(eps, x1, N_iters=3, disp=False)
    r2 = T.double_rotosolve(eps, x2, N_iters=3, disp=False, batch_eps=batch)
    assert isinstance(r1, T.RotosolveResult) and r1.message == ''
    assert np.allclose(r1.x, r2.x) and np.allclose(r1.history, r2.history)
    assert r1.history[-1] <= r1.history[0] + 1e-12 and calls == [6] * 6
    assert r1.x is x1                                   # updated in place like the reference


def test_rotosolve_module_drivers():
    H = np.diag([1.0, -1.0]).astype(complex)

    def state(p):
        return np.array([np.cos(p[0] / 2), np.sin(p[0] / 2) * np.exp(1j * p[1])])

    es, hist = RS.rotosolve(H, state, np.array([0.3, 0.1]), N_iters=2)
    assert abs(es[-1] + 1) < 1e-12                      # <Z> = cos(theta) -> -1 in one update
    es2, p2 = RS.double_rotosolve(H, state, np.array([0.3, 0.1]), N_iters=2)
    assert abs(es2[-1] + 1) < 1e-9

    def batch_eps(P):
        return np.array([np.real(state(p).conj() @ H @ state(p)) for p in P])

    e3, p3 = RS.batched_rotosolve(batch_eps, np.array([[0.3, 0.1], [2.0, -1.0], [1.0, 0.5]]), N_iters=1)
    assert e3.shape == (1, 3) and np.allclose(e3[-1], -1)
    e4, p4 = RS.batched_double_rotosolve(batch_eps, np.array([[0.3, 0.1], [2.0, -1.0]]), N_iters=1)
    assert np.allclose(e4[-1], -1, atol=1e-8)


def test_rotosolve_state_functions_match_the_oracle_circuits():
    """rotosolve.py:15-62: the four state functions of the variational-environment problem, assembled from the
    SWAP-test operators exactly like rotosolve.py:203-211, reproduce the oracle's restatement of
    ground_state.py:170-228; `rotosolve(op_H(H), op_state, params)` runs as in the reference's driver."""
    from oracle import qmps_oracle as O
    from qmps_amd import rotosolve as RS
    from qmps_amd.ground_state import swap
    rng = np.random.default_rng(12)
    p = rng.standard_normal(30)
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    f, (energy, u_pur, v_pur, uv_pur) = O.opt_environment_objective(p, h)
    e_state, v_state, u_state, uv_state = (RS.op_state(p, w) for w in ('energy', 'v_purity', 'u_purity', 'uv_purity'))
    assert abs(np.real(e_state.conj() @ RS.op_H(h) @ e_state) - energy) < 1e-12
    assert abs(np.real(v_state.conj() @ np.kron(np.eye(2), np.kron(swap(), np.eye(2))) @ v_state) - v_pur) < 1e-12
    assert abs(np.real(u_state.conj() @ np.kron(np.eye(4), np.kron(swap(), np.eye(4))) @ u_state) - u_pur) < 1e-12
    assert abs(np.real(uv_state.conj() @ np.kron(np.kron(np.eye(2), swap()), np.eye(4)) @ uv_state) - uv_pur) < 1e-12
    es, hist = RS.rotosolve(RS.op_H(h), RS.op_state, p.copy(), N_iters=1)
    assert len(es) == 1 and es[0] <= energy + 1e-9
    a, b = RS.evo_Hs()
    assert a.shape == (64, 64) and b.shape == (16, 16) and RS.swapper().shape == (64, 64)


def test_full_tomography_env_objective_and_get_env(golden):
    """tests/test_represent.py:50-58 of the reference: the exact environment zeroes the Bloch-vector objective
    (< 1e-6 there); a random environment does not; the Nelder-Mead `get_env` finds one that does."""
    from oracle import qmps_oracle as O
    from qmps_amd import represent as R
    rng = np.random.default_rng(31)
    for U in O.haar_unitaries(rng, 4, 3):
        A = T.unitary_to_tensor(U)
        _, r = O.env_dense_eig(A)
        V = T.environment_to_unitary(np.linalg.cholesky(r))
        assert R.full_tomography_env_objective_function(R.FullStateTensor(U), R.FullEnvironment(V)) < 1e-12
        bad = T.environment_to_unitary(rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)))
        assert R.full_tomography_env_objective_function(R.FullStateTensor(U), R.FullEnvironment(bad)) > 1e-3
    V2 = R.get_env(U, C0=rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)))
    assert np.allclose(V2.conj().T @ V2, np.eye(4))
    assert R.full_tomography_env_objective_function(R.FullStateTensor(U), R.FullEnvironment(V2)) < 1e-4
    # Bloch vector helper: |0> and |+>
    assert np.allclose(R.bloch_vector_of(np.array([1, 0, 0, 0]), 0), [0, 0, 1])
    plus0 = np.kron(np.array([1, 1]) / np.sqrt(2), np.array([1, 0]))
    assert np.allclose(R.bloch_vector_of(plus0, 0), [1, 0, 0]) and np.allclose(R.bloch_vector_of(plus0, 1), [0, 0, 1])


def test_guess_initial_full_parameter_optimizer():
    """tools.py:287-305: 1 - |<Bell|(u x conj(U4(p)))|Bell>|^2 == 1 - |tr(U4(p)^+ u)|^2 / 16, checked against the
    explicit 4-qubit circuit; zero at the generating parameters."""
    from qmps_amd.represent import CNOT, H, MatrixGate, final_state, line_qubits
    rng = np.random.default_rng(2)
    p0, p1 = rng.standard_normal(15), rng.standard_normal(15)
    u = G.U4(p0)
    opt = T.GuessInitialFullParameterOptimizer(R.FullStateTensor(u), initial_guess=p1)
    assert opt.objective_function(p0) < 1e-14
    q = line_qubits(4)
    ops = [H(q[0]), H(q[1]), CNOT(q[0], q[2]), CNOT(q[1], q[3]), MatrixGate(u)(q[0], q[1]),
           MatrixGate(G.U4(p1).conj())(q[2], q[3]), CNOT(q[0], q[2]), CNOT(q[1], q[3]), H(q[0]), H(q[1])]
    amp = final_state(ops, 4)[0]
    assert abs(opt.objective_function(p1) - (1 - abs(amp) ** 2)) < 1e-12


def test_batched_bfgs_lockstep_driver():
    """tools.batched_bfgs on objectives with known minimisers: T independent problems in lock-step, central-difference
    gradients and the backtracking ladder as batches; converged rows stay put while the others continue."""
    from qmps_amd.tools import batched_bfgs, batched_fd_gradient
    rng = np.random.default_rng(3)
    T, P = 5, 4
    M = rng.standard_normal((T, P, P))
    Q = np.einsum('tij,tkj->tik', M, M) + 0.5 * np.eye(P)
    c = rng.standard_normal((T, P))
    calls = {'grad': 0, 'line': 0}

    def make(G, key):
        def f(C):
            calls[key] += 1
            assert C.shape == (T * G, P)
            t = np.arange(T * G) // G
            d = C - c[t]
            return 0.5 * np.einsum('bi,bij,bj->b', d, Q[t], d) + 0.1 * np.sum(d ** 4, axis=1)
        return f
    f0, g0 = batched_fd_gradient(make(2 * P + 1, 'grad'), np.zeros((T, P)))
    assert np.abs(g0 - (np.einsum('tij,tj->ti', Q, -c) - 0.4 * c ** 3)).max() < 1e-6
    res = batched_bfgs(make(2 * P + 1, 'grad'), make(8, 'line'), np.zeros((T, P)), maxiter=100, gtol=1e-7)
    assert res['converged'].all()
    assert np.abs(res['x'] - c).max() < 1e-6
    assert res['history'].shape == (res['nit'] + 1, T)
    assert np.all(np.diff(res['history'], axis=0) <= 1e-15)          # monotone per trajectory
    assert calls['line'] == res['nit'] and calls['grad'] == res['nit'] + 2


def test_batched_bfgs_with_supplied_gradient_and_two_stage_ladder():
    """value_and_grad replaces the central-difference batches; first_rungs evaluates the ladder in two stages (the second one
    only when some trajectory needs it) - same minimisers."""
    from qmps_amd.tools import batched_bfgs
    rng = np.random.default_rng(4)
    T, P = 6, 3
    c = rng.standard_normal((T, P))
    sizes = []

    def rosen_like(C, t):
        d = C - c[t]
        return np.sum(d ** 2, axis=1) + 5.0 * (d[:, 0] * d[:, 1]) ** 2 + 0.3 * np.sum(d ** 4, axis=1)

    def line(C):
        G = len(C) // T
        sizes.append(G)
        return rosen_like(C, np.arange(len(C)) // G)

    def vg(X):
        f = rosen_like(X, np.arange(T))
        h = 1e-6
        g = np.empty_like(X)
        for k in range(P):
            e = np.zeros(P)
            e[k] = h
            g[:, k] = (rosen_like(X + e, np.arange(T)) - rosen_like(X - e, np.arange(T))) / (2 * h)
        return f, g
    X0 = c + rng.standard_normal((T, P))
    res = batched_bfgs(None, line, X0, maxiter=200, gtol=1e-8, value_and_grad=vg, first_rungs=2)
    assert res['converged'].all() and np.abs(res['x'] - c).max() < 1e-6
    assert set(sizes) <= {2, 6} and sizes.count(2) == res['nit'] and sizes.count(6) < res['nit']
    res_full = batched_bfgs(None, line, X0, maxiter=200, gtol=1e-8, value_and_grad=vg)
    assert np.abs(res_full['x'] - res['x']).max() < 1e-6


def test_batched_nelder_mead_reproduces_scipy():
    """tools.batched_nelder_mead takes scipy's decisions on the same values: same simplex sequence, same minimiser, same
    evaluation count as scipy.optimize.minimize(method='Nelder-Mead') - with the points of an iteration evaluated as one batch."""
    from scipy.optimize import minimize
    from qmps_amd.tools import batched_nelder_mead
    rng = np.random.default_rng(9)
    for N in (2, 4, 7):
        c = rng.standard_normal(N)
        Q = rng.standard_normal((N, N))
        Q = Q @ Q.T + np.eye(N)

        def f(x):
            d = np.asarray(x) - c
            return float(d @ Q @ d + 0.2 * np.sum(np.cos(3 * d)))
        x0 = rng.standard_normal(N)
        ref = minimize(f, x0, method='Nelder-Mead', tol=1e-9, options={'maxiter': 4000})
        for spec in (True, False):
            res = batched_nelder_mead(lambda X: np.array([f(x) for x in X]), x0, xatol=1e-9, fatol=1e-9, maxiter=4000, speculate=spec)
            if True:
                assert res.nit == ref.nit and res.nfev == ref.nfev, (N, spec, res.nit, ref.nit, res.nfev, ref.nfev)
                assert np.array_equal(res.x, ref.x) and res.fun == ref.fun
            assert res.n_batches <= res.nit + 1 + (res.nit if not spec else res.nit)
        assert res.nfev_batched >= res.nfev


def test_batched_bfgs_speculative_full_step():
    """speculative=True evaluates objective and gradient at the full step first and falls back to the rest of the ladder only
    when some trajectory rejects it: the same iterates as the plain ladder, fewer batches."""
    from qmps_amd.tools import batched_bfgs
    rng = np.random.default_rng(6)
    T, P = 5, 3
    c = rng.standard_normal((T, P))
    n = {'vg': 0, 'line': 0}

    def val(C, t):
        d = C - c[t]
        return np.sum(d ** 2, axis=1) + 3.0 * (d[:, 0] * d[:, 1]) ** 2 + 0.5 * np.sum(d ** 4, axis=1)

    def line(C):
        n['line'] += 1
        G = len(C) // T
        return val(C, np.arange(len(C)) // G)

    def vg(X):
        n['vg'] += 1
        h = 1e-6
        g = np.stack([(val(X + h * np.eye(P)[k], np.arange(T)) - val(X - h * np.eye(P)[k], np.arange(T))) / (2 * h) for k in range(P)], axis=1)
        return val(X, np.arange(T)), g
    X0 = c + 1.5 * rng.standard_normal((T, P))
    plain = batched_bfgs(None, line, X0, maxiter=100, gtol=1e-8, value_and_grad=vg)
    n_plain = dict(n)
    n.update(vg=0, line=0)
    spec = batched_bfgs(None, line, X0, maxiter=100, gtol=1e-8, value_and_grad=vg, speculative=True)
    assert spec['nit'] == plain['nit'] and np.abs(spec['history'] - plain['history']).max() < 1e-12
    assert np.abs(spec['x'] - plain['x']).max() < 1e-10 and spec['converged'].all()
    assert n['line'] < n_plain['line'] and n['vg'] + n['line'] < n_plain['vg'] + n_plain['line']


def test_state_gate_adapter_and_native_flag():
    """StateGate's constructor takes the parameters alone - the host paths build gates through `build_gate`; the
    lock-step evolver's C driver is chosen only for the option combinations it implements."""
    from qmps_amd import represent as R
    from qmps_amd import new_time_evolve as NT
    p = np.linspace(0.1, 0.6, 6)
    g = R.build_gate(R.StateGate, 2, p)
    assert isinstance(g, R.StateGate) and np.allclose(R.unitary(g) @ R.unitary(g).conj().T, np.eye(4))
    with pytest.raises(ValueError):
        R.build_gate(R.StateGate, 4, p)
    assert isinstance(R.build_gate(R.ShallowCNOTStateTensor, 4, np.zeros(4)), R.ShallowCNOTStateTensor)
    A = NT.state_tensor(p, D=2, state_tensor=R.StateGate)          # (host path: no device needed)
    assert A.shape == (2, 2, 2) and np.allclose(sum(a.conj().T @ a for a in A), np.eye(2))
    assert NT.state_tensor_of(R.StateGate, 2, p).shape == (2, 2, 2)


def test_optimize_restarts_is_the_restart_loop_in_lock_step():
    """tools.optimize_restarts / Optimizer.optimize_restarts (round 6): the reference's restart loops (scripts/ground_state_finding.py:137-154,
    scripts/noisy_optimization.py:46-72) as one lock-step over batches.  On a toy objective with a batched form (no device needed): every restart
    reaches the minimum scipy's BFGS reaches from the same start (a convex objective: one basin), no scalar objective call is made, the best
    restart becomes `optimized_result`, `update_state` runs once; a method without a lock-step form falls back to `optimize()` per restart and
    leaves the settings and the initial guess as they were."""
    from scipy.optimize import minimize
    from qmps_amd.tools import Optimizer

    calls = {'batch': 0, 'scalar': 0, 'update': 0}
    a = np.array([0.3, -1.2, 0.8])
    M = np.array([[2.0, 0.3, 0.0], [0.3, 1.0, -0.2], [0.0, -0.2, 1.5]])

    def f(P):
        Q = np.atleast_2d(np.asarray(P, dtype=float)) - a
        return np.einsum('bi,ij,bj->b', Q, M, Q) + 0.1 * np.sum(Q ** 4, axis=1) - 0.7

    class Toy(Optimizer):
        def objective_function(self, p):
            calls['scalar'] += 1
            return float(f(p)[0])

        def batch_objective_function(self, P):
            calls['batch'] += 1
            return f(P)

        def update_state(self):
            calls['update'] += 1

    rng = np.random.default_rng(8)
    X0 = 2.0 * rng.standard_normal((12, 3))
    opt = Toy(initial_guess=X0[0].copy())
    opt.change_settings({'verbose': False, 'store_values': False, 'tol': 1e-7})
    res = opt.optimize_restarts(X0, method='BFGS')
    assert len(res) == 12 and calls['scalar'] == 0 and calls['update'] == 1 and calls['batch'] < 200
    for r, x0 in zip(res, X0):
        ref = minimize(lambda p: float(f(p)[0]), x0, method='BFGS', tol=1e-9)
        assert abs(r.fun - ref.fun) < 1e-9 and np.abs(r.x - ref.x).max() < 1e-5 and r.success, (r.fun, ref.fun)
    assert opt.optimized_result.fun == min(r.fun for r in res) and opt.restart_results is res and abs(opt.optimized_result.fun + 0.7) < 1e-9
    # a method without a lock-step form: optimize() per restart; settings and initial guess untouched afterwards
    before = (dict(opt.settings), opt.initial_guess.copy())
    res2 = opt.optimize_restarts(X0[:3], method='Nelder-Mead', maxiter=600, tol=1e-8)
    assert len(res2) == 3 and opt.settings == before[0] and np.array_equal(opt.initial_guess, before[1])
    assert all(abs(r_.fun + 0.7) < 1e-6 for r_ in res2)


def test_bond_dimension_embedding_keeps_the_state():
    """`insu2N`, `extractv`, `embed_bond_dimension` (round 6): the D -> 2D hand-over of the reference's bond-dimension driver
    (scripts/bond_dimension.py:21-35, 50; xmps.spin's versions are not in the reference tree - conventions are this module's `SU`).
    Pinned by what the reference says they must do: insu2N(v) generates U x 1; extractv inverts SU up to a global phase (also far from the
    identity, where the principal logarithm wraps); at eps = 0 the embedded unitary gives the tensor A x 1 - the same iMPS - and the
    oracle's energy per site is unchanged; the reference's eps = 4e-2 moves it by a few per cent only."""
    from qmps_amd.ground_state import SU, Hamiltonian, embed_bond_dimension, extractv, insu2N
    rng = np.random.default_rng(21)
    H = Hamiltonian({'XX': 1, 'YY': 1}).to_matrix()
    for D in (2, 4):
        for scale in (0.3, 3.0):
            v = scale * rng.standard_normal((2 * D) ** 2 - 1)
            U = SU(v, 2 * D)
            assert np.abs(SU(insu2N(v), 4 * D) - np.kron(U, np.eye(2))).max() < 1e-12
            W = SU(extractv(U), 2 * D)
            ph = np.vdot(W, U) / abs(np.vdot(W, U))
            assert np.abs(ph * W - U).max() < 1e-10
            A = O.unitary_to_tensor(U[None])[0]
            A2 = O.unitary_to_tensor(SU(embed_bond_dimension(v, eps=0.0), 4 * D)[None])[0]
            ref = np.einsum('sij,ab->siajb', A, np.eye(2)).reshape(2, 2 * D, 2 * D)
            ph = np.vdot(ref, A2) / abs(np.vdot(ref, A2))
            assert np.abs(A2 - ph * ref).max() < 1e-10                      # A x 1 up to a global phase
            e1, e2 = O.energy_closed_form(A, H), O.energy_closed_form(A2, H)
            assert abs(e1 - e2) < 1e-10
            e3 = O.energy_closed_form(O.unitary_to_tensor(SU(embed_bond_dimension(v), 4 * D)[None])[0], H)
            assert abs(e3 - e1) < 0.25
    with pytest.raises(ValueError):
        insu2N(np.zeros(5))
