"""CPU: pin the oracle against the reference's golden vectors and known answers."""
import numpy as np
import pytest

from oracle import qmps_oracle as O


# ---- reference-generated vectors (tests/golden/make_golden.py, ref_* arrays) -------------------
def test_hamiltonian_kat(golden):
    """Exact 4x4 TFIM matrix of tests/test_ground_state.py:26-38 and the reference's own output."""
    J, g = -1, 1
    kat = np.array([[J, g / 2, g / 2, 0], [g / 2, -J, 0, g / 2], [g / 2, 0, -J, g / 2], [0, g / 2, g / 2, J]])
    assert np.allclose(O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), kat)
    assert np.allclose(O.hamiltonian_matrix({'ZZ': -1, 'IX': 0.5, 'XI': 0.5}), kat)
    assert np.array_equal(golden['ref_h_tfim'], O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
    assert np.array_equal(golden['ref_h_tfim_split'], golden['ref_h_tfim'])
    assert np.allclose(golden['ref_h_xxz'], O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}))
    assert np.allclose(golden['ref_h_xy'], O.hamiltonian_matrix({'XX': 1, 'YY': 1}))
    assert np.allclose(golden['ref_h_tfim_g07'], O.hamiltonian_matrix({'ZZ': -1, 'X': 0.7}))


@pytest.mark.parametrize('D', [2, 4, 8, 16])
def test_unitary_to_tensor_matches_reference(D, golden, c_oracle):
    U, A = golden[f'U_D{D}'], golden[f'ref_A_D{D}']
    assert np.array_equal(O.unitary_to_tensor(U), A)
    assert np.array_equal(c_oracle.unitary_to_tensor(U), A)
    for a in A:  # left isometry
        assert np.allclose(sum(x.conj().T @ x for x in a), np.eye(D))


def test_tensor_to_unitary_reference_checks(golden):
    """tools.py:131-137 five checks passed in the reference run; the round trip reproduces A."""
    assert golden['ref_t2u_passed'].all()
    assert np.allclose(golden['ref_t2u_roundtrip'], golden['ref_A_D2'])
    for a, u_ref in zip(golden['ref_A_D2'], golden['ref_t2u_U']):
        u = O.tensor_to_unitary(a)
        assert np.allclose(u.conj().T @ u, np.eye(4))
        assert np.allclose(u[:, :2], u_ref[:, :2])          # the isometry block is fixed,
        assert np.allclose(O.unitary_to_tensor(u), a)      # the completion columns are arbitrary


@pytest.mark.parametrize('D', [2, 4, 8])
def test_environment_to_unitary_matches_reference(D, golden):
    for L, V_ref in zip(golden[f'oracle_L_D{D}'], golden[f'ref_V_D{D}']):
        V = O.environment_to_unitary(L)
        assert np.allclose(V.conj().T @ V, np.eye(D * D))
        assert np.allclose(V[:, 0], V_ref[:, 0])            # only column 0 is defined
        assert np.allclose(V_ref[:, 0], L.reshape(-1) / np.linalg.norm(L))


def test_merge_and_helpers_match_reference(golden):
    A2 = golden['ref_A_D2']
    for i in range(len(A2)):
        assert np.allclose(O.merge(A2[i], A2[(i + 1) % len(A2)]), golden['ref_merge_D2'][i])
    v = golden['realvec_in']
    assert np.allclose(golden['ref_from_real_vector'], v[:4] + 1j * v[4:])


# ---- double restatement: state-vector path == closed form ------------------------------------
@pytest.mark.parametrize('D', [2, 4, 8])
def test_statevector_equals_closed_form(D, golden):
    h = golden['ref_h_tfim']
    for U, A, r, V, E_sv, E_cf in zip(golden[f'U_D{D}'], golden[f'ref_A_D{D}'], golden[f'oracle_r_D{D}'],
                                      golden[f'ref_V_D{D}'], golden[f'oracle_E_statevec_D{D}'],
                                      golden[f'oracle_E_closed_D{D}']):
        assert abs(E_sv - E_cf) < 1e-13
        assert abs(O.energy_statevector(U, h, V) - E_sv) < 1e-13       # with the REFERENCE's V
        assert abs(O.energy_closed_form(A, h, r) - E_cf) < 1e-13
        assert abs(O.reference_structured_energy(U, h) - E_cf) < 1e-12  # full reference-structured path


def test_statevector_by_explicit_kronecker_products():
    """scripts/ground_state_finding.py:119-128: (U x 1 x 1)(1 x U x 1)(1 x 1 x V)|0000>, D = 2."""
    rng = np.random.default_rng(5)
    U = O.haar_unitaries(rng, 4, 1)[0]
    V = O.get_env_exact(U)
    I = np.eye(2)
    mb = lambda ops: __import__('functools').reduce(np.kron, ops)  # noqa: E731
    psi = mb([U, I, I]) @ mb([I, U, I]) @ mb([I, I, V]) @ mb([np.array([1, 0])] * 4)
    assert np.allclose(psi, O.state_vector(U, V, 2))
    Ha = -np.kron(O.SZ, O.SZ) + 0.5 * (np.kron(I, O.SX) + np.kron(O.SX, I))
    e = np.real(psi.conj() @ mb([I, Ha, I]) @ psi)
    assert abs(e - O.energy_closed_form(O.unitary_to_tensor(U), O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))) < 1e-13


def test_ansatz_by_explicit_kronecker_products():
    """ShallowCNOTStateTensor(2, [b, g]) built by hand: CNOT(q0,q1) . (H x 1) . (rx x rx) . (rz x rz)."""
    b, g = 0.37, -1.21
    rz = np.diag([np.exp(-0.5j * b), np.exp(0.5j * b)])
    c, s = np.cos(g / 2), np.sin(g / 2)
    rx = np.array([[c, -1j * s], [-1j * s, c]])
    Hd = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]])
    U = cnot @ np.kron(Hd, np.eye(2)) @ np.kron(rx, rx) @ np.kron(rz, rz)
    assert np.allclose(U, O.shallow_cnot_unitary(2, [b, g]))
    # D = 4: three qubits, ladder CNOT(q1,q2) then CNOT(q0,q1)
    I = np.eye(2)
    c12, c01 = np.kron(I, cnot), np.kron(cnot, I)
    U4 = c01 @ c12 @ np.kron(Hd, np.eye(4)) @ np.kron(rx, np.kron(rx, rx)) @ np.kron(rz, np.kron(rz, rz))
    assert np.allclose(U4, O.shallow_cnot_unitary(4, [b, g]))


# ---- physics known answers --------------------------------------------------------------------
def test_tfim_exact_energy_integral():
    assert abs(O.tfim_exact_energy(1.0) - (-4 / np.pi)) < 1e-9    # tests/test_ground_state.py:101-102


def test_variational_bounds(golden):
    """E(params) >= E0_exact(g) (tests/test_ground_state.py:218) for every state in the fixtures.
    D2_gse (scripts/noisy_optimization.py:93, TenPy iDMRG chi = 2) is the reference's D = 2 yardstick;
    it is not a bound (a D = 2 iMPS reaches -1.27254, see test_D2_optimum_beats_D2_gse)."""
    E0 = O.tfim_exact_energy(1.0)
    D2_gse = -1.269909412573
    for D in (2, 4, 8, 16):
        assert np.all(golden[f'oracle_E_closed_D{D}'] >= E0)
    assert np.all(golden['oracle_cnot_E_D2'] >= D2_gse - 1e-9)
    assert np.all(golden['oracle_cnot_E_D4'] >= E0)


def test_D2_optimum_beats_D2_gse():
    """A specific D = 2 state unitary whose energy density is -1.2725425 < D2_gse, above E0; checked by
    both restatements and by the bulk bond energy of an explicit 16-site chain built from A."""
    from scipy.optimize import minimize
    from qmps_amd.ground_state import SU      # parameterisation only (host-side numpy)
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    rng = np.random.default_rng(2024)
    rng.standard_normal(15)
    res = minimize(lambda p: O.energy_closed_form(O.unitary_to_tensor(SU(p, 4)), h), rng.standard_normal(15),
                   method='Nelder-Mead', options={'maxiter': 3000, 'xatol': 1e-9, 'fatol': 1e-12})
    U = SU(res.x, 4)
    A = O.unitary_to_tensor(U)
    E = O.energy_closed_form(A, h)
    assert O.tfim_exact_energy(1.0) < E < -1.2720 and abs(E - O.energy_statevector(U, h)) < 1e-12
    N = 16
    T = np.eye(2, dtype=complex)               # open left bond: the left environment of an isometry is 1
    for _ in range(N):
        T = np.einsum('pi,sij->psj', T, A).reshape(-1, 2)
    _, r = O.env_dense_eig(A)
    w, v = np.linalg.eigh(r)                   # purify the right boundary with sqrt(r): exact bulk RDM
    psi = (T @ ((v * np.sqrt(w)) @ v.conj().T)).reshape((2,) * (N + 2))
    k = 1 + N // 2
    axes = [i for i in range(N + 2) if i not in (k, k + 1)]
    rho = np.tensordot(psi, psi.conj(), axes=(axes, axes)).reshape(4, 4)
    assert abs(np.trace(rho) - 1) < 1e-12 and abs(np.real(np.trace(h @ rho)) - E) < 1e-12


def test_reference_fixture_A(golden):
    """fixtures/A.npy: header [d, D, n] then 8 complex numbers.  After left-canonicalisation
    (QR gauge) the tensor runs through the oracle; the energy lies above the exact E0."""
    flat = golden['ref_fixture_A_flat']
    d, D, n = (int(x.real) for x in flat[:3])
    assert (d, D, n) == (2, 2, 1)
    A = flat[3:].reshape(d, D, D)
    # left-canonical gauge: A_s -> l^{1/2} A_s l^{-1/2} with l the left fixed point
    w, v = np.linalg.eig(O.transfer_matrix(A).T)
    l = v[:, np.argmax(abs(w))].reshape(D, D)
    l = l / np.trace(l)
    l = (l + l.conj().T) / 2
    ev, evec = np.linalg.eigh(l.T)
    sq = (evec * np.sqrt(ev)) @ evec.conj().T
    AL = np.stack([sq @ a @ np.linalg.inv(sq) for a in A]) / np.sqrt(abs(w).max())
    assert np.allclose(sum(a.conj().T @ a for a in AL), np.eye(D), atol=1e-10)
    E = O.energy_closed_form(AL, O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
    assert O.tfim_exact_energy(1.0) <= E <= 2.0


@pytest.mark.parametrize('D', [2, 4, 8])
def test_environment_fixed_point_properties(D, golden):
    """tests/test_represent.py:23-31: r = C C^+ is a right eigenvector with eta = 1, identity a left one."""
    for A, r in zip(golden[f'ref_A_D{D}'], golden[f'oracle_r_D{D}']):
        assert np.allclose(O.apply_transfer(A, r), r, atol=1e-12)
        assert np.allclose(np.einsum('sij,ik,skl->jl', A.conj(), np.eye(D), A), np.eye(D))
        assert np.linalg.eigvalsh(r).min() > 0


def test_bloch_vector_is_one_site_expectation():
    """tests/test_represent.py:33-48: Bloch vector of qubit 1 of State(U,V) (n = 1) equals the
    one-site expectation values tr(A_t r A_s^+) sigma[s,t]."""
    rng = np.random.default_rng(11)
    U = O.haar_unitaries(rng, 4, 1)[0]
    A = O.unitary_to_tensor(U)
    _, r = O.env_dense_eig(A)
    V = O.environment_to_unitary(O.env_cholesky(r))
    psi = O.state_vector(U, V, 1).reshape(2, 2, 2)        # [a, sigma, b]
    rho1 = np.einsum('asb,atb->st', psi, psi.conj())
    for P in (O.SX, O.SY, O.SZ):
        lhs = np.real(np.trace(rho1 @ P))
        rhs = np.real(sum(P[s, t] * np.trace(A[t] @ r @ A[s].conj().T) for s in range(2) for t in range(2)))
        assert abs(lhs - rhs) < 1e-13


# ---- power iteration: numpy twin == C twin; agrees with dense eig ------------------------------
@pytest.mark.parametrize('D', [2, 4, 8, 16])
def test_c_oracle_equals_numpy_oracle(D, golden, c_oracle):
    A, h = golden[f'ref_A_D{D}'], golden['ref_h_tfim']
    out = c_oracle.energy_batch(A, h, want_r=True, want_rho=True)
    assert np.array_equal(out['iters'], golden[f'oracle_iters_D{D}'])
    assert np.abs(out['E'][:, 0] - golden[f'oracle_E_power_D{D}']).max() < 1e-13
    assert np.abs(out['E'][:, 0] - golden[f'oracle_E_closed_D{D}']).max() < 1e-11
    assert np.abs(out['r'] - golden[f'oracle_r_D{D}']).max() < 1e-11
    for a, r, rho in zip(A, out['r'], out['rho']):
        assert np.allclose(rho, O.two_site_rdm(a, r), atol=1e-13)
        assert abs(np.trace(rho) - 1) < 1e-12


@pytest.mark.parametrize('handoff,skip,period', [(0, 6, 4), (0, 3, 0), (0, 0, 2), (16, 0, 4), (64, 0, 0)])
def test_squaring_schedules_reach_the_dense_eig_environment(handoff, skip, period, golden, c_oracle):
    """The repeated-squaring restatements (pure squaring; squarings then products with T^(2^m), squaring every
    `period`-th) are the power method taken 2^m steps at a time: numpy twin == C twin step for step, same fixed point
    as the dense eigen-solve, never fewer equivalent steps than the plain iteration needs."""
    A, h = golden['ref_A_D4'], golden['ref_h_tfim']
    out = c_oracle.energy_batch(A, h, want_r=True, handoff=handoff, skip=skip, period=period)
    plain = golden['oracle_iters_D4']
    assert np.all(out['status'] == 0)
    assert np.abs(out['r'] - golden['oracle_r_D4']).max() < 1e-11
    assert np.abs(out['E'][:, 0] - golden['oracle_E_closed_D4']).max() < 1e-11
    assert np.all(out['iters'] >= np.minimum(plain, handoff + 1) - 1)
    for a, r, it in zip(A[:8], out['r'], out['iters']):
        r2, it2, st2 = O.env_power_iteration(a, handoff=handoff, skip=skip, period=period)
        assert it2 == it and st2 == 0 and np.abs(r2 - r).max() < 1e-13
        _, r_eig = O.env_dense_eig(a)
        assert np.abs(r - r_eig).max() < 1e-11


def test_c_oracle_status_codes(c_oracle):
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    rng = np.random.default_rng(3)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 5))
    out = c_oracle.energy_batch(A, h, max_iter=3)
    assert np.all(out['status'] == 1) and np.all(out['iters'] == 3)
    prod = O.unitary_to_tensor(np.eye(8, dtype=complex)[None])
    out = c_oracle.energy_batch(prod, h)
    assert out['status'][0] == 2 and abs(out['E'][0, 0] + 1.0) < 1e-12


def test_two_site_cell_oracle(golden):
    h = golden['ref_h_tfim']
    for U1, U2, E in zip(golden['cell_U1'], golden['cell_U2'], golden['oracle_cell_E']):
        A1, A2 = O.unitary_to_tensor(U1), O.unitary_to_tensor(U2)
        assert abs(O.two_site_cell_energy_closed(A1, A2, h) - E) < 1e-12
    # a uniform cell (U1 == U2) reduces to the single-site energy
    U = golden['U_D2'][0]
    assert abs(O.two_site_cell_energy(U, U, h) - golden['oracle_E_closed_D2'][0]) < 1e-12


def test_rotosolve_updates():
    th = np.linspace(-3, 3, 7)
    for a, b, c in [(0.3, 1.1, -0.2), (-0.7, 0.4, 2.0)]:
        f = lambda x: a * np.sin(x + b) + c  # noqa: E731
        x0 = 0.4
        d = O.rotosolve_update(f(x0), f(x0 + np.pi / 2), f(x0 - np.pi / 2))
        assert f(x0 + d) <= min(f(x0 + t) for t in th) + 1e-12
    g = lambda x: 0.8 * np.sin(2 * x + 0.3) + 0.5 * np.sin(x - 1.0)  # noqa: E731
    d = O.double_rotosolve_update(*[g(0.2 + s) for s in O.ROTO_SHIFTS])
    xs = np.linspace(-np.pi, np.pi, 20001)
    assert g(0.2 + d) <= g(xs).min() + 1e-6


# ---- f-3: time-evolution overlap -------------------------------------------------------------
def test_overlap_circuit_identity():
    """The reference's asserts (qmps/new_time_evolve.py:174-184) and SURVEY App. B-3: the 6-qubit circuit
    amplitude is psi[0] = eta/2 (up to the eigenvector's phase), for W = 1 and for W = exp(-i h dt);
    for W = 1 the two-site eta is the square of the one-site one; eta = 1 for B = A."""
    from scipy.linalg import expm
    rng = np.random.default_rng(17)
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    for WW in (np.eye(4, dtype=complex), expm(-0.1j * h)):
        for _ in range(3):
            A = O.unitary_to_tensor(O.haar_unitaries(rng, 4, 1)[0])
            B = O.unitary_to_tensor(O.haar_unitaries(rng, 4, 1)[0])
            eta, r = O.overlap_eta(A, B, WW)
            amp = O.overlap_circuit_amplitude(A, B, WW, r)
            assert abs(2 * abs(amp) - abs(eta)) < 1e-12
            assert abs(O.overlap_objective(A, B, WW) + np.sqrt(2 * abs(amp))) < 1e-12
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 4, 1)[0])
    B = O.unitary_to_tensor(O.haar_unitaries(rng, 4, 1)[0])
    w1 = np.linalg.eigvals(O.transfer_matrix(A, B))
    eta2, _ = O.overlap_eta(A, B, np.eye(4))
    assert abs(abs(eta2) - np.abs(w1).max() ** 2) < 1e-12
    assert abs(O.overlap_eta(A, A, np.eye(4))[0] - 1) < 1e-12


def test_remaining_ansatz_builders_are_unitary_and_match_their_gate_lists():
    """Oracle builders of ShallowCNOTStateTensor_nonuniform / ExactAfter4 / StateGate: unitary, and equal to a product of the
    elementary gates written out by hand for one layer (represent.py:312-332, 356-380, 406-423)."""
    rng = np.random.default_rng(41)
    for D in (2, 4, 8):
        n = int(np.log2(D)) + 1
        for U in (O.shallow_cnot_nonuniform_unitary(D, rng.standard_normal(4 * n)), O.exact_after4_unitary(D, rng.standard_normal(12))):
            assert np.abs(U.conj().T @ U - np.eye(2 * D)).max() < 1e-13
    # D = 2, one layer of the nonuniform ansatz by hand: CNOT(q0, q1) (rx(p3) x rx(p2) after rz ...) in big-endian kron order
    p = rng.standard_normal(4)
    hand = O.CNOT @ np.kron(O.rx(p[2]), O.rx(p[3])) @ np.kron(O.rz(p[0]), O.rz(p[1]))
    assert np.abs(O.shallow_cnot_nonuniform_unitary(2, p) - hand).max() < 1e-14
    # ExactAfter4 at D = 2: SWAP(q0, q1) twice (i = 0 and the cyclic i = 1) cancels
    q = rng.standard_normal(6)
    hand = O.CNOT @ np.kron(O.rz(q[2]), O.rz(q[5])) @ np.kron(O.rx(q[1]), O.rx(q[4])) @ np.kron(O.rz(q[0]), O.rz(q[3]))
    assert np.abs(O.exact_after4_unitary(2, q) - hand).max() < 1e-14
    # StateGate: XX**e and YY**f commute with each other and have eigenvalues 1 / e^{i pi t}
    s = rng.standard_normal(6)
    U = O.state_gate_unitary(s)
    assert np.abs(U.conj().T @ U - np.eye(4)).max() < 1e-13
    XX = O._pauli_pair_power(np.array([[0, 1], [1, 0]], dtype=complex), 0.37)
    w = np.linalg.eigvals(XX)
    assert np.allclose(sorted(np.angle(w)), sorted([0, 0, 0.37 * np.pi, 0.37 * np.pi]))


def test_overlap_arpack_route_equals_the_dense_eigen_solve():
    """The reference's route (xmps Map -> scipy.sparse.linalg.eigs, ARPACK in operator form) and the dense eigen-solve of the
    D^2 x D^2 matrix give the same dominant eigenvalue and ray for time-step candidates at every bond dimension."""
    from scipy.linalg import expm
    rng = np.random.default_rng(23)
    WW = expm(-0.05j * O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
    for D, P in ((2, 8), (4, 4), (8, 6), (16, 8)):
        for _ in range(3):
            p = rng.standard_normal(P)
            A = O.unitary_to_tensor(O.shallow_cnot_unitary(D, p))
            B = O.unitary_to_tensor(O.shallow_cnot_unitary(D, p + 0.05 * rng.standard_normal(P)))
            e1, r1 = O.overlap_eta(A, B, WW)
            e2, r2 = O.overlap_eta_arpack(A, B, WW)
            assert abs(e1 - e2) < 1e-12
            assert abs(abs(np.vdot(r1, r2)) - 1) < 1e-10


def test_bench_schedule_replay_matches_the_oracle_schedule(golden, c_oracle):
    """bench.py turns the per-item step counts it reads back into executed squarings / mat-vecs by replaying the
    kernel's schedule; the replay must land exactly on the step counts the oracle's restatement of that schedule
    produces, for every cap."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bench', os.path.join(os.path.dirname(os.path.dirname(__file__)), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    A, h = golden['ref_A_D4'], golden['ref_h_tfim']
    for cap in (10000, 300, 130, 64, 5):
        out = c_oracle.energy_batch(A, h, max_iter=cap, handoff=0, skip=6, period=4)
        for k in np.unique(out['iters']):
            nsq, nmv = bench.squaring_schedule_ops(int(k), 6, 4, cap, True)
            # re-walk the schedule with those counts
            m = 0
            while m < 6 and (2 << m) <= cap:
                m += 1
            it, left, count, sq = (1 << m) if m > 0 else 0, nmv, 0, m
            while left:
                it += 1 << m
                left -= 1
                count += 1
                if left and count == 4:
                    m += 1
                    sq += 1
                    count = 0
            assert it == k and sq == nsq, (cap, k, nsq, nmv)


def test_d2_optimum_of_the_tfim():
    """What is the best D = 2 energy of the TFIM at g = 1?  The reference's figure script draws a "D = 2" line at
    D2_gse = -1.269909412573 (scripts/noisy_optimization.py:93) and the round-4 verdict asked for the bench's D = 2 leg to be read against
    it (D2_gse is a TenPy iDMRG number at chi = 2; test_D2_optimum_beats_D2_gse above exhibits one state below it).  The OPTIMUM of the
    D = 2 manifold is -1.2725424859 (7.0e-4 above the exact -4/pi), found here two independent ways with
    the oracle's closed-form energy - scipy BFGS over the 15 angles of the universal two-qubit gate (ShallowFullStateTensor,
    represent.py:382-404) and over a free complex 4 x 2 isometry (QR of 16 real numbers) - and on the device by
    examples/ground_state_tfim.py (tests/test_examples_gpu.py).  The reference-executed fixtures pin the energy of this gate family to
    the reference's own objective (tests/test_refshim_cpu.py), so this is the number the reference's code would report too."""
    from scipy.optimize import minimize
    h = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
    exact = -4.0 / np.pi

    def e_gate(p):
        return O.energy_closed_form(O.unitary_to_tensor(O.shallow_full_unitary(p)), h)

    def tensor_of(x):
        Q, _ = np.linalg.qr((x[:8] + 1j * x[8:]).reshape(4, 2))
        return np.ascontiguousarray(Q.reshape(2, 2, 2).transpose(1, 0, 2))            # Q[(i, s), j] -> A[s, i, j]: left-isometric

    def e_free(x):
        return O.energy_closed_form(tensor_of(x), h)

    r1 = minimize(e_gate, np.random.default_rng(0).standard_normal(15), method='BFGS', options={'maxiter': 400})
    r2 = minimize(e_free, np.random.default_rng(5).standard_normal(16), method='BFGS', options={'maxiter': 600})
    assert abs(r1.fun - (-1.2725424859)) < 1e-6 and abs(r2.fun - (-1.2725424859)) < 1e-6
    assert exact < min(r1.fun, r2.fun) and max(r1.fun, r2.fun) < -1.269909412573 - 2e-3
    # the two optima are the same physical state: fidelity per site 1
    A1, A2 = O.unitary_to_tensor(O.shallow_full_unitary(r1.x)), tensor_of(r2.x)
    ov = abs(np.linalg.eigvals(O.transfer_matrix(A1, A2))).max()
    assert abs(ov - 1.0) < 1e-4


def test_oracle_environment_on_a_defective_transfer_matrix():
    """Round 5 (randomised stress of the energy path): ExactAfter4 at D = 2 with angles on the pi/4 grid has the transfer spectrum (1, 0, 0, 0)
    with a nilpotent block; numpy's eig returned a `dominant eigenvector` with residual 0.125, the oracle's energy was 0.625 where the device - and
    the power method, and the definition - give 0.5.  The oracle now checks its eigenvector against the fixed-point equation and polishes it."""
    import evolve_replay as ER
    p = np.array([-0.00954598307922015, -np.pi / 2, np.pi, np.pi / 2, np.pi, np.pi / 2, np.pi, 0.0, -np.pi / 4, -np.pi / 4, np.pi, -np.pi / 2])
    A = ER.tensor(5, 2, p)
    w = np.sort(np.abs(np.linalg.eigvals(O.transfer_matrix(A))))[::-1]
    assert abs(w[0] - 1.0) < 1e-12 and w[1] < 1e-6
    eta, r = O.env_dense_eig(A)
    assert np.abs(O.apply_transfer(A, r) - r).max() < 1e-13 and abs(np.trace(r) - 1.0) < 1e-13
    assert abs(O.energy_closed_form(A, O.hamiltonian_matrix({'XX': 1.0, 'YY': 1.0, 'ZZ': 0.5})) - 0.5) < 1e-12
    assert abs(O.energy_closed_form(A, O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})) - 0.35355339059327373) < 1e-12

