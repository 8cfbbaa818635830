"""CPU tests: the oracle and the host-side API mirror against tests/golden/refshim_golden.npz - numbers produced by the
REFERENCE'S OWN Python (gate classes, State, the optimisers' objective functions, scripts/loschmidt.py:obj, the environment
embeddings, the rotosolve drivers) executed in the build container over the documented-convention cirq / xmps stand-ins of
tests/golden/cirq_shim.py (generator: tests/golden/make_refshim_golden.py).  Nothing here reads /root/reference.

What these fixtures pin that the `oracle_*` ones could not: gate order and qubit placement of every ansatz class, the register
layout of State(U, V, 2) and where h sits in it, the 6-qubit overlap circuit and its normalisation (psi[0] = eta / 2), the
two-site-cell and variational-environment circuits, and - through the recorded `minimize_scalar` calls and the drivers'
parameter / energy histories - the exact update rule and trajectory of the rotosolve drivers.  What stays documented-only:
cirq's named-gate matrices and simulator endianness, xmps's SU / U4 maps (see cirq_shim.py)."""
import os

import numpy as np
import pytest

from oracle import qmps_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'refshim_golden.npz')


@pytest.fixture(scope='module')
def g():
    return np.load(GOLD)


ORACLE_BUILDERS = {
    'cnot': O.shallow_cnot_unitary, 'qaoa': O.shallow_qaoa_unitary, 'cnot3': O.shallow_cnot3_unitary,
    'nonuniform': O.shallow_cnot_nonuniform_unitary, 'exactafter4': O.exact_after4_unitary,
    'full': lambda D, p: O.shallow_full_unitary(p),
}
ENERGY_TAGS = ['cnot_D2_d1', 'cnot_D2_d2', 'cnot_D4_d2', 'cnot_D8_d3_xxz', 'cnot_D16_d4', 'qaoa_D4_d2', 'cnot3_D4_d2', 'nonuniform_D2_d2',
               'nonuniform_D4_d2', 'exactafter4_D2_d2', 'exactafter4_D4_d2', 'full_D2']


def tag_D(tag):
    return int(tag.split('_')[1][1:])


def host_class(name):
    from qmps_amd import represent as R
    return {'cnot': R.ShallowCNOTStateTensor, 'qaoa': R.ShallowQAOAStateTensor, 'cnot3': R.ShallowCNOTStateTensor3,
            'nonuniform': R.ShallowCNOTStateTensor_nonuniform, 'exactafter4': R.ExactAfter4, 'full': R.ShallowFullStateTensor}[name]


@pytest.mark.parametrize('tag', ENERGY_TAGS)
def test_ansatz_unitaries_and_energies_match_the_reference_run(tag, g):
    """a-2 / a-7: represent.py:268-423 gate lists and ground_state.py:150-168, both executed by the reference."""
    from qmps_amd import represent as R
    name, D = tag.split('_')[0], tag_D(tag)
    h = g['h_xxz'] if 'xxz' in tag else g['h_tfim']
    for p, U_ref, E_ref in zip(g[f'params_{tag}'], g[f'refshim_U_{tag}'], g[f'refshim_E_{tag}']):
        U = ORACLE_BUILDERS[name](D, p)
        assert np.abs(U - U_ref).max() < 1e-13
        assert np.abs(R.unitary(host_class(name)(D, p)) - U_ref).max() < 1e-13          # the host API mirror's own circuit model
        A = O.unitary_to_tensor(U)
        assert abs(O.energy_closed_form(A, h) - E_ref) < 1e-12
        if D <= 4:
            assert abs(O.energy_statevector(U, h) - E_ref) < 1e-12


def test_stategate_and_state_register(g):
    from qmps_amd import represent as R
    for p, U_ref in zip(g['params_stategate'], g['refshim_U_stategate']):
        assert np.abs(O.state_gate_unitary(p) - U_ref).max() < 1e-13
        assert np.abs(R.unitary(R.StateGate(p)) - U_ref).max() < 1e-13
    # the whole register of State(U, V, 2) (represent.py:258-262), built with the reference's own V
    U = O.shallow_cnot_unitary(2, g['params_cnot_D2_d2'][0])
    psi = O.state_vector(U, g['refshim_state_V_cnot_D2_d2_0'], 2)
    assert np.abs(psi - g['refshim_state_psi_cnot_D2_d2_0']).max() < 1e-13
    # and V itself: first column = vec(cholesky(r)^dagger) / norm (tools.py:97-108, 176-182); the completion is arbitrary
    V = O.get_env_exact(U)
    assert np.abs(V[:, 0] - g['refshim_state_V_cnot_D2_d2_0'][:, 0]).max() < 1e-12


def test_config1_landscape_is_flat_in_the_reference_itself(g):
    """BASELINE.json configs[0] / [1] as written - TFIM, D = 2, depth 1, ShallowCNOTStateTensor(2, [beta, gamma]) - has
    E(beta, gamma) = 0 for every (beta, gamma): a property of the reference's ansatz, not of a restatement.  The reference's own
    objective returns 0 (to rounding) for the six seeded parameter pairs.  Why: with w_j = rx(gamma) rz(beta)|j> (the columns of
    a unitary W) and a = H rx(gamma) rz(beta)|0> (|a_0|^2 = |a_1|^2 = 1/2 for every angle), the gate list rz x rz, rx x rx,
    H(q0), CNOT(q0, q1) (represent.py:301-307) gives  A_sigma[i, j] = a_i (X^i w_j)_sigma.  Then
      sum_sigma A_sigma r A_sigma^+ [i, i]  = |a_i|^2 tr r = tr r / 2,   [0, 1] = a_0 conj(a_1) tr(r W^+ X W),
    so r = 1/2 is the fixed point (tr X = 0), and the one-site density matrix is
      (1/2) sum_{i,j} |a_i|^2 X^i w_j w_j^+ X^i = 1/2:   <X> = <Y> = <Z> = 0.
    <ZZ> = 0 as well (checked numerically below with both oracle constructions over a grid); only <XX> depends on the angles."""
    assert np.abs(g['refshim_E_cnot_D2_d1']).max() < 1e-15
    XX = O.hamiltonian_matrix({'XX': 1})
    ZZ = O.hamiltonian_matrix({'ZZ': 1})
    X1 = O.hamiltonian_matrix({'X': 1})
    Z1 = O.hamiltonian_matrix({'Z': 1})
    xx = []
    for beta in np.linspace(-3, 3, 7):
        for gamma in np.linspace(-3, 3, 7):
            U = O.shallow_cnot_unitary(2, [beta, gamma])
            A = O.unitary_to_tensor(U)
            _, r = O.env_dense_eig(A)
            assert np.abs(r / np.trace(r) - np.eye(2) / 2).max() < 1e-12
            for op in (ZZ, X1, Z1):
                assert abs(O.energy_statevector(U, op)) < 1e-12
                assert abs(O.energy_closed_form(A, op)) < 1e-12
            xx.append(O.energy_closed_form(A, XX))
    assert np.ptp(xx) > 0.5                                   # <XX> does move: the landscape is flat for TFIM, not for every h
    # depth 2 (four angles) has a landscape: the reference-run energies are spread
    assert np.ptp(g['refshim_E_cnot_D2_d2']) > 0.3


def test_nonsparse_cell_and_variational_environment_objectives(g):
    """a-8 / a-9 / a-12 circuits (ground_state.py:251-266, 291-331, 170-229); the SU map itself is NOT pinned (look-up stand-in)."""
    h = g['h_tfim']
    for D in (2, 4):
        for U, E in zip(g[f'U_nonsparse_D{D}'], g[f'refshim_E_nonsparse_D{D}']):
            assert abs(O.energy_closed_form(O.unitary_to_tensor(U), h) - E) < 1e-12
    for U1, U2, E in zip(g['U1_cell'], g['U2_cell'], g['refshim_E_cell']):
        assert abs(O.two_site_cell_energy(U1, U2, h) - E) < 1e-12
        assert abs(O.two_site_cell_energy_closed(O.unitary_to_tensor(U1), O.unitary_to_tensor(U2), h) - E) < 1e-12
    for p, f in zip(g['params_optenv'], g['refshim_optenv']):
        assert abs(O.opt_environment_objective(p, h)[0] - f) < 1e-12


def test_environment_embeddings(g):
    """time_evolve_tools.py:38-74 executed by the reference (its SWAP constant from the stand-in): oracle and host mirror."""
    from qmps_amd import time_evolve_tools as T
    for k, q in enumerate(g['embed_q']):
        for mod in (O, T):
            L, R = mod.put_env_on_left_site(q), mod.put_env_on_right_site(q)
            # rows 0, 1 carry q; rows 2, 3 are scipy's null_space completion (same LAPACK call here and there)
            assert np.abs(L - g['refshim_put_left'][k]).max() < 1e-12
            assert np.abs(R - g['refshim_put_right'][k]).max() < 1e-12
            assert np.abs(L.conj().T @ L - np.eye(4)).max() < 1e-12
        _, n = T.put_env_on_left_site(q, ret_n=True)
        assert abs(n - g['refshim_put_left_n'][k]) < 1e-13 and abs(n - np.linalg.norm(q)) < 1e-13
        assert np.abs(T.get_env_off_left_site(g['refshim_put_left'][k]) - g['refshim_off_left'][k]).max() < 1e-14
        assert np.abs(T.get_env_off_right_site(g['refshim_put_right'][k]) - g['refshim_off_right'][k]).max() < 1e-14
        assert np.abs(g['refshim_off_left'][k] - q / np.linalg.norm(q)).max() < 1e-12      # the round trip the names promise
        assert np.abs(g['refshim_off_right'][k] - q / np.linalg.norm(q)).max() < 1e-12


@pytest.mark.parametrize('name', ['loschmidt', 'loschmidt_full'])
def test_overlap_objective_is_the_reference_circuit(name, g):
    """f-3 / N-2: scripts/loschmidt.py:209-239 executed by the reference: objective == -sqrt|eta|, 2 |psi[0]| == |eta|, and the
    oracle's own 6-qubit circuit gives the same amplitude (up to the eigenvector's phase)."""
    build = (lambda p: O.shallow_cnot_unitary(2, p)) if name == 'loschmidt' else O.shallow_full_unitary
    for w, WW in (('W', g['WW_loschmidt'] if name == 'loschmidt' else g['WW_nte']), ('I', np.eye(4, dtype=complex))):
        for pc, k, f_ref, amp in zip(g[f'{name}_p_cand'], g[f'{name}_ref_idx'], g[f'refshim_{name}_obj_{w}'], g[f'refshim_{name}_psi0_{w}']):
            A = O.unitary_to_tensor(build(g[f'{name}_p_ref'][k]))
            Bt = O.unitary_to_tensor(build(pc))
            eta, r = O.overlap_eta(A, Bt, WW)
            assert abs(O.overlap_objective(A, Bt, WW) - f_ref) < 1e-12
            assert abs(abs(eta) - 2 * abs(amp)) < 1e-12
            assert abs(abs(O.overlap_circuit_amplitude(A, Bt, WW, r)) - abs(amp)) < 1e-12
            assert abs(-np.sqrt(abs(O.overlap_eta_arpack(A, Bt, WW)[0])) - f_ref) < 1e-10
    # candidates near the reference state with W = 1 overlap almost perfectly; far ones do not
    f = g[f'refshim_{name}_obj_I']
    assert f[:4].max() < -0.98 and f[4:].min() > -0.98


@pytest.mark.parametrize('name', ['loschmidt', 'loschmidt_full'])
def test_time_evolution_loop_run_by_the_reference(name, g):
    """N-2: the loop `A_ = tensor(params); res = minimize(obj, params, (A_, WW)); params = res.x` (qmps/new_time_evolve.py:276-292,
    scripts/loschmidt.py:367-375) executed with the reference's objective and gate and scipy's default BFGS: three consecutive
    time steps from three starting points.  The oracle's objective reproduces the value scipy reported at the parameters it
    returned (and at the starting point of every step); a time step starts away from its minimum (W moved the state) and ends close
    to -1; and the lock-step BFGS of the host mirror (`tools.batched_bfgs`, the loop the device drivers implement) driven by the
    ORACLE reaches the reference-run minima from the reference-run starting points."""
    from qmps_amd import tools as T
    build = (lambda p: O.shallow_cnot_unitary(2, p)) if name == 'loschmidt' else O.shallow_full_unitary
    WW = g['WW_loschmidt'] if name == 'loschmidt' else g['WW_nte']
    X, F, F0 = g[f'refshim_evolve_{name}_x'], g[f'refshim_evolve_{name}_f'], g[f'refshim_evolve_{name}_f_start']
    assert np.abs(X[:, 0] - g[f'evolve_{name}_x0']).max() == 0.0
    n_traj, n_steps = F.shape
    for t in range(n_traj):
        for k in range(n_steps):
            A = O.unitary_to_tensor(build(X[t, k]))
            assert abs(O.overlap_objective(A, O.unitary_to_tensor(build(X[t, k + 1])), WW) - F[t, k]) < 1e-12
            assert abs(O.overlap_objective(A, A, WW) - F0[t, k]) < 1e-12
    assert np.all(F < F0 - 1e-5) and F.max() < -0.99

    def batch_f(k):
        A = [O.unitary_to_tensor(build(X[t, k])) for t in range(n_traj)]
        # (rows are trajectory-major: row i of a batch of n_traj G candidates belongs to trajectory i // G)
        return lambda Z: np.array([O.overlap_objective(A[i // (len(Z) // n_traj)], O.unitary_to_tensor(build(z)), WW) for i, z in enumerate(Z)])
    worst = 0.0
    for k in range(n_steps):
        f = batch_f(k)
        res = T.batched_bfgs(f, f, X[:, k].copy(), maxiter=200)
        worst = max(worst, np.abs(res['fun'] - F[:, k]).max())
        # (never meaningfully above scipy's minimum; scipy's forward differences stop it ~1e-8 short now and then)
        assert np.all(res['fun'] < F[:, k] + 1e-6), (k, res['fun'] - F[:, k])
    print(name, 'max |f_lockstep - f_reference_run|', worst)


def test_bounded_scalar_minimiser_reproduces_every_recorded_scipy_call(g):
    """tools.py:451 / rotosolve.py:237: the 486 `minimize_scalar(f, bounds=[-pi, pi])` calls the reference made while the
    fixtures were generated - coefficients of f, scipy's x, f(x) and evaluation count.  The oracle's restatement of the bounded
    Brent search takes the same decisions: same evaluation count, x within 1e-10 (the coefficients were recovered by a
    least-squares fit, 1e-16 relative)."""
    fits = g['refshim_roto_fits']
    assert len(fits) > 400
    n_local = 0
    for a, b, c, d, x, fx, nfev in fits:
        f = lambda t: a * np.sin(2 * t) + b * np.cos(2 * t) + c * np.sin(t) + d * np.cos(t)      # noqa: E731
        xx, ff, nn = O.fminbound(f, -np.pi, np.pi)
        assert nn == int(nfev) and abs(xx - x) < 1e-10 and abs(ff - fx) < 1e-12
        P, u, Q, v = np.hypot(a, b), np.arctan2(b, a), np.hypot(c, d), np.arctan2(d, c)
        assert abs(O.double_sinusoid_fminbound(P, u, Q, v) - x) < 1e-9
        xg = O.double_sinusoid_argmin(P, u, Q, v)
        n_local += f(xg) < fx - 1e-6
    # the reference's rule is a LOCAL search: a tenth of its answers are not the global minimiser of the fit
    assert 20 < n_local < len(fits) // 4


def _oracle_eps(D, h):
    def eps(p):
        return O.energy_closed_form(O.unitary_to_tensor(O.shallow_cnot_unitary(D, p)), h)
    return eps


@pytest.mark.parametrize('tag,D,hname', [('D2_d2', 2, 'h_tfim'), ('D4_d2', 4, 'h_tfim'), ('D8_d3_xxz', 8, 'h_xxz')])
def test_host_double_rotosolve_follows_the_reference_run(tag, D, hname, g):
    """a-10: qmps/tools.py:422-457 run by the reference on ITS objective; the host mirror `tools.double_rotosolve` on the oracle's."""
    from qmps_amd import tools
    x0, E_ref, x_ref = g[f'roto_{tag}_x0'], g[f'refshim_droto_{tag}_E'], g[f'refshim_droto_{tag}_x']
    sweeps = E_ref.shape[0]
    for r in range(len(x0)):
        res = tools.double_rotosolve(_oracle_eps(D, g[hname]), x0[r].copy(), sweeps, disp=False)
        assert np.abs(np.array(res.history) - E_ref[:, r]).max() < 1e-8
        assert np.abs(res.x - x_ref[-1, r]).max() < 1e-6


@pytest.mark.parametrize('tag,D', [('D2_d2', 2), ('D4_d2', 4)])
def test_host_old_api_rotosolve_follows_the_reference_run(tag, D, g):
    """qmps/rotosolve.py:154-181 and :183-241 (state function + H), run by the reference with State(U, V_exact, 2) as the state."""
    from qmps_amd import rotosolve as RS
    h = g['h_tfim']
    Hfull = np.kron(np.kron(np.eye(D), h), np.eye(D))

    def state_fn(x):
        U = O.shallow_cnot_unitary(D, x)
        return O.state_vector(U, O.get_env_exact(U), 2)
    x0 = g[f'roto_old_{tag}_x0']
    E1, X1 = g[f'refshim_roto_old_{tag}_E'], g[f'refshim_roto_old_{tag}_x']
    E2, X2 = g[f'refshim_droto_old_{tag}_E'], g[f'refshim_droto_old_{tag}_x']
    sweeps = E1.shape[0]
    follow = 0
    for r in range(len(x0)):
        es, S = RS.rotosolve(Hfull, state_fn, x0[r].copy(), (), sweeps)
        # (single-frequency updates at a flat direction - atan2 of two rounding-level numbers - may differ between evaluators: the
        # ENERGY history is what the trajectories share; parameters are compared where they agree)
        assert np.abs(np.array(es) - E1[:, r]).max() < 1e-8
        follow += np.abs(np.arctan2(np.sin(np.array(S) - X1[:, r]), np.cos(np.array(S) - X1[:, r]))).max() < 1e-6
        es2, x2 = RS.double_rotosolve(Hfull, state_fn, x0[r].copy(), (), sweeps)
        assert np.abs(es2 - E2[:, r]).max() < 1e-8
        assert np.abs(x2 - X2[-1, r]).max() < 1e-6
    assert follow >= len(x0) - 2


def test_device_source_of_the_update_rule_takes_scipys_recorded_decisions(g):
    """qmps_amd/csrc/qmps_roto_rule.h - the source the gfx950 kernels include - built for the host (tests/csrc/libroto_emu.so):
    every one of the reference's recorded `minimize_scalar` calls, same minimiser to 1e-9 (xatol is 1e-5; a value-only search
    resolves ~sqrt(eps)); the GLOBAL rule is never worse on the fitted curve and differs in the calls where scipy's is local."""
    import ctypes
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, 'csrc', 'libroto_emu.so')
    if not os.path.exists(so):
        pytest.skip('tests/csrc not built (python -c "import __graft_entry__ as g; g.build()")')
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.roto_emu_steps.argtypes = [ctypes.c_long, dp, ctypes.c_int, dp]
    fits = g['refshim_roto_fits']
    abcd = np.ascontiguousarray(fits[:, :4])
    out = np.empty((2, len(fits)))
    for rule in (0, 1):
        lib.roto_emu_steps(len(fits), abcd.ctypes.data_as(dp), rule, out[rule].ctypes.data_as(dp))
    assert np.abs(out[0] - fits[:, 4]).max() < 1e-9
    f = lambda k, x: fits[k, 0] * np.sin(2 * x) + fits[k, 1] * np.cos(2 * x) + fits[k, 2] * np.sin(x) + fits[k, 3] * np.cos(x)   # noqa: E731
    better = 0
    for k in range(len(fits)):
        assert f(k, out[1, k]) <= f(k, out[0, k]) + 1e-12
        assert f(k, out[1, k]) <= f(k, np.linspace(-np.pi, np.pi, 20001)).min() + 1e-12           # ... and it IS global
        better += f(k, out[1, k]) < f(k, out[0, k]) - 1e-6
    assert 20 < better < len(fits) // 4
    assert np.abs(out).max() <= np.pi


def test_variational_overlap_route_run_by_the_reference(g):
    """`get_overlap`'s objective closure (qmps/time_evolve_tools.py:98-128) and `obj_state` (qmps/new_time_evolve.py:223-247) executed by
    the reference: the circuit amplitude for a GIVEN environment is the Rayleigh form of the mixed transfer map,
    psi[0] = <r^, T(r^)>_F / 2 on 6 qubits (Bell pair) and / sqrt(2) on 5 (StateGate) - what `qmps_overlap_amplitude` computes."""
    def rayleigh(A, Bt, WW, r):
        C = np.tensordot(WW, O.merge(A, A), [1, 0])
        Bm = O.merge(Bt, Bt)
        rh = r / np.linalg.norm(r)
        return np.vdot(rh, sum(C[s] @ rh @ Bm[s].conj().T for s in range(4)))
    for k in range(3):
        A = O.unitary_to_tensor(O.shallow_full_unitary(g['varenv_p1'][k]))
        Bt = O.unitary_to_tensor(O.shallow_full_unitary(g['varenv_p2'][k]))
        for rs, f_ref in zip(g['varenv_probe_rs'][k], g['refshim_get_overlap_obj'][k]):
            r = (rs[:4] + 1j * rs[4:]).reshape(2, 2)
            assert abs(-abs(rayleigh(A, Bt, np.eye(4), r)) - f_ref) < 1e-12
            assert abs(-2 * abs(O.overlap_circuit_amplitude(A, Bt, np.eye(4), r / np.linalg.norm(r))) - f_ref) < 1e-12
        # the Nelder-Mead minimum the reference returns is a LOCAL estimate of the map's numerical radius: bounded by its largest
        # singular value (and, being local, not always beyond |eta|: -0.478 against |eta| = 0.583 in the first case)
        smax = np.linalg.svd(O.transfer_matrix(np.tensordot(np.eye(4), O.merge(A, A), [1, 0]), O.merge(Bt, Bt)), compute_uv=False)[0]
        assert -smax - 1e-12 < g['refshim_get_overlap_min'][k] <= g['refshim_get_overlap_obj'][k].min() + 1e-12
        # obj_state: psi[0] of the 5-qubit register
        p_ = g['obj_state_p'][k]
        WW = g['WW_nte'] if k else np.eye(4)
        r = O.state_gate_unitary(p_[15:])[:, 0].reshape(2, 2)
        assert abs(rayleigh(A, Bt, WW, r) / np.sqrt(2) - g['refshim_obj_state_psi'][k][0]) < 1e-12
        assert abs(np.linalg.norm(g['refshim_obj_state_psi'][k]) - 1) < 1e-12


def test_host_obj_state_is_the_reference_register(g):
    """`qmps_amd.new_time_evolve.obj_state` (host state-vector pass over the package's gate objects) against the register the reference's
    own function returned.  Entries 0..15 read rows 0, 1 of L = put_env_on_right_site(r^+) (fixed by r); entries 16..31 read its
    null_space() completion rows, the same scipy routine on both sides."""
    from qmps_amd import new_time_evolve as N
    from qmps_amd.tools import unitary_to_tensor
    from qmps_amd.represent import unitary
    for k in range(3):
        A = unitary_to_tensor(unitary(N.gate(g['varenv_p1'][k])))
        psi = N.obj_state(g['obj_state_p'][k], A, g['WW_nte'] if k else np.eye(4))
        assert psi.shape == (32,)
        assert np.abs(psi[:16] - g['refshim_obj_state_psi'][k][:16]).max() < 1e-12
        assert abs(np.linalg.norm(psi[16:]) - np.linalg.norm(g['refshim_obj_state_psi'][k][16:])) < 1e-12
        assert abs(np.vdot(psi, N.obj_H() @ psi).real + abs(g['refshim_obj_state_psi'][k][0]) ** 2) < 1e-12


def test_nsphere_is_the_reference_parametrisation(g):
    """time_evolve_tools.Nsphere (qmps/time_evolve_tools.py:25-36), run by the reference: points on the unit sphere."""
    from qmps_amd import time_evolve_tools as T
    for v, x in zip(g['nsphere_v'], g['refshim_nsphere']):
        assert np.abs(T.Nsphere(v) - x).max() < 1e-14 and abs(np.linalg.norm(x) - 1) < 1e-14


def test_the_reference_self_tests_ran_green_over_the_stand_ins(g):
    """make_refshim_golden.py section 8 executes qmps/new_time_evolve.py:run_tests (the reference's own asserts: embeddings, 1- and 2-site
    circuit identities with R and L, the 6-qubit overlap identity) over cirq_shim: the generator stops at the first failing assert, the
    fixture records how many iterations passed.  The oracle satisfies the same 6-qubit identity with ITS fixed points."""
    assert g['refshim_reference_selftests_passed'][0] >= 5
    rng = np.random.default_rng(4)
    for _ in range(4):
        A, Bt = (O.unitary_to_tensor(u) for u in O.haar_unitaries(rng, 4, 2))
        M = O.transfer_matrix(O.merge(A, A), O.merge(Bt, Bt))
        w, v = np.linalg.eig(M)
        k = int(np.argmax(abs(w)))
        wl, vl = np.linalg.eig(M.conj().T)
        kl = int(np.argmax(abs(wl)))
        r, l = v[:, k].reshape(2, 2), vl[:, kl].reshape(2, 2)
        r, l = r / np.linalg.norm(r), l / np.linalg.norm(l)
        # scripts/loschmidt.py:228-238 with L carrying l^+ instead of r^+: 2 psi[0] = eta tr(l^+ r)
        U, Up = O.tensor_to_unitary(A), O.tensor_to_unitary(Bt)
        ops = [(O.HAD, [3]), (O.CNOT, [3, 4]), (U, [2, 3]), (U, [1, 2]), (O.put_env_on_right_site(l.conj().T), [0, 1]),
               (O.put_env_on_left_site(r), [4, 5]), (Up.conj().T, [1, 2]), (Up.conj().T, [2, 3]), (O.CNOT, [3, 4]), (O.HAD, [3])]
        assert abs(2 * O._run_ops(6, ops)[0] - w[k] * np.trace(l.conj().T @ r)) < 1e-12
