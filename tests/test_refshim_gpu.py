"""GPU tests: the HIP path (through the C-ABI) against tests/golden/refshim_golden.npz - outputs of the REFERENCE'S OWN Python run in
the build container over the documented cirq / xmps stand-ins (tests/golden/make_refshim_golden.py, cirq_shim.py).  Tolerance
of the north star: 1e-10 on energies and overlap eigenvalues.  Nothing here reads /root/reference."""
import os

import numpy as np
import pytest

from oracle import qmps_oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'refshim_golden.npz')


@pytest.fixture(scope='module')
def g():
    return np.load(GOLD)


#            fixture tag          D   device ansatz kind
CASES = [('cnot_D2_d1', 2, 0), ('cnot_D2_d2', 2, 0), ('cnot_D4_d2', 4, 0), ('cnot_D8_d3_xxz', 8, 0), ('cnot_D16_d4', 16, 0),
         ('qaoa_D4_d2', 4, 1), ('full_D2', 2, 2), ('cnot3_D4_d2', 4, 3), ('nonuniform_D2_d2', 2, 4), ('nonuniform_D4_d2', 4, 4),
         ('exactafter4_D2_d2', 2, 5), ('exactafter4_D4_d2', 4, 5)]


@pytest.mark.parametrize('tag,D,kind', CASES)
def test_device_energies_from_parameters_match_the_reference_objective(tag, D, kind, g, engine_factory):
    """SparseFullEnergyOptimizer.objective_function_exact_environment (ground_state.py:150-168) as the reference computes it:
    parameters in, energy out; the device builds the ansatz unitary, the tensor, the environment and the energy."""
    h = g['h_xxz'] if 'xxz' in tag else g['h_tfim']
    eng = engine_factory(D, 4096)
    E, it, st = eng.energies_from_params(kind, g[f'params_{tag}'], h)
    assert np.all(st == 0)
    assert np.abs(E[:, 0] - g[f'refshim_E_{tag}']).max() < 1e-10
    # and from the reference-built unitaries (the benchmark's input form)
    E2, _, st2 = eng.energies(g[f'refshim_U_{tag}'], h, kind='unitary')
    assert np.all(st2 == 0) and np.abs(E2[:, 0] - g[f'refshim_E_{tag}']).max() < 1e-10


def test_device_nonsparse_cell_and_variational_environment(g, engine_factory):
    h = g['h_tfim']
    for D in (2, 4):
        E, _, st = engine_factory(D, 4096).energies(g[f'U_nonsparse_D{D}'], h, kind='unitary')
        assert np.all(st == 0) and np.abs(E[:, 0] - g[f'refshim_E_nonsparse_D{D}']).max() < 1e-10
    eng = engine_factory(2, 4096)
    out = eng.cell2_energies(g['U1_cell'], g['U2_cell'], h)
    assert np.abs(np.asarray(out[0]).reshape(len(g['refshim_E_cell']), -1)[:, 0] - g['refshim_E_cell']).max() < 1e-10
    f = eng.opt_env_objective(g['params_optenv'], h)
    assert np.abs(f - g['refshim_optenv']).max() < 1e-10


@pytest.mark.parametrize('name,kind', [('loschmidt', 0), ('loschmidt_full', 2)])
def test_device_overlap_is_the_reference_circuit_amplitude(name, kind, g, engine_factory):
    """scripts/loschmidt.py:209-239 run by the reference: |eta| = 2 |psi[0]| and objective = -sqrt(2 |psi[0]|), D = 2, candidates
    given as PARAMETERS (device ansatz) and as tensors."""
    build = (lambda p: O.shallow_cnot_unitary(2, p)) if name == 'loschmidt' else O.shallow_full_unitary
    eng = engine_factory(2, 4096)
    for w, WW in (('W', g['WW_loschmidt'] if name == 'loschmidt' else g['WW_nte']), ('I', np.eye(4, dtype=complex))):
        A = np.stack([O.unitary_to_tensor(build(g[f'{name}_p_ref'][k])) for k in g[f'{name}_ref_idx']])
        eta, _, st = eng.overlaps(A, g[f'{name}_p_cand'], WW, kind='params', ansatz=kind)
        assert np.all(st == 0)
        assert np.abs(np.abs(eta) - 2 * np.abs(g[f'refshim_{name}_psi0_{w}'])).max() < 1e-10
        assert np.abs(-np.sqrt(np.abs(eta)) - g[f'refshim_{name}_obj_{w}']).max() < 1e-10
        cand = np.stack([O.unitary_to_tensor(build(p)) for p in g[f'{name}_p_cand']])
        eta2, _, st2 = eng.overlaps(A, cand, WW)
        assert np.all(st2 == 0) and np.abs(np.abs(eta2) - 2 * np.abs(g[f'refshim_{name}_psi0_{w}'])).max() < 1e-10


@pytest.mark.parametrize('name,cls_name', [('loschmidt', 'ShallowCNOTStateTensor'), ('loschmidt_full', 'ShallowFullStateTensor')])
def test_device_time_evolution_reaches_the_reference_run_minima(name, cls_name, g):
    """N-2 against a reference-EXECUTED trajectory: the loop of qmps/new_time_evolve.py:276-292 / scripts/loschmidt.py:367-375 run by the
    reference's own objective and gate with scipy's default BFGS (tests/golden/make_refshim_golden.py, section 6).  `evolve(method='BFGS')`
    - the device-resident optimiser at D = 2 - started from the reference run's parameters of every time step reaches the reference
    run's minimum of that step (1e-6: scipy's forward differences against central ones), starts the step at the same objective
    (1e-10), and over the three consecutive steps of one call stays on the reference run's states (fidelity per site)."""
    from qmps_amd import new_time_evolve as NT, represent as R
    cls = getattr(R, cls_name)
    build = (lambda p: O.shallow_cnot_unitary(2, p)) if name == 'loschmidt' else O.shallow_full_unitary
    WW = g['WW_loschmidt'] if name == 'loschmidt' else g['WW_nte']
    X, F, F0 = g[f'refshim_evolve_{name}_x'], g[f'refshim_evolve_{name}_f'], g[f'refshim_evolve_{name}_f_start']
    n_traj, n_steps = F.shape
    worst = 0.0
    for k in range(n_steps):
        H, info = NT.evolve(X[:, k].copy(), WW, 1, method='BFGS', D=2, state_tensor=cls, tol=1e-13, options={'maxiter': 200}, return_info=True)
        f_start, f_end = info['fun'][0][0], info['fun'][0][-1]
        assert np.abs(f_start - F0[:, k]).max() < 1e-10
        worst = max(worst, np.abs(f_end - F[:, k]).max())
        assert np.all(f_end < F[:, k] + 1e-6), (k, f_end - F[:, k])
        assert np.abs(f_end - F[:, k]).max() < 1e-5, (k, f_end - F[:, k])
        for t in range(n_traj):      # the same physical state as the reference run's: |<A(x_dev), A(x_ref)>| per site
            o = abs(O.overlap_eta(O.unitary_to_tensor(build(H[1][t])), O.unitary_to_tensor(build(X[t, k + 1])), np.eye(4))[0])
            assert abs(o - 1.0) < 1e-4, (k, t, o)
    print(name, 'max |f_device - f_reference_run| over', n_traj * n_steps, 'time steps:', worst)
    # three consecutive steps in one call, from the reference run's starting points
    H3, info3 = NT.evolve(X[:, 0].copy(), WW, n_steps, method='BFGS', D=2, state_tensor=cls, tol=1e-13, options={'maxiter': 200}, return_info=True)
    f3 = np.array([f[-1] for f in info3['fun']]).T           # (n_traj, n_steps)
    assert np.abs(f3 - F).max() < 1e-4 and f3.max() < -0.99
    for t in range(n_traj):
        o = abs(O.overlap_eta(O.unitary_to_tensor(build(H3[-1][t])), O.unitary_to_tensor(build(X[t, -1])), np.eye(4))[0])
        assert abs(o - 1.0) < 1e-3, (t, o)


@pytest.mark.parametrize('tag,D,hname', [('D2_d2', 2, 'h_tfim'), ('D4_d2', 4, 'h_tfim'), ('D8_d3_xxz', 8, 'h_xxz')])
def test_device_double_rotosolve_follows_the_reference_run(tag, D, hname, g, engine_factory):
    """qmps/tools.py:422-457 (what Optimizer.optimize('Rotosolve') runs) executed by the reference on its own objective: the
    device driver, with its DEFAULT rule, must return the same energy history (1e-8) and the same parameters."""
    x0, E_ref, x_ref = g[f'roto_{tag}_x0'], g[f'refshim_droto_{tag}_E'], g[f'refshim_droto_{tag}_x']
    eng = engine_factory(D, 4096)
    eng.set_hamiltonian(g[hname])
    es, p = eng.double_rotosolve(0, x0, E_ref.shape[0])
    assert np.abs(es - E_ref).max() < 1e-8, np.abs(es - E_ref).max(0)
    assert np.abs(p - x_ref[-1]).max() < 1e-6
    # one sweep only: the intermediate record too
    es1, p1 = eng.double_rotosolve(0, x0, 1)
    assert np.abs(es1[0] - E_ref[0]).max() < 1e-8 and np.abs(p1 - x_ref[0]).max() < 1e-6


@pytest.mark.parametrize('tag,D', [('D2_d2', 2), ('D4_d2', 4)])
def test_device_single_frequency_rotosolve_follows_the_reference_run(tag, D, g, engine_factory):
    """qmps/rotosolve.py:154-181 executed by the reference with State(U, V_exact, 2) as its state function."""
    x0, E_ref, X_ref = g[f'roto_old_{tag}_x0'], g[f'refshim_roto_old_{tag}_E'], g[f'refshim_roto_old_{tag}_x']
    eng = engine_factory(D, 4096)
    eng.set_hamiltonian(g['h_tfim'])
    es, p = eng.rotosolve(0, x0, E_ref.shape[0])
    assert np.abs(es - E_ref).max() < 1e-8, np.abs(es - E_ref).max(0)
    d = np.abs(np.arctan2(np.sin(p - X_ref[-1]), np.cos(p - X_ref[-1]))).max(1)
    assert (d < 1e-6).sum() >= len(x0) - 2           # flat directions (rz on |0>): see tests/test_rotosolve_gpu.py's docstring
    # the double-frequency driver of the same old API (rotosolve.py:183-241) is the same rule as tools.py's
    E2, X2 = g[f'refshim_droto_old_{tag}_E'], g[f'refshim_droto_old_{tag}_x']
    es2, p2 = eng.double_rotosolve(0, x0, E2.shape[0])
    assert np.abs(es2 - E2).max() < 1e-8 and np.abs(p2 - X2[-1]).max() < 1e-6


def test_device_update_rule_takes_scipys_recorded_decisions(g, engine_factory):
    """Every `minimize_scalar(f, bounds=[-pi, pi])` call the reference made while the fixtures were generated (coefficients of f,
    scipy's x): the device build of the rule (qmps_roto_rule_probe: the function roto_update_kernel and the whole-run D = 2, 8
    kernels call) returns scipy's minimiser for every one of them - 1e-9, four orders below scipy's own xatol.  The GLOBAL rule
    is the documented departure: never worse on the fitted curve, different in the calls where scipy's answer is a local minimum."""
    from qmps_amd import _lib as L
    fits = g['refshim_roto_fits']
    eng = engine_factory(2, 4096)
    th = eng.roto_rule_probe(fits[:, :4], L.ROTO_REFERENCE)
    assert np.abs(th - fits[:, 4]).max() < 1e-9, np.sort(np.abs(th - fits[:, 4]))[-5:]
    tg = eng.roto_rule_probe(fits[:, :4], L.ROTO_GLOBAL_ARGMIN)
    f = lambda k, x: fits[k, 0] * np.sin(2 * x) + fits[k, 1] * np.cos(2 * x) + fits[k, 2] * np.sin(x) + fits[k, 3] * np.cos(x)   # noqa: E731
    better = 0
    for k in range(len(fits)):
        assert f(k, tg[k]) <= f(k, np.linspace(-np.pi, np.pi, 20001)).min() + 1e-12
        better += f(k, tg[k]) < fits[k, 5] - 1e-6
    assert 20 < better < len(fits) // 4
    # the rule is per call, not per engine: after a run with the global rule the default is the reference's again
    x0, E_ref = g['roto_D2_d2_x0'], g['refshim_droto_D2_d2_E']
    eng.set_hamiltonian(g['h_tfim'])
    es_glob, _ = eng.double_rotosolve(0, x0, 2, rule=L.ROTO_GLOBAL_ARGMIN)
    es_ref, _ = eng.double_rotosolve(0, x0, 2)
    assert np.isfinite(es_glob).all() and np.abs(es_ref - E_ref).max() < 1e-8


def test_variational_overlap_route_matches_the_reference_run(g):
    """`get_overlap` (qmps/time_evolve_tools.py:95-131) and `obj_state` (qmps/new_time_evolve.py:223-247) through the drop-in modules:
    the objective closure at the recorded environments to 1e-12, the Nelder-Mead minimum the reference returned (same scipy simplex
    on the device objective) to 1e-7, psi[0] of the 5-qubit register to 1e-12."""
    from qmps_amd import new_time_evolve as N
    from qmps_amd import time_evolve_tools as T
    from qmps_amd.represent import unitary
    from qmps_amd.tools import unitary_to_tensor
    for k in range(3):
        A = unitary_to_tensor(unitary(T.gate(g['varenv_p1'][k])))
        Bt = unitary_to_tensor(unitary(T.gate(g['varenv_p2'][k])))
        rs = g['varenv_probe_rs'][k]
        q = (rs[:, :4] + 1j * rs[:, 4:]).reshape(-1, 2, 2)
        amp = T.overlap_amplitudes(A, np.repeat(Bt[None], len(q), 0), np.eye(4), q)
        assert np.abs(-2 * np.abs(amp) - g['refshim_get_overlap_obj'][k]).max() < 1e-12
        f = T.get_overlap(g['varenv_p1'][k], g['varenv_p2'][k], initial=g['varenv_initial'][k].copy(), options={'disp': False})
        assert abs(f - g['refshim_get_overlap_min'][k]) < 1e-7, (k, f, g['refshim_get_overlap_min'][k])
        WW = g['WW_nte'] if k else np.eye(4)
        a5 = N.obj_state_amplitudes(g['obj_state_p'][k], A, WW)
        assert abs(a5[0] - g['refshim_obj_state_psi'][k][0]) < 1e-12
        assert abs(N.obj_state(g['obj_state_p'][k], A, WW)[0] - a5[0]) < 1e-12


def test_the_reference_self_test_suite_through_the_drop_in_module(g):
    """qmps/new_time_evolve.py:run_tests - the identities the reference asserts about its own circuits - with the fixed points coming from
    the device (right AND left: the left one is the right fixed point of the daggered tensors, the convention the reference's asserts pin,
    fixture `refshim_reference_selftests_passed`: the original suite executed green over the stand-ins)."""
    from qmps_amd import new_time_evolve as N
    assert g['refshim_reference_selftests_passed'][0] >= 5
    N.run_tests(6, rng=np.random.default_rng(9))
