"""GPU tests of SURVEY 8(f)-2: the device-resident rotosolve drivers against ORACLE-DRIVEN trajectories.

The reference drivers (qmps/rotosolve.py:154-181 single-frequency, qmps/tools.py:422-457 double-frequency) are replayed
on the CPU with the C oracle as the evaluator (oracle ansatz circuit -> unitary -> tensor -> power-iteration
environment -> closed-form energy) and the oracle's update rules; the device runs (`qmps_rotosolve`,
`qmps_double_rotosolve`: ansatz, environment, energy, fit and update all on the GPU) must follow them:
sweep energies within 1e-8 of the oracle-driven run, and the oracle's energy AT THE DEVICE'S FINAL PARAMETERS equal to
the device's (the ansatz families have flat directions - rz on a |0> input is a phase - along which atan2(0, 0) moves
a parameter arbitrarily without changing the state, so parameter vectors are compared through the energies they give,
and directly only where the two runs stay within 1e-7 of each other)."""
import numpy as np
import pytest

from oracle import qmps_oracle as O

pytestmark = pytest.mark.gpu

KINDS = {0: ('ShallowCNOT', O.shallow_cnot_unitary, 2), 1: ('ShallowQAOA', O.shallow_qaoa_unitary, 2),
         3: ('ShallowCNOT3', O.shallow_cnot3_unitary, 3)}
SHIFTS3 = np.array([0.0, np.pi / 2, -np.pi / 2])
SHIFTS6 = np.array([0.0, np.pi, np.pi / 2, -np.pi / 2, np.pi / 4, -np.pi / 4])


def wrap(x):
    return np.arctan2(np.sin(x), np.cos(x))


def oracle_energies(c_oracle, builder, D, params, h):
    A = np.stack([O.unitary_to_tensor(builder(D, p)) for p in params])
    out = c_oracle.energy_batch(A, h, tol=1e-15, max_iter=200000)   # evaluator noise well below the 1e-8 bar (it grows ~10x per sweep)
    return out['E'].sum(1), out['status']


def oracle_trajectory(c_oracle, builder, D, P0, h, sweeps, double, global_argmin=False):
    """Lock-step replay of the reference driver for R restarts; an evaluation without a valid environment leaves the
    restart's parameter untouched (the device rule).  Double-frequency update: the reference's bounded scalar search (tools.py:451;
    oracle `fminbound`, pinned call by call to scipy's recorded answers) unless `global_argmin`.  Returns (energies (sweeps, R), params, ever_invalid (R,),
    list of (six samples, theta) for the double-frequency updates)."""
    params = np.array(P0, dtype=float, copy=True)
    R, P = params.shape
    shifts = SHIFTS6 if double else SHIFTS3
    es, bad, fits = [], np.zeros(R, bool), []
    for _ in range(sweeps):
        for i in range(P):
            batch = np.repeat(params[:, None, :], len(shifts), axis=1)
            batch[:, :, i] += shifts
            e, st = oracle_energies(c_oracle, builder, D, batch.reshape(-1, P), h)
            e, st = e.reshape(R, -1), st.reshape(R, -1)
            for r in range(R):
                if np.any(st[r] != 0):
                    bad[r] = True
                    continue
                if double:
                    th = (O.double_sinusoid_argmin if global_argmin else O.double_sinusoid_fminbound)(*O.double_sinusoid_coefficients(*e[r]))
                    fits.append((e[r].copy(), th))
                    params[r, i] += th                                   # tools.py:453: not re-wrapped
                else:
                    params[r, i] = wrap(params[r, i] + O.rotosolve_update(*e[r]))   # rotosolve.py:175-177
        e, st = oracle_energies(c_oracle, builder, D, params, h)
        bad |= st != 0
        es.append(e)
    return np.array(es), params, bad, fits


@pytest.mark.parametrize('D,kind', [(2, 0), (4, 0), (2, 1), (4, 1), (2, 3), (4, 3)])
def test_single_frequency_trajectory_follows_the_oracle(D, kind, c_oracle, engine_factory):
    name, builder, per = KINDS[kind]
    rng = np.random.default_rng(100 * D + kind)
    R, sweeps = 24, 3
    depth = 1 if D == 2 else 2
    P0 = rng.standard_normal((R, per * depth))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    es_ref, p_ref, bad, _ = oracle_trajectory(c_oracle, builder, D, P0, h[None], sweeps, double=False)
    eng = engine_factory(D, 4096)
    eng.set_hamiltonian(h)
    es, p = eng.rotosolve(kind, P0, sweeps)
    good = ~bad
    assert good.sum() >= R - 4, name
    # (a restart that passes a nearly flat direction - atan2 of two numbers at rounding level - may leave the oracle's
    # trajectory for good: allow a few, require the rest to follow it to 1e-8)
    # (the QAOA family, X**beta and ZZ**gamma with period 2 in the exponent, runs along singular environments (beta = 0):
    # its trajectories amplify the evaluator's rounding ~100x per sweep - 1e-6 there, 1e-8 for the CNOT families)
    follows = good & (np.abs(es - es_ref).max(0) < (1e-6 if kind == 1 else 1e-8))
    assert follows.sum() >= R - 4, (name, np.abs(es - es_ref).max(0))
    e_at_p, st_at_p = oracle_energies(c_oracle, builder, D, p, h[None])
    assert np.abs(e_at_p - es[-1])[good & (st_at_p == 0)].max() < 1e-9, name     # the device's parameters give the device's energies
    assert (np.abs(wrap(p - p_ref)).max(1) < 1e-7).sum() >= 1 or D == 2, name


@pytest.mark.parametrize('D,kind,global_argmin', [(2, 0, False), (4, 0, False), (4, 3, False), (2, 0, True), (4, 0, True)])
def test_double_frequency_trajectory_follows_the_oracle(D, kind, global_argmin, c_oracle, engine_factory):
    from qmps_amd import _lib as L
    name, builder, per = KINDS[kind]
    rng = np.random.default_rng(300 * D + kind)
    R, sweeps = 24, 2
    depth = 1 if D == 2 else 2
    P0 = rng.standard_normal((R, per * depth))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    es_ref, p_ref, bad, fits = oracle_trajectory(c_oracle, builder, D, P0, h[None], sweeps, double=True, global_argmin=global_argmin)
    eng = engine_factory(D, 4096)
    eng.set_hamiltonian(h)
    es, p = eng.double_rotosolve(kind, P0, sweeps, rule=L.ROTO_GLOBAL_ARGMIN if global_argmin else L.ROTO_REFERENCE)
    good = ~bad
    assert good.sum() >= R - 4
    assert np.abs(es - es_ref)[:, good].max() < 1e-8
    e_at_p, st_at_p = oracle_energies(c_oracle, builder, D, p, h[None])
    # (one restart may end next to a degenerate transfer spectrum - eigenvalues 1 and 1 - 3e-9 seen at D = 2: the fixed point, and
    # with it the energy, is then defined to ~1e-8 only, whoever evaluates it)
    dev = np.abs(e_at_p - es[-1])[good & (st_at_p == 0)]
    assert (dev < 1e-9).sum() >= len(dev) - 1 and dev.max() < 1e-7
    # the update rule against the reference's own call, scipy's minimize_scalar on the same samples (tools.py:451)
    same = 0
    for M, th in fits:
        Pq, u, Q, v = O.double_sinusoid_coefficients(*M)
        f = lambda x: Pq * np.sin(2 * x + u) + Q * np.sin(x + v)
        ts = O.double_rotosolve_update(*M)
        if global_argmin:
            # QMPS_ROTO_GLOBAL_ARGMIN: never worse than scipy's (local) minimiser on the fitted curve, equal when both sit in one basin
            assert f(th) <= f(ts) + 1e-12
            if abs(wrap(th - ts)) < 1e-2:
                same += 1
                assert abs(wrap(th - ts)) < 5e-5
        else:
            # QMPS_ROTO_REFERENCE (the default): scipy's decision, every time (the trajectory above IS the device's: es == es_ref)
            assert abs(wrap(th - ts)) < 5e-5
            same += 1
    assert same == len(fits) if not global_argmin else same > len(fits) // 4


@pytest.mark.parametrize('double', [False, True])
def test_config3_trajectories_follow_the_oracle(double, c_oracle, engine_factory):
    """BASELINE.json configs[3] exactly: Heisenberg XXZ, D = 8, ShallowCNOT depth 3 (6 angles), 256 random restarts x 3 angle
    samples (single frequency) - and the six-sample double-frequency driver - replayed with the C oracle as evaluator."""
    name, builder, per = KINDS[0]
    rng = np.random.default_rng(8300 + int(double))
    D, R, sweeps = 8, 256, 2
    P0 = rng.standard_normal((R, 6))
    h = O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})
    es_ref, p_ref, bad, _ = oracle_trajectory(c_oracle, builder, D, P0, h[None], sweeps, double=double)
    eng = engine_factory(D, 4096)
    eng.set_hamiltonian(h)
    es, p = (eng.double_rotosolve if double else eng.rotosolve)(0, P0, sweeps)
    good = ~bad
    assert good.sum() >= R - 8
    # a restart that crosses a nearly flat direction may leave the oracle's trajectory for good (see the module docstring):
    # allow a few of the 256, require the others to follow it to 1e-8
    follows = good & (np.abs(es - es_ref).max(0) < 1e-8)
    assert follows.sum() >= R - 12, (follows.sum(), np.sort(np.abs(es - es_ref).max(0))[-16:])
    e_at_p, st_at_p = oracle_energies(c_oracle, builder, D, p, h[None])
    assert np.abs(e_at_p - es[-1])[good & (st_at_p == 0)].max() < 1e-9
    assert np.nanmin(es[-1]) < np.nanmin(es[0]) + 1e-12 and np.nanmean(es[-1]) < np.nanmean(es_ref[0]) + 1e-9


def test_optimizer_rotosolve_runs_on_the_device(c_oracle):
    """Optimizer.optimize() with settings['method'] = 'Rotosolve' (tools.py:248-270): the whole double-frequency run is
    one C call - the host objective is never evaluated per parameter."""
    from qmps_amd.ground_state import Hamiltonian, SparseFullEnergyOptimizer
    H = Hamiltonian({'ZZ': -1, 'X': 1})
    x0 = np.array([0.37, -0.81, 0.52, 0.23])
    opt = SparseFullEnergyOptimizer(H.to_matrix(), D=4, depth=2, initial_guess=x0.copy(),
                                    settings={'method': 'Rotosolve', 'maxiter': 3, 'verbose': False})

    def boom(*a, **k):
        raise AssertionError('host objective called during a device rotosolve')
    opt.batch_objective_function = boom
    opt.objective_function = boom
    res = opt.optimize()
    es_ref, p_ref, bad, _ = oracle_trajectory(c_oracle, O.shallow_cnot_unitary, 4, x0[None], H.to_matrix()[None], 3, double=True)
    assert not bad[0]
    assert np.abs(np.array(res.history) - es_ref[:, 0]).max() < 1e-8 and abs(res.fun - es_ref[-1, 0]) < 1e-8
    assert res.x is opt.initial_guess
    e_at_x, st_at_x = oracle_energies(c_oracle, O.shallow_cnot_unitary, 4, res.x[None], H.to_matrix()[None])
    assert st_at_x[0] == 0 and abs(e_at_x[0] - res.fun) < 1e-9
    assert res.fun > -4 / np.pi - 1e-9                                # variational bound E0 = -4/pi (test_ground_state.py:101)


def test_invalid_environments_leave_parameters_finite():
    """QAOA angles (0, gamma) give a product state: its environment is rank one (NOT_PD, the reference's LinAlgError
    branch, ground_state.py:153-157); ShallowCNOT at D = 4 has degenerate points at (0, 0, +-pi/2, +-pi/2), which the +-pi/2
    shifts of rotosolve do hit.  The scalar objective returns the previous value; the batched host drivers and the
    device drivers leave the affected parameter untouched - no NaN ever reaches a parameter vector."""
    from qmps_amd import rotosolve as RS
    from qmps_amd import tools as T
    from qmps_amd.ground_state import Hamiltonian, SparseFullEnergyOptimizer
    from qmps_amd.represent import ShallowQAOAStateTensor
    H = Hamiltonian({'ZZ': -1, 'X': 1})
    opt = SparseFullEnergyOptimizer(H.to_matrix(), D=2, depth=1, state_tensor=ShallowQAOAStateTensor,
                                    initial_guess=np.zeros(2), settings={'verbose': False})
    batch = opt.batch_objective_function(np.array([[0.0, 0.0], [0.0, 0.5], [0.3, -0.2]]))
    assert np.all(np.isnan(batch[:2])) and np.isfinite(batch[2])     # product states: not positive definite -> NaN in a batch
    P0 = np.array([[0.0, 0.0], [0.3, -0.2], [0.0, 1.0]])
    e1, p1 = RS.batched_rotosolve(opt.batch_objective_function, P0, N_iters=2)
    e2, p2 = RS.batched_double_rotosolve(opt.batch_objective_function, P0, N_iters=1)
    e3, p3 = RS.device_rotosolve(opt, P0, N_iters=2)
    e4, p4 = RS.device_double_rotosolve(opt, P0, N_iters=1)
    for p in (p1, p2, p3, p4):
        assert np.all(np.isfinite(p))
    assert np.abs(p1 - p3).max() < 1e-7                               # host-batched and device drivers agree, NaN rows included
    x = np.zeros(2)
    r = T.double_rotosolve(opt.objective_function, x, N_iters=1, disp=False, batch_eps=opt.batch_objective_function)
    assert np.all(np.isfinite(r.x))
    assert np.isfinite(e3[:, 1]).all() and np.isfinite(e4[:, 1]).all()
    # D = 4 ShallowCNOT start at an invalid point
    opt4 = SparseFullEnergyOptimizer(H.to_matrix(), D=4, depth=2, initial_guess=np.zeros(4), settings={'verbose': False})
    bad0 = np.array([[0.0, 0.0, np.pi / 2, np.pi / 2], [0.2, 0.1, -0.4, 0.3]])
    # (a degenerate point: the fixed point of the transfer map is not unique there.  The direct solve hands it to the power
    # method, which returns the projection of r_0 = 1/D - a valid environment - or NOT_PD / not converged -> NaN)
    e_bad = opt4.batch_objective_function(bad0)[0]
    assert np.isnan(e_bad) or e_bad > -4 / np.pi - 1e-9
    for drv in (RS.batched_rotosolve, RS.device_rotosolve):
        args = (opt4.batch_objective_function, bad0) if drv is RS.batched_rotosolve else (opt4, bad0)
        e, p = drv(*args, N_iters=1)
        assert np.all(np.isfinite(p)) and np.isfinite(e[:, 1]).all()


def test_cached_sweep_graph_is_replayed_faithfully(engine_factory):
    """The captured sweep is kept in the context: a second call of the same shape replays it.  Whatever happened to the
    context in between, the run and the state it leaves (energies, tensors of the final parameters) are the same."""
    from qmps_amd import _lib
    rng = np.random.default_rng(77)
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    for D, P in ((4, 4), (8, 6)):
        R = 40
        eng = engine_factory(D, 1024)
        eng.set_hamiltonian(h)
        P0 = rng.standard_normal((R, P))
        h1, p1 = eng.rotosolve(_lib.ANSATZ_SHALLOW_CNOT, P0, 3)
        eng.set_tensors(O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, 100)))     # something else resident in between
        eng.launch(100)
        h2, p2 = eng.rotosolve(_lib.ANSATZ_SHALLOW_CNOT, P0, 3)
        assert np.array_equal(h1, h2, equal_nan=True) and np.array_equal(p1, p2)
        E, it, st = eng.results(R)
        ok = st == 0
        assert np.abs(E[:, 0] - h2[-1])[ok].max() < 1e-13
        A = eng.tensors(R)
        for b in range(0, R, 7):
            assert np.abs(A[b] - O.unitary_to_tensor(O.shallow_cnot_unitary(D, p2[b])[None])[0]).max() < 1e-13
        # other start vectors through the same graph
        P1 = rng.standard_normal((R, P))
        h3, p3 = eng.rotosolve(_lib.ANSATZ_SHALLOW_CNOT, P1, 3)
        assert not np.array_equal(p3, p2)
        for b in np.flatnonzero(~np.isnan(h3[-1]))[::9]:
            assert abs(h3[-1][b] - O.energy_closed_form(O.unitary_to_tensor(O.shallow_cnot_unitary(D, p3[b])[None])[0], h)) < 1e-9


@pytest.mark.parametrize('double', [False, True])
def test_d2_whole_run_kernel_matches_the_step_by_step_path(double, engine_factory, monkeypatch):
    """D = 2: every sweep of every restart runs inside ONE kernel launch (single- and double-frequency); the same run through
    the step-by-step path (ansatz / energy / update kernels per parameter) gives the same trajectories."""
    from qmps_amd import _lib
    rng = np.random.default_rng(123)
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    eng = engine_factory(2, 4096)
    eng.set_hamiltonian(h)
    run = eng.double_rotosolve if double else eng.rotosolve
    for kind, P in ((_lib.ANSATZ_SHALLOW_CNOT, 8), (_lib.ANSATZ_SHALLOW_QAOA, 4), (_lib.ANSATZ_SHALLOW_FULL, 15)):
        P0 = rng.standard_normal((50, P))
        monkeypatch.delenv('QMPS_NO_FUSED_ROTO', raising=False)
        h1, p1 = run(kind, P0, 3)
        monkeypatch.setenv('QMPS_NO_FUSED_ROTO', '1')
        h2, p2 = run(kind, P0, 3)
        monkeypatch.delenv('QMPS_NO_FUSED_ROTO', raising=False)
        both = ~(np.isnan(h1).any(0) | np.isnan(h2).any(0))
        assert both.mean() > 0.8
        # trajectories agree while they stay on the same branch (a flat direction may send the two minimisers apart)
        # (round 6: the whole-run kernel keeps a cos / sin table of the restart's angles where the step-by-step path's tensor builder computes every
        # sincos inside the circuit - the same numbers, contracted differently by the compiler: the two paths differ in the last bit per evaluation
        # and a rotosolve sweep amplifies that - atan2 / Brent chains - to ~1e-7 in three sweeps; the typical restart stays at 1e-12)
        close = both & (np.abs(p1 - p2).max(1) < 1e-5)
        dh = np.abs(h1 - h2)[:, close]
        assert close.mean() > 0.8 and dh.max() < 2e-6 and np.median(dh) < 1e-10 and np.median(dh[0]) < 1e-11
        # and every final energy is the oracle's energy at the parameters that came back
        build = {_lib.ANSATZ_SHALLOW_CNOT: O.shallow_cnot_unitary, _lib.ANSATZ_SHALLOW_QAOA: O.shallow_qaoa_unitary}.get(kind)
        if build is not None:
            for b in np.flatnonzero(both)[::7]:
                Ab = O.unitary_to_tensor(build(2, p1[b])[None])[0]
                assert abs(h1[-1][b] - sum(O.energy_closed_form(Ab, h[t]) for t in range(2))) < 1e-9


@pytest.mark.parametrize('double', [False, True])
def test_d8_whole_run_kernel_matches_the_step_by_step_path(double, c_oracle, engine_factory, monkeypatch):
    """D = 8 (ShallowCNOT families): a workgroup per restart, a wave per shift, every sweep inside ONE launch - with the state
    tensor built by the wave-distributed circuit (butterflies across lanes) - against the step-by-step path (ansatz / solve +
    energy / update kernels per parameter) and against the oracle at the final parameters."""
    from qmps_amd import _lib
    rng = np.random.default_rng(888)
    h = np.stack([O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}), 0.3 * O.hamiltonian_matrix({'ZZ': -1, 'X': 1})])
    eng = engine_factory(8, 4096)
    eng.set_hamiltonian(h)
    run = eng.double_rotosolve if double else eng.rotosolve
    for kind, P, builder in ((_lib.ANSATZ_SHALLOW_CNOT, 6, O.shallow_cnot_unitary), (_lib.ANSATZ_SHALLOW_CNOT3, 9, O.shallow_cnot3_unitary),
                             (_lib.ANSATZ_SHALLOW_CNOT, 8, O.shallow_cnot_unitary)):      # (fewer than three layers: no full-rank environment at D = 8)
        R = 180 if double else 40          # (six shifts: the whole-run kernel serves 1 024 < 6 R <= 1 536 evaluations per update)
        P0 = rng.standard_normal((R, P))
        monkeypatch.delenv('QMPS_NO_FUSED_ROTO', raising=False)
        h1, p1 = run(kind, P0, 3)
        E1 = eng.results(R)[0].sum(1)                    # the resident state the call leaves: energies of the final vectors
        monkeypatch.setenv('QMPS_NO_FUSED_ROTO', '1')
        h2, p2 = run(kind, P0, 3)
        monkeypatch.delenv('QMPS_NO_FUSED_ROTO', raising=False)
        both = ~(np.isnan(h1).any(0) | np.isnan(h2).any(0))
        assert both.mean() > 0.8
        close = both & (np.abs(wrap(p1 - p2)).max(1) < 1e-6)
        assert close.mean() > 0.8, close.mean()
        # the two paths build the tensor differently (wave-distributed butterflies / one lane per column): rounding-level
        # differences, amplified from sweep to sweep like the evaluator noise of the oracle-driven tests above
        # (first sweep: 1e-10 for all but the odd restart - one update in a thousand - where the bounded search, fed coefficients that
        # differ at 1e-12 between the two paths, takes another decision: its angle then moves by up to scipy's tolerance, 1e-5)
        d0 = np.abs(h1 - h2)[0, both]
        assert (d0 < 1e-10).mean() > 0.98 and d0.max() < (5e-6 if double else 1e-8) and (np.abs(h1 - h2)[:, close].max(0) < 1e-8).mean() > 0.9
        assert np.abs(E1 - h1[-1])[both].max() < 1e-10
        e_at_p, st_at_p = oracle_energies(c_oracle, builder, 8, p1, h)
        ok = both & (st_at_p == 0)
        assert np.abs(e_at_p - h1[-1])[ok].max() < 1e-9


def test_config1_family_with_a_landscape_reaches_the_d2_optimum(c_oracle, engine_factory):
    """BASELINE.json configs[1] as written is a flat landscape (tests/test_refshim_cpu.py); the same configuration with the D = 2
    universal gate (ShallowFullStateTensor, 15 angles) and the double-frequency rule of Optimizer('Rotosolve') - what bench.py reports
    as `config1_family_rotosolve_D2_shallowfull_double` - optimises: after 24 sweeps the best of 64 restarts is below -1.26 and the
    mean below -1.2 (the reference quotes D2_gse = -1.269909412573 as the D = 2 optimum, scripts/noisy_optimization.py:93; the exact
    ground-state energy -4/pi bounds everything from below), and the ORACLE's energy at the device's final parameters is the device's."""
    rng = np.random.default_rng(15)
    R, sweeps = 64, 24
    P0 = rng.standard_normal((R, 15))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    eng = engine_factory(2, 4096)
    eng.set_hamiltonian(h)
    es, p = eng.double_rotosolve(2, P0, sweeps)
    assert np.isfinite(es).all()
    # (dense eigen-solve of the transfer matrix: optimised states sit close to the critical point, where the oracle's plain power
    # iteration takes > 10^5 steps to 1e-15)
    e_at_p = np.array([O.energy_closed_form(O.unitary_to_tensor(O.shallow_full_unitary(q)), h) for q in p])
    # (restarts that settle at E = -1.25 sit next to a degenerate transfer spectrum - eigenvalues 1 and 1 - 1e-6: the fixed point, and with
    # it the energy, is defined to ~1e-6 only, whoever evaluates it; see test_double_frequency_trajectory_follows_the_oracle)
    dev = np.abs(e_at_p - es[-1])
    assert (dev < 1e-9).mean() > 0.6 and dev.max() < 1e-5, np.sort(dev)[-5:]
    assert es[-1].min() < -1.26 and es[-1].mean() < -1.2 and es[-1].min() > -4 / np.pi
    assert es[-1].mean() < es[0].mean()
