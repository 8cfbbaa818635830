"""GPU: BASELINE.json configs[4] as a workload - device-backed TIME EVOLUTION at D = 2, 4, 16 (qmps/new_time_evolve.py:252-302,
scripts/loschmidt.py:335-383, scripts/rotosolve.py:270-294) replayed on the host with the ORACLE (numpy dense eig of the
D^2 x D^2 mixed transfer matrix, oracle.overlap_eta) as evaluator; plus the pieces the drivers are made of: reference states
built from parameters on the device, trajectory-major candidate groups, resident warm starts, solver statistics.
Tolerance: 1e-8 on every recorded objective value -sqrt|eta| (eta itself 1e-10, as in test_overlap_gpu.py)."""
import numpy as np
import pytest
from scipy.linalg import expm

import evolve_replay as ER
from oracle import qmps_oracle as O
from qmps_amd import _lib as L
from qmps_amd import new_time_evolve as NT
from qmps_amd import represent as R

pytestmark = pytest.mark.gpu

H_TFIM = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
F_TOL = 1e-8


def WW_of(dt):
    return expm(-1j * dt * H_TFIM)


@pytest.mark.parametrize('D,kind,P,G', [(2, 2, 15, 5), (4, 0, 4, 4), (8, 0, 6, 3), (16, 0, 8, 3)])
def test_grouped_candidates_against_device_built_references(D, kind, P, G, engine_factory):
    """Reference tensors from parameters on the device, candidate b against reference b // G, objective -sqrt|eta|
    computed by the kernel, statistics - all against the oracle."""
    rng = np.random.default_rng(40 + D)
    eng = engine_factory(D, 4096)
    T = 5
    WW = WW_of(0.05)
    ref = rng.standard_normal((T, P))
    cand = (ref[:, None, :] + 0.05 * rng.standard_normal((T, G, P))).reshape(T * G, P)
    eng.overlap_set_refs_params(kind, ref, WW)
    eng.overlap_set_group(G)
    eng.set_ansatz_params(kind, cand)
    eng.overlap_stats(reset=True)
    eng.overlap_launch(T * G, tol=1e-13, want_r=True)
    eta, rounds, st, r = eng.overlap_results(T * G, want_r=True)
    f = eng.overlap_objective(T * G)
    assert np.all(st == 0)
    for b in range(T * G):
        A = ER.tensor(kind, D, ref[b // G])
        ref_eta, r_ref = O.overlap_eta(A, ER.tensor(kind, D, cand[b]), WW)
        assert abs(eta[b] - ref_eta) < 1e-10, (b, eta[b], ref_eta)
        assert abs(f[b] + np.sqrt(abs(ref_eta))) < 1e-10
        assert abs(abs(np.vdot(r_ref, r[b])) - 1.0) < 1e-8
    s = eng.overlap_stats()
    assert s['evaluations'] == T * G and s['rounds_sum'] == int(rounds.sum()) and s['rounds_max'] == int(rounds.max()) and s['not_converged'] == 0
    # a window inside the groups: candidates [2 G, 4 G) against references 2 and 3
    eng.set_window(2 * G)
    eng.overlap_launch(2 * G, tol=1e-13)
    eta_w, _, st_w = eng.overlap_results(2 * G)
    assert np.all(st_w == 0) and np.abs(eta_w - eta[2 * G:4 * G]).max() < 1e-12
    with pytest.raises(L.QmpsError):
        eng.set_window(1)
        eng.overlap_launch(G, tol=1e-13)          # a window must start at a multiple of the group
    eng.set_window(0)
    eng.overlap_set_group(0)
    with pytest.raises(L.QmpsError):
        eng.overlap_launch(T * G, tol=1e-13)      # 5 references, 15+ candidates, no group: refused, not mis-addressed


@pytest.mark.parametrize('D,P', [(8, 6), (16, 8)])
def test_warm_start_from_resident_fixed_points(D, P, engine_factory):
    """QMPS_OVERLAP_WARM: the same batch shape re-evaluated after the candidates moved a little starts from the fixed
    points the previous launch left in the slots: same eigenvalues (oracle), fewer power steps; unmoved candidates are
    accepted by their first step; a launch without resident fixed points is refused."""
    rng = np.random.default_rng(50 + D)
    eng = engine_factory(D, 4096)
    T, G = 6, 4
    WW = WW_of(0.05)
    ref = rng.standard_normal((T, P))
    cand = (ref[:, None, :] + 0.03 * rng.standard_normal((T, G, P))).reshape(T * G, P)
    eng.overlap_set_refs_params(0, ref, WW)
    eng.overlap_set_group(G)
    eng.set_ansatz_params(0, cand)
    from qmps_amd import EnergyEngine
    with EnergyEngine(D, 64) as fresh:                        # a context that has never kept fixed points refuses a warm launch
        fresh.overlap_set_refs_params(0, ref, WW)
        fresh.overlap_set_group(G)
        fresh.set_ansatz_params(0, cand)
        fresh.overlap_launch(T * G, tol=1e-12)
        with pytest.raises(L.QmpsError):
            fresh.overlap_launch(T * G, tol=1e-12, warm=True)
    eng.overlap_launch(T * G, tol=1e-12, want_r=True)
    eta0, rounds0, st0 = eng.overlap_results(T * G)
    assert np.all(st0 == 0)
    eng.overlap_launch(T * G, tol=1e-12, warm=True)           # nothing moved: one step each
    eta1, rounds1, st1 = eng.overlap_results(T * G)
    assert np.all(st1 == 0) and np.all(rounds1 <= 2) and np.abs(eta1 - eta0).max() < 1e-11
    moved = cand + 1e-6 * rng.standard_normal(cand.shape)     # finite-difference neighbours of the same iterates
    eng.set_ansatz_params(0, moved)
    eng.overlap_launch(T * G, tol=1e-12, warm=True)
    eta2, rounds2, st2 = eng.overlap_results(T * G)
    assert np.all(st2 == 0)
    assert rounds2.mean() < 0.75 * rounds0.mean(), (rounds2.mean(), rounds0.mean())
    for b in range(0, T * G, 5):
        ref_eta, _ = O.overlap_eta(ER.tensor(0, D, ref[b // G]), ER.tensor(0, D, moved[b]), WW)
        assert abs(eta2[b] - ref_eta) < 1e-10


@pytest.mark.parametrize('D,P', [(4, 4), (8, 6), (16, 8)])
def test_two_sided_gradient_vs_oracle_central_differences(D, P, engine_factory):
    """qmps_overlap_gradient: objective and central-difference gradient of T iterates from ONE right and ONE left eigen-solve
    each (the neighbours by eta' = <y, T'(r)>/<y, r>) against the oracle's central differences of dense eigen-solves."""
    rng = np.random.default_rng(80 + D)
    eng = engine_factory(D, 4096)
    T, h = 5, 1e-6
    WW = WW_of(0.05)
    ref = rng.standard_normal((T, P))
    X = ref + 0.03 * rng.standard_normal((T, P))
    eng.overlap_set_refs_params(0, ref, WW)
    eng.overlap_stats(reset=True)
    f, g, st = eng.overlap_gradient(0, X, h=h, tol=1e-13)
    assert np.all(st == 0)
    cold = eng.overlap_stats(reset=True)
    assert cold['evaluations'] == 2 * T                       # two solves per iterate, not 2 P + 1
    for t in range(T):
        A = ER.tensor(0, D, ref[t])
        assert abs(f[t] - ER.objective(0, D, A, X[t], WW)) < 1e-10
        for k in range(P):
            e = np.zeros(P)
            e[k] = h
            g_ref = (ER.objective(0, D, A, X[t] + e, WW) - ER.objective(0, D, A, X[t] - e, WW)) / (2 * h)
            assert abs(g[t, k] - g_ref) < 2e-7, (t, k, g[t, k], g_ref)
    # warm: the iterates moved by a BFGS step's worth; same answers, fewer power steps on both sides
    X2 = X + 1e-3 * rng.standard_normal(X.shape)
    f2, g2, st2 = eng.overlap_gradient(0, X2, h=h, tol=1e-13, warm=True)
    warm = eng.overlap_stats()
    assert np.all(st2 == 0)
    f3, g3, st3 = eng.overlap_gradient(0, X2, h=h, tol=1e-13)
    assert np.abs(f2 - f3).max() < 1e-11 and np.abs(g2 - g3).max() < 1e-6
    if D >= 8:
        assert warm['rounds_sum'] < cold['rounds_sum']
    with pytest.raises(L.QmpsError):
        eng.overlap_gradient(0, X2[:3], h=h, warm=True)        # the resident fixed points belong to 5 trajectories


# (D, ansatz kind, parameters, trajectories, steps, sweeps, shifts per parameter, seed)
ROTO_CASES = [(2, 2, 15, 3, 3, 2, 3, 11),      # the reference's own case: ShallowFullStateTensor(2, .), new_time_evolve.py:186-187
              (2, 0, 8, 3, 3, 2, 6, 12),       # scripts/loschmidt.py:203-207: ShallowCNOTStateTensor(2, .) with 8 angles, double frequency
              (4, 0, 4, 3, 3, 2, 3, 13),
              (4, 0, 4, 2, 3, 1, 6, 14),
              (16, 0, 8, 2, 2, 1, 3, 903)]     # configs[4]: D = 16, depth 4 (the +-pi/2 candidates are FAR from the reference
                                               # state: crowded rings of eigenvalues, up to 10^4 power steps on this seed - a few
                                               # hundred map applications through the Krylov fall-back)


@pytest.mark.parametrize('D,kind,P,T,n_steps,n_sweeps,nsh,seed', ROTO_CASES)
def test_device_time_evolution_by_rotosolve_vs_oracle_replay(D, kind, P, T, n_steps, n_sweeps, nsh, seed, engine_factory):
    """qmps_evolve_rotosolve (every step, sweep, parameter and trajectory inside one C call) against the host replay that
    evaluates every candidate with the oracle's dense eigen-solve.
    Six shifts: the machinery is held to 1e-8 with QMPS_ROTO_GLOBAL_ARGMIN on both sides (a minimiser resolved to 1e-15); with the
    DEFAULT rule - the reference's bounded Brent search, xatol 1e-5 - device and replay take the same decisions but a search that
    sees function values only cannot place its minimiser better than ~sqrt(eps) (1e-8 .. 1e-7 here: sin / sincos of two maths
    libraries differ in the last bit), and -sqrt|eta| is NOT stationary at the minimiser of the fitted curve, so the recorded
    objectives agree to first order in that: 5e-6 (scipy's own tolerance allows 1e-5 in the angle)."""
    rng = np.random.default_rng(seed)
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    eng = engine_factory(D, 4096)
    if nsh == 6:
        _, _, fh_d = eng.evolve_rotosolve(kind, X0, WW, n_steps=n_steps, n_sweeps=n_sweeps, double_frequency=True, max_rounds=60 if D <= 4 else 5000, tol=1e-12)
        _, fh_r = ER.replay_rotosolve(kind, D, X0, WW, n_steps, n_sweeps, nsh)
        assert np.abs(fh_d - fh_r).max() < 5e-6, np.abs(fh_d - fh_r).max()
    eng.overlap_stats(reset=True)
    Xf, ph, fh = eng.evolve_rotosolve(kind, X0, WW, n_steps=n_steps, n_sweeps=n_sweeps, double_frequency=nsh == 6,
                                      max_rounds=60 if D <= 4 else 5000, tol=1e-12, rule=L.ROTO_GLOBAL_ARGMIN)
    stats = eng.overlap_stats()
    assert stats['not_converged'] == 0, stats
    assert stats['evaluations'] == n_steps * n_sweeps * (P * nsh + 1) * T
    ph_ref, fh_ref = ER.replay_rotosolve(kind, D, X0, WW, n_steps, n_sweeps, nsh, global_argmin=True)
    assert np.abs(fh - fh_ref).max() < F_TOL, np.abs(fh - fh_ref).max()
    # the parameters themselves: the recorded objective IS the oracle's objective of the device's parameter vectors against
    # the device's previous ones (a gauge angle the objective does not depend on may differ from the replay: its three samples
    # are equal to rounding and atan2(~0, ~0) is anybody's guess - so the vectors are compared through what they describe)
    prev = X0
    for step in range(n_steps):
        for t in range(T):
            f_t = ER.objective(kind, D, ER.tensor(kind, D, prev[t]), ph[step, t], WW)
            assert abs(f_t - fh[step, -1, t]) < F_TOL
            o = abs(O.overlap_eta(ER.tensor(kind, D, ph[step, t]), ER.tensor(kind, D, ph_ref[step, t]), np.eye(4))[0])
            assert abs(o - 1.0) < 1e-6, (step, t, o)          # same physical state as the replay's
        prev = ph[step]
    assert np.array_equal(Xf, ph[-1])
    assert np.all(fh < 0) and np.all(fh >= -1 - 1e-12)
    # (no claim that the sweeps IMPROVE the objective: -sqrt|eta| with the exact environment is not a sinusoid of a gate angle -
    # the reference's rotosolve time evolution, scripts/rotosolve.py:270-294, fits a variational environment for that reason -
    # so this driver is checked for what it computes; the optimiser that does the physics is method='BFGS' below)


def test_rotosolve_evolution_d16_above_the_queue_threshold(engine_factory):
    """at D = 16 a sweep of more than 2 048 candidates (nsh T > 2 048) runs the four-wave kernel with its
    work queue INSIDE the captured sweep graph - the queue counter must exist before the capture starts (it used to be
    hipMalloc'ed lazily at the first such launch, which a stream capture rejects).  700 trajectories x 3 shifts; the recorded
    objectives are the oracle's at the device's parameters."""
    rng = np.random.default_rng(77)
    D, kind, P, T = 16, 0, 8, 700
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    eng = engine_factory(D, 3 * T)
    eng.overlap_stats(reset=True)
    Xf, ph, fh = eng.evolve_rotosolve(kind, X0, WW, n_steps=1, n_sweeps=1, double_frequency=False, max_rounds=5000, tol=1e-12)
    stats = eng.overlap_stats()
    assert stats['not_converged'] == 0 and stats['evaluations'] == (P * 3 + 1) * T, stats
    for t in (0, 349, 699):
        f_t = ER.objective(kind, D, ER.tensor(kind, D, X0[t]), ph[0, t], WW)
        assert abs(f_t - fh[0, -1, t]) < F_TOL, (t, f_t, fh[0, -1, t])


@pytest.mark.parametrize('D,P,T,iters', [(2, 8, 4, 12), (4, 4, 4, 12), (16, 8, 3, 8)])
def test_lockstep_bfgs_time_evolution(D, P, T, iters):
    """`evolve(..., method='BFGS')`: batched central-difference gradients and backtracking ladders of all trajectories on
    the device (warm-started at D = 16) against the SAME lock-step driver with the oracle as evaluator."""
    from qmps_amd.tools import batched_bfgs
    rng = np.random.default_rng(70 + D)
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    n_steps = 3
    # D = 2: the neighbours are eigen-solved one by one; D = 4: both gradient routes; D = 16: two-sided + two-stage ladder
    # the numpy loop: its per-iteration history is compared below (the default, the one-call native driver: next test)
    opts = {'maxiter': iters, 'native': False, 'speculative': D == 16}   # D = 16: objective + gradient at the full step first, ladder on rejection
    H, info = NT.evolve(X0, WW, n_steps, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-13,
                        options=opts, return_info=True)
    if D == 4:
        # carried inverse Hessians: the same minima (objective to 1e-8), fewer iterations from the second time step on
        H_c, info_c = NT.evolve(X0, WW, n_steps, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-13,
                                options={'maxiter': 40, 'carry_hessian': True, 'native': False, 'speculative': False}, return_info=True)
        H_i, info_i = NT.evolve(X0, WW, n_steps, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-13,
                                options={'maxiter': 40, 'native': False, 'speculative': False}, return_info=True)
        for a, b in zip(info_c['fun'], info_i['fun']):
            assert np.abs(a[-1] - b[-1]).max() < 1e-7
        assert sum(info_c['nit'][1:]) < sum(info_i['nit'][1:])
        H_fd, info_fd = NT.evolve(X0, WW, n_steps, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-13,
                                  options={'maxiter': iters, 'gradient': 'fd', 'native': False, 'speculative': False}, return_info=True)
        for a, b in zip(info['fun'], info_fd['fun']):
            assert a.shape == b.shape and np.abs(a - b).max() < F_TOL
    assert H.shape == (n_steps + 1, T, P)
    X = X0.copy()
    for step in range(n_steps):
        A = [ER.tensor(0, D, X[t]) for t in range(T)]

        def fb(G):
            # D = 16: ARPACK in operator form, the reference's own route (xmps Map -> scipy eigs) - the dense eigen-solve of
            # hundreds of 256 x 256 matrices takes minutes; every 7th candidate is cross-checked against it
            def f(C):
                v = np.array([ER.objective(0, D, A[b // G], C[b], WW, arpack=D >= 16) for b in range(len(C))])
                if D >= 16:
                    for b in range(0, len(C), 7):
                        assert abs(v[b] - ER.objective(0, D, A[b // G], C[b], WW)) < 1e-10
                return v
            return f
        if D == 16:                                   # the same driver options on the host side
            G = 2 * P + 1
            fd = fb(G)

            def vg_host(Z, fd=fd, G=G):
                from qmps_amd.tools import batched_fd_gradient
                return batched_fd_gradient(fd, Z, 1e-6)
            res = batched_bfgs(None, fb(7), X, maxiter=iters, value_and_grad=vg_host, speculative=True)
        else:
            res = batched_bfgs(fb(2 * P + 1), fb(8), X, maxiter=iters)
        dev = info['fun'][step]
        assert dev.shape == res['history'].shape, (dev.shape, res['history'].shape)
        assert np.abs(dev - res['history']).max() < F_TOL, np.abs(dev - res['history']).max()
        # continue from the DEVICE's iterate: the comparison of the next step is then about that step only
        assert np.abs(H[step + 1] - res['x']).max() < 1e-4
        X = H[step + 1]
        # the optimiser did its job: the projected state is closer to W|A A> than the unevolved one
        assert np.all(dev[-1] <= dev[0] + 1e-12)
        assert dev[-1].mean() < -0.999
    if D == 16:
        s = info['solver']['gradient_batches']
        assert s['not_converged'] == 0
        assert s['evaluations'] >= 2 * T * sum(n + 1 for n in info['nit'])      # two solves per iterate and iteration (+ re-evaluations after a rejected full step)


def scipy_bfgs_step(kind, D, x_start, WW, **opts):
    """One time step the reference's way (qmps/new_time_evolve.py:284, scripts/loschmidt.py:367-375): scipy.optimize.minimize(obj,
    params, (A_, WW)) - BFGS with forward-difference gradients and a Wolfe line search, started from the previous parameters - on
    the ORACLE's objective -sqrt|eta| (dense eigen-solve)."""
    from scipy.optimize import minimize
    A = ER.tensor(kind, D, x_start)
    return minimize(lambda p: ER.objective(kind, D, A, p, WW), np.array(x_start, dtype=float), method='BFGS', options=opts or None)


@pytest.mark.parametrize('kind,P,driver', [(2, 15, 'device'), (2, 15, 'host'), (0, 8, 'device')])
def test_lockstep_bfgs_reaches_scipys_minima_at_the_reference_case(kind, P, driver, engine_factory):
    """how far are the lock-step driver's per-step minima (central differences / eigen-solved neighbours,
    Armijo ladder) from scipy's BFGS (forward differences, Wolfe search) on the same objective?  The reference's own case: D = 2,
    ShallowFullStateTensor with 15 angles (new_time_evolve.py:186-187, 276-292), 5 time steps, 8 trajectories; every time step of
    every trajectory is ALSO minimised by scipy from the same starting point against the same reference state.
    Bar: final objective within 1e-6 of scipy's (either way: whoever stops lower), and the two minimisers describe the same
    physical state: |<A_dev|A_scipy>| per site >= 1 - 1e-6."""
    rng = np.random.default_rng(2024 + kind)
    T, n_steps = 8, 5
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    eng = engine_factory(2, T * (2 * P + 1))
    run = eng.evolve_bfgs_device if driver == 'device' else eng.evolve_bfgs
    dev = run(kind, X0, WW, n_steps=n_steps, maxiter=200, tol=1e-13)
    prev = X0
    worst_f, worst_fid, scipy_lower, dev_lower = 0.0, 1.0, 0, 0
    for step in range(n_steps):
        for t in range(T):
            res = scipy_bfgs_step(kind, 2, prev[t], WW)
            f_dev = dev['fun'][step, t]
            # the recorded value IS the oracle's objective at the device's parameters
            assert abs(ER.objective(kind, 2, ER.tensor(kind, 2, prev[t]), dev['params_hist'][step, t], WW) - f_dev) < F_TOL
            worst_f = max(worst_f, abs(f_dev - res.fun))
            scipy_lower += res.fun < f_dev - 1e-9
            dev_lower += f_dev < res.fun - 1e-9
            fid = abs(O.overlap_eta(ER.tensor(kind, 2, dev['params_hist'][step, t]), ER.tensor(kind, 2, res.x), np.eye(4))[0])
            worst_fid = min(worst_fid, fid)
        prev = dev['params_hist'][step]
    print(f'lock-step ({driver}) vs scipy BFGS, kind {kind}: max |f_dev - f_scipy| = {worst_f:.2e}, min fidelity {worst_fid:.9f}, '
          f'scipy lower in {scipy_lower}, device lower in {dev_lower} of {n_steps * T} minimisations')
    assert worst_f < 1e-6 and worst_fid > 1 - 1e-6
    assert dev['fun'].mean() < -0.999


def test_config4_full_size_256_trajectories():
    """BASELINE.json configs[4] at its stated size: TFIM quench, D = 16, depth 4 (8 angles), 256 trajectories (the
    evolution was only ever tested with T <= 32).  Two time steps of the default driver; the recorded objectives are the ORACLE's
    (ARPACK in operator form - the reference's route - and, for two of them, the dense eigen-solve) at the device's parameters, every
    solve converged, and a trajectory evolved inside the batch of 256 ends where it ends when evolved in a batch of 5."""
    rng = np.random.default_rng(2560)
    D, P, T, n_steps = 16, 8, 256, 2
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    H, info = NT.evolve(X0, WW, n_steps, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-12,
                        options={'maxiter': 30}, return_info=True)
    assert H.shape == (n_steps + 1, T, P) and np.all(np.isfinite(H))
    assert info['solver']['gradient_batches']['not_converged'] == 0
    f_end = np.array([f[-1] for f in info['fun']])          # (n_steps, T): objective at the end of each time step
    assert f_end.shape == (n_steps, T) and f_end.mean() < -0.999 and f_end.max() < -0.99
    pick = [0, 1, 97, 255]
    for step in range(n_steps):
        for t in pick:
            A = ER.tensor(0, D, H[step][t])
            f_t = ER.objective(0, D, A, H[step + 1][t], WW, arpack=True)
            assert abs(f_t - f_end[step, t]) < F_TOL, (step, t, f_t, f_end[step, t])
    for t in pick[:2]:
        assert abs(ER.objective(0, D, ER.tensor(0, D, H[1][t]), H[2][t], WW) - f_end[1, t]) < F_TOL
    Hs, infos = NT.evolve(X0[[0, 1, 97, 255, 13]], WW, n_steps, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-12,
                          options={'maxiter': 30}, return_info=True)
    fs_end = np.array([f[-1] for f in infos['fun']])
    assert np.abs(fs_end[:, :4] - f_end[:, pick]).max() < 1e-8


@pytest.mark.parametrize('D,P,carry', [(2, 8, True), (4, 4, False), (8, 6, True), (16, 8, True), (16, 8, False)])
def test_native_bfgs_driver_takes_the_decisions_of_the_numpy_loop(D, P, carry):
    """qmps_evolve_bfgs - the lock-step BFGS time step with its host arithmetic in C++ inside the library, one C call for the
    whole evolution - against tools.batched_bfgs(speculative=True) driving the same device batches from numpy: same iteration
    counts, same objectives, same parameters (the two sum their dot products in different orders: rounding-level differences)."""
    rng = np.random.default_rng(300 + D)
    T, n_steps = 6, 3
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    out = {}
    for native in (False, True):
        H, info = NT.evolve(X0, WW, n_steps, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-13,
                            options={'maxiter': 25, 'speculative': True, 'carry_hessian': carry, 'native': native, 'device_driver': False,
                                     'adaptive_gradient': False}, return_info=True)      # (the numpy loop solves every batch to max(tol, 1e-8))
        out[native] = (H, info)
    (Hn, In), (Hp, Ip) = out[True], out[False]
    assert Hn.shape == Hp.shape == (n_steps + 1, T, P)
    # (D = 2 with carried inverse Hessians: eight angles on two qubits leave flat directions along which the two drivers' rounding-level
    # differences travel - a line search may end an iteration earlier or later; everywhere else the counts are identical)
    if D == 2 and carry:
        assert np.abs(np.array(In['nit']) - np.array(Ip['nit'])).max() <= 2
    else:
        assert list(In['nit']) == list(Ip['nit'])
    for a, b in zip(In['fun'], Ip['fun']):       # (native history: objective at the start and at the end of the time step)
        # (the start of a step is NOT a minimum: first order in the ~1e-6 by which the two drivers' parameters differ)
        assert a.shape == (2, T) and np.abs(a[-1] - b[-1]).max() < (1e-8 if (D == 2 and carry) else 1e-9) and np.abs(a[0] - b[0]).max() < 1e-7
    # (D = 2, depth 4: eight angles on two qubits - flat directions along which rounding-level differences of the two drivers'
    # dot products travel freely; the objectives above agree to 1e-9)
    assert np.abs(Hn - Hp).max() < (1e-6 if D >= 4 else 2e-3)
    assert all(f[-1].mean() < -0.999 for f in In['fun'])
    # one C call for three time steps = three calls of one step each (resident fixed points and inverse Hessians carried over)
    # (device_driver=False: this test is about the host loop in C++; the device-resident optimiser of D = 2, 4 has its own tests above)
    ev = NT.LockstepEvolver(D, T, P, R.ShallowCNOTStateTensor, None, 1e-13, 25, 1e-5, 1e-6, speculative=True, carry_hessian=carry, device_driver=False, adaptive_gradient=False)
    ev2 = NT.LockstepEvolver(D, T, P, R.ShallowCNOTStateTensor, None, 1e-13, 25, 1e-5, 1e-6, speculative=True, carry_hessian=carry, device_driver=False, adaptive_gradient=False)
    try:
        whole = ev.steps(X0, WW, n_steps)
        X = X0
        for k in range(n_steps):
            r = ev2.step(X, WW)
            X = r['x']
            assert np.array_equal(r['fun'], whole['fun'][k]) and np.array_equal(X, whole['params_hist'][k])
        assert np.array_equal(Hn[1:], whole['params_hist'])
        assert whole['gradient_batches'] >= n_steps and whole['gradient_ms'] > 0
    finally:
        ev.close()
        ev2.close()


@pytest.mark.parametrize('adaptive', [False, True])
@pytest.mark.parametrize('D,P,T,carry,start', [(16, 8, 256, True, 'near'), (16, 8, 9, False, 'near'), (8, 6, 40, True, 'near'), (8, 6, 12, False, 'far'),
                                                (16, 8, 5, True, 'far')])
def test_device_algebra_takes_the_decisions_of_the_host_loop(D, P, T, carry, start, adaptive, engine_factory, monkeypatch):
    """Round 5: at D = 8, 16 the algebra between two gradient evaluations (directions, Armijo test of the full step, rank-two update
    of H^-1, masks, next candidates) runs in kernels on device-resident state (qmps_evolve_lockstep.hip), the host only enqueues
    chains of iterations and finishes the rare iteration with a rejected full step.  QMPS_EVOLVE_HOST_ALGEBRA selects the round-4
    host loop.  Same evaluations, same expressions in the same order without contraction: the two must agree on every number -
    iteration counts, objectives, parameters, inverse Hessians - to the last bit, also across a continued call, with and without
    rejected steps ('far': the second time step starts from a perturbed point, so that full steps get rejected), with the fixed and
    with the adaptive tolerance of the gradient solves (QMPS_BFGS_ADAPTIVE_GRADIENT: the same rule in both implementations)."""
    rng = np.random.default_rng(5000 + D + T)
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    n_steps = 3
    out = {}
    # 'device': blind chains of iterations (no counters, no events: what the timed region of bench.py runs); 'device_counted': one
    # iteration per chain with HIP events around every evaluation (the instrumented pass); 'host': the round-4 loop
    for name in ('device', 'device_counted', 'host'):
        if name == 'host':
            monkeypatch.setenv('QMPS_EVOLVE_HOST_ALGEBRA', '1')
        else:
            monkeypatch.delenv('QMPS_EVOLVE_HOST_ALGEBRA', raising=False)
        eng = engine_factory(D, T * (2 * P + 1))
        cnt = name != 'device'
        a = eng.evolve_bfgs(0, X0, WW, n_steps=1, maxiter=30, tol=1e-12, carry_hessian=carry, counters=cnt, adaptive_gradient=adaptive)
        X1 = a['x'] + (0.3 * np.random.default_rng(7).standard_normal(a['x'].shape) if start == 'far' else 0.0)
        b = eng.evolve_bfgs(0, X1, WW, n_steps=n_steps, maxiter=30, tol=1e-12, carry_hessian=carry, hess_inv=a['hess_inv'] if carry else None,
                            warm=(start == 'near'), counters=cnt, adaptive_gradient=adaptive)
        out[name] = (a, b)
    monkeypatch.delenv('QMPS_EVOLVE_HOST_ALGEBRA', raising=False)
    for k in (0, 1):
        hs = out['host'][k]
        for name in ('device', 'device_counted'):
            dv = out[name][k]
            assert np.array_equal(dv['nit'], hs['nit']), (name, dv['nit'], hs['nit'])
            assert np.array_equal(dv['fun'], hs['fun']) and np.array_equal(dv['fun_start'], hs['fun_start']), name
            assert np.array_equal(dv['x'], hs['x']) and np.array_equal(dv['params_hist'], hs['params_hist']), name
            assert np.array_equal(dv['hess_inv'], hs['hess_inv']), name
        dv = out['device_counted'][k]
        assert dv['gradient_batches'] == hs['gradient_batches'] and dv['ladder_batches'] == hs['ladder_batches'] and dv['nfev'] == hs['nfev']
        assert dv['gradient_ms'] > 0
    if start == 'far':
        assert out['device_counted'][1]['ladder_batches'] > 0          # the rare path was exercised
    # and the numbers are the oracle's: objective at the end of the last step against the previous parameters
    dv = out['device'][1]
    prev = dv['params_hist'][-2] if n_steps > 1 else None
    for t in (0, T - 1):
        f_t = ER.objective(0, D, ER.tensor(0, D, prev[t]), dv['params_hist'][-1, t], WW, arpack=D >= 16)
        assert abs(f_t - dv['fun'][-1, t]) < F_TOL


@pytest.mark.parametrize('D,P,T', [(16, 8, 48), (8, 6, 40)])
def test_adaptive_gradient_tolerance_reaches_the_same_minima(D, P, T, engine_factory):
    """QMPS_BFGS_ADAPTIVE_GRADIENT: the eigen-solves behind a trajectory's gradient stop at clamp(1e-3 max|g|, 1e-8, 1e-6) instead of 1e-8.
    An inexact-gradient rule must not move the minima: per time step the final objectives of the two runs agree to 1e-8 (the recorded
    objective comes from the two-sided quotient either way and is the ORACLE's at the device's parameters), the lock-step iteration
    counts differ by at most one, the states agree per site to 1e-7, and the adaptive run spends fewer power steps."""
    rng = np.random.default_rng(7100 + D)
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    eng = engine_factory(D, T * (2 * P + 1))
    out = {}
    for adaptive in (False, True):
        eng.overlap_stats(reset=True)
        r = eng.evolve_bfgs(0, X0, WW, n_steps=4, maxiter=40, tol=1e-12, carry_hessian=True, adaptive_gradient=adaptive)
        out[adaptive] = (r, eng.overlap_stats())
    (fx, sf), (ad, sa) = out[False], out[True]
    assert np.abs(ad['fun'] - fx['fun']).max() < 1e-8, np.abs(ad['fun'] - fx['fun']).max()
    assert np.abs(ad['nit'].astype(int) - fx['nit'].astype(int)).max() <= 1, (ad['nit'], fx['nit'])
    assert sa['not_converged'] == 0 and sf['not_converged'] == 0
    assert sa['rounds_sum'] < 0.9 * sf['rounds_sum'], (sa['rounds_sum'], sf['rounds_sum'])
    prev = ad['params_hist'][-2]
    for t in (0, T // 2, T - 1):
        f_t = ER.objective(0, D, ER.tensor(0, D, prev[t]), ad['params_hist'][-1, t], WW, arpack=D >= 16)
        assert abs(f_t - ad['fun'][-1, t]) < F_TOL
        o = abs(O.overlap_eta_arpack(ER.tensor(0, D, ad['x'][t]), ER.tensor(0, D, fx['x'][t]), np.eye(4))[0])
        assert abs(o - 1.0) < 1e-7, (t, o)


@pytest.mark.parametrize('D,P', [(8, 6), (16, 8)])
def test_two_sided_objective_is_second_order_in_the_residuals(D, P, engine_factory):
    """QMPS_OVERLAP_TWO_SIDED_F: with the eigen-solves stopped at a residual of 1e-8 the objective from <y, T(r)>/<y, r> is as
    accurate as a solve to 1e-14 (error = product of the residuals), the one-sided estimate of the right solve is not, and the
    gradient inherits the residual to first order - what qmps_evolve_bfgs relies on for its gradient batches."""
    from qmps_amd import _lib
    rng = np.random.default_rng(40 + D)
    T = 24
    X = rng.standard_normal((T, P))
    Z = X + 0.02 * rng.standard_normal((T, P))
    eng = engine_factory(D, T * (2 * P + 1))
    eng.overlap_set_refs_params(_lib.ANSATZ_SHALLOW_CNOT, X, WW_of(0.05))
    f0, g0, st0 = eng.overlap_gradient(_lib.ANSATZ_SHALLOW_CNOT, Z, tol=1e-14)
    eng.overlap_stats(reset=True)
    f1, g1, st1 = eng.overlap_gradient(_lib.ANSATZ_SHALLOW_CNOT, Z, tol=1e-8)
    loose = eng.overlap_stats(reset=True)
    f2, g2, st2 = eng.overlap_gradient(_lib.ANSATZ_SHALLOW_CNOT, Z, tol=1e-8, two_sided_f=True)
    assert (st0 == 0).all() and (st1 == 0).all() and (st2 == 0).all()
    assert np.abs(f2 - f0).max() < 1e-13 < np.abs(f1 - f0).max()
    assert np.abs(g2 - g0).max() < 2e-7 and np.array_equal(g1, g2)
    eng.overlap_gradient(_lib.ANSATZ_SHALLOW_CNOT, Z, tol=1e-12)
    tight = eng.overlap_stats()
    assert loose['rounds_sum'] < 0.85 * tight['rounds_sum']


@pytest.mark.parametrize('kind,P,carry', [(2, 15, False), (0, 8, False), (0, 8, True), (6, 6, False)])
def test_device_resident_bfgs_d2_takes_the_decisions_of_the_host_driver(kind, P, carry, engine_factory):
    """qmps_evolve_bfgs_device (D = 2: the optimiser on the device, one wave per trajectory, the whole run in one launch)
    against qmps_evolve_bfgs (the same iteration with its loop on the host, lock-step): per trajectory the same parameters, the same
    objectives, the same iteration counts - the host driver's floating-point expressions are reproduced operation for operation,
    and both call the same device code for tensors and eigen-solves - and the recorded objectives are the ORACLE's (dense
    eigen-solve) at the device's parameters.  ShallowFull (15 angles: the reference's own case, new_time_evolve.py:186-187),
    ShallowCNOT (scripts/loschmidt.py:203-207), StateGate."""
    rng = np.random.default_rng(900 + kind + P)
    T, n_steps = 7, 3
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    eng = engine_factory(2, T * (2 * P + 1))
    host = eng.evolve_bfgs(kind, X0, WW, n_steps=n_steps, maxiter=40, tol=1e-13, carry_hessian=carry)
    dev = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=40, tol=1e-13, carry_hessian=carry)
    # (the slowest trajectory's count: a line search that fails an iteration earlier or later along a flat direction moves it by a few)
    print('nit dev (max per step)', dev['nit'].max(axis=1), 'host', host['nit'])
    assert dev['nit'].shape == (n_steps, T) and np.abs(dev['nit'].max(axis=1) - host['nit']).max() <= 4
    # (not bit for bit - the compilers contract the two drivers' expressions differently here and there: rounding-level differences,
    # which travel freely along the flat directions of eight / fifteen angles on two qubits; objectives agree to 1e-9)
    print('max |f_dev - f_host|', np.abs(dev['fun'] - host['fun']).max(), 'max |x_dev - x_host|', np.abs(dev['params_hist'] - host['params_hist']).max())
    # first time step: rounding level; later ones: both drivers stop where max|g| < gtol = 1e-5, i.e. within ~1e-8 of the minimum
    assert np.abs(dev['fun'][0] - host['fun'][0]).max() < 1e-8 and np.abs(dev['fun_start'][0] - host['fun_start'][0]).max() < 1e-12
    # (carried inverse Hessians at D = 2 are ill-conditioned along the flat directions: a line search that fails a little earlier or
    # later moves the stopping point by more - DESIGN 5.2: "at D = 2, 4 the identity start is the faster one")
    assert np.abs(dev['fun'] - host['fun']).max() < (1e-5 if carry else 1e-7) and np.abs(dev['fun_start'] - host['fun_start']).max() < 1e-5
    # the parameter vectors through what they describe (flat directions: eight / fifteen angles on two qubits): the same physical states
    for step in range(n_steps):
        for t in range(T):
            o = abs(O.overlap_eta(ER.tensor(kind, 2, dev['params_hist'][step, t]), ER.tensor(kind, 2, host['params_hist'][step, t]), np.eye(4))[0])
            assert abs(o - 1.0) < (1e-4 if carry else 1e-6), (step, t, o)
    assert dev['failed_evaluations'] == 0 and dev['nfev'] > 0 and dev['kernel_ms'] > 0
    # the oracle's objective at the device's parameters: previous parameters = the reference state of the step
    prev = X0
    for step in range(n_steps):
        for t in (0, T - 1):
            f_t = ER.objective(kind, 2, ER.tensor(kind, 2, prev[t]), dev['params_hist'][step, t], WW)
            assert abs(f_t - dev['fun'][step, t]) < F_TOL
        prev = dev['params_hist'][step]
    assert dev['fun'][-1].mean() < -0.999
    # a continued run (inverse Hessians handed back in) = the one-call run
    if carry:
        a = eng.evolve_bfgs_device(kind, X0, WW, n_steps=2, maxiter=40, tol=1e-13, carry_hessian=True)
        b = eng.evolve_bfgs_device(kind, a['x'], WW, n_steps=1, maxiter=40, tol=1e-13, carry_hessian=True, hess_inv=a['hess_inv'])
        # (same kernel, same decisions; since the eigenvalue solves of a pass start from the eigenvalues of the pass before - ABI 6.5 - the first pass of a
        # continued run starts cold where the one-call run's did not: the same numbers to the rounding of a solve, amplified by the flat minimum)
        # (a flipped comparison moves a trajectory along the gate's flat directions: the STATES agree, as host and device do above)
        assert np.median(np.abs(b['x'] - dev['x']).max(axis=1)) < 1e-6 and np.abs(b['fun'][0] - dev['fun'][2]).max() < 1e-5      # (gtol 1e-5: the tolerance host against device gets above)
        for t in range(T):
            o = abs(O.overlap_eta(ER.tensor(kind, 2, b['x'][t]), ER.tensor(kind, 2, dev['x'][t]), np.eye(4))[0])
            assert abs(o - 1.0) < 1e-4, (t, o)


@pytest.mark.parametrize('cls,P', [(R.ShallowCNOTStateTensor_nonuniform, 8), (R.ExactAfter4, 12)])
def test_d2_kinds_without_a_device_resident_kernel_run_the_host_driver(cls, P, engine_factory):
    """ShallowCNOTStateTensor_nonuniform (kind 4) and ExactAfter4 (kind 5) pass check_ansatz at D = 2 but launch_evolve_bfgs_d2 has
    no kernel for them: evolve(method='BFGS') must route them to the host-loop driver (qmps_evolve_bfgs), and the C entry point of
    the device-resident driver must refuse them with QMPS_ERR_ARG and a message - not a HIP launch error."""
    rng = np.random.default_rng(77 + P)
    X0 = rng.standard_normal((3, P))
    WW = WW_of(0.05)
    H, info = NT.evolve(X0, WW, 2, method='BFGS', D=2, state_tensor=cls, options={'maxiter': 30}, return_info=True)
    assert H.shape == (3, 3, P)
    kind = cls.device_kind
    for step in range(2):
        for t in range(3):
            f_t = ER.objective(kind, 2, ER.tensor(kind, 2, H[step, t]), H[step + 1, t], WW)
            assert abs(f_t - info['fun'][step][-1, t]) < F_TOL
            assert f_t <= ER.objective(kind, 2, ER.tensor(kind, 2, H[step, t]), H[step, t], WW) + 1e-12
    eng = engine_factory(2, 3 * (2 * P + 1))
    with pytest.raises(L.QmpsError, match='no device-resident kernel'):
        eng.evolve_bfgs_device(kind, X0, WW, n_steps=1, maxiter=5)
    host = eng.evolve_bfgs(kind, X0, WW, n_steps=1, maxiter=5)           # the context is still usable after the refusal
    assert np.all(np.isfinite(host['fun']))


def test_device_resident_bfgs_d4_against_the_host_driver_and_the_oracle(engine_factory):
    """qmps_evolve_bfgs_device at D = 4 (a workgroup per trajectory: the point eigen-solved by squaring on the matrix cores, its
    neighbours by the two-sided quotient) against qmps_evolve_bfgs (host loop, the same gradient formula): the same minima - objectives to 1e-8, the
    same physical states - and the recorded objectives are the ORACLE's (dense eigen-solve) at the device's parameters.  Also the
    per-trajectory iteration counts: nobody waits for the slowest trajectory."""
    rng = np.random.default_rng(1404)
    kind, P, T, n_steps = 0, 4, 9, 3
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    eng = engine_factory(4, T * (2 * P + 1))
    host = eng.evolve_bfgs(kind, X0, WW, n_steps=n_steps, maxiter=40, tol=1e-13)
    dev = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=40, tol=1e-13)
    assert dev['failed_evaluations'] == 0 and dev['nit'].shape == (n_steps, T)
    print('max |f_dev - f_host|', np.abs(dev['fun'] - host['fun']).max(), 'nit dev (max per step)', dev['nit'].max(axis=1), 'host', host['nit'])
    assert np.abs(dev['fun'] - host['fun']).max() < 1e-7 and np.abs(dev['fun_start'][0] - host['fun_start'][0]).max() < 1e-10
    prev = X0
    for step in range(n_steps):
        for t in range(T):
            f_t = ER.objective(kind, 4, ER.tensor(kind, 4, prev[t]), dev['params_hist'][step, t], WW)
            assert abs(f_t - dev['fun'][step, t]) < F_TOL, (step, t)
            o = abs(O.overlap_eta(ER.tensor(kind, 4, dev['params_hist'][step, t]), ER.tensor(kind, 4, host['params_hist'][step, t]), np.eye(4))[0])
            assert abs(o - 1.0) < 1e-6, (step, t, o)
        prev = dev['params_hist'][step]
    assert dev['fun'][-1].mean() < -0.999 and dev['nit'].min() < dev['nit'].max()
    # a rejected full step sends the workgroup through a ladder pass of its own (the backtracking points are eigen-solved): a short
    # ladder and a crude first rung provoke rejections - the minima do not move
    odd = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=60, tol=1e-13, alphas=(2.5, 1.0, 0.3, 0.05, 0.005))
    assert np.abs(odd['fun'][-1] - dev['fun'][-1]).max() < 1e-6
    # depth 3 (six angles, twelve neighbours on eight waves: some waves evaluate two)
    X6 = rng.standard_normal((3, 6))
    eng6 = engine_factory(4, 3 * 13)
    h6 = eng6.evolve_bfgs(kind, X6, WW, n_steps=2, maxiter=40, tol=1e-13)
    d6 = eng6.evolve_bfgs_device(kind, X6, WW, n_steps=2, maxiter=40, tol=1e-13)
    assert np.abs(d6['fun'] - h6['fun']).max() < 1e-7 and d6['failed_evaluations'] == 0
    with pytest.raises(Exception, match='eight waves'):
        eng.evolve_bfgs_device(kind, X0, WW, alphas=tuple(0.5 ** k for k in range(11)))


def test_a_map_without_a_unique_fixed_point_is_reported_not_hidden():
    """Found by the randomised stress of round 5 (profiles/experiments/r05/stress_evolve.py, 1 trajectory in 4 000 cases): BFGS can walk a
    trajectory INTO a product state - here the D = 8 ShallowCNOT angles (-pi/2, 0, pi/2, 0, 0, 0) - whose mixed transfer map has a whole ring of
    dominant eigenvalues of equal modulus 0.995 and different phases.  There is no unique fixed point: the solves report status 1, the objective is
    NaN from then on, the parameters stay where they are, `evolve` WARNS and names the trajectory - and the other trajectories of the batch
    are untouched (the reference's ARPACK would return one of the eigenvectors and carry on; the two-sided gradient cannot)."""
    D, P = 8, 6
    WW = WW_of(0.1)
    special = np.array([-np.pi / 2, 0.0, np.pi / 2, 0.0, 0.0, 0.0])
    A = ER.tensor(0, D, special)
    w = np.linalg.eigvals(O.transfer_matrix(np.tensordot(WW, O.merge(A, A), [1, 0]), O.merge(A, A)))
    w = np.sort(np.abs(w))[::-1]
    assert int((w > w[0] - 1e-9).sum()) >= 5 and abs(w[0] - 0.99502085) < 1e-7          # many equal moduli on top, no dominant eigenvalue
    X0 = np.stack([special, np.random.default_rng(3).standard_normal(P)])
    with pytest.warns(RuntimeWarning, match='did not converge'):
        H, info = NT.evolve(X0, WW, 2, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-12, options={'maxiter': 40}, return_info=True)
    f_end = np.array([f[-1] for f in info['fun']])
    assert np.all(np.isnan(f_end[:, 0])) and np.all(np.isfinite(f_end[:, 1])) and f_end[:, 1].max() < -0.99
    assert info['no_unique_fixed_point'] == {0: 0}
    assert np.array_equal(H[:, 0], np.stack([special] * 3))                   # the trajectory stays where it was


def test_device_resident_bfgs_d16_against_the_lockstep_driver_and_the_oracle(engine_factory):
    """qmps_evolve_bfgs_device at D = 16 (qmps_evolve_d16.hip: a workgroup of eight waves per trajectory - two teams of four iterate the
    right and the left fixed point on the matrix cores, every wave builds and probes central-difference neighbours in LDS) against the
    lock-step driver qmps_evolve_bfgs, which evaluates the same formulae batch-wise: the same iteration counts, the same minima
    (objectives to 1e-8), the same physical states; the recorded objectives are the ORACLE's (ARPACK in operator form, the
    reference's route) at the device's parameters.  Carried and fresh inverse Hessians, the adaptive and the tight tolerance of
    the gradient's solves, CNOT3 (three angles per layer), a ladder that provokes rejected full steps, and the evolve() option."""
    rng = np.random.default_rng(1616)
    kind, D, P, T, n_steps = 0, 16, 8, 7, 3
    X0 = rng.standard_normal((T, P))
    WW = WW_of(0.05)
    eng = engine_factory(D, T * (2 * P + 1))
    for carry, adaptive, tight in ((False, False, False), (True, True, False), (True, False, True)):
        host = eng.evolve_bfgs(kind, X0, WW, n_steps=n_steps, maxiter=40, tol=1e-12, carry_hessian=carry, adaptive_gradient=adaptive, tight_gradient=tight)
        dev = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=40, tol=1e-12, carry_hessian=carry, adaptive_gradient=adaptive, tight_gradient=tight)
        assert dev['failed_evaluations'] == 0 and dev['nit'].shape == (n_steps, T)
        print('carry', carry, 'adaptive', adaptive, 'tight', tight, 'max |f_dev - f_host|', np.abs(dev['fun'] - host['fun']).max(),
              'nit dev (max per step)', dev['nit'].max(axis=1), 'host', host['nit'])
        assert np.abs(dev['fun'] - host['fun']).max() < 1e-8 and np.abs(dev['fun_start'][0] - host['fun_start'][0]).max() < 1e-10
        assert np.abs(dev['nit'].max(axis=1) - np.asarray(host['nit'])).max() <= 1
        assert dev['fun'][-1].mean() < -0.999
    prev = X0
    for step in range(n_steps):
        for t in range(T):
            f_t = ER.objective(kind, D, ER.tensor(kind, D, prev[t]), dev['params_hist'][step, t], WW, arpack=True)
            assert abs(f_t - dev['fun'][step, t]) < F_TOL, (step, t)
            o = abs(O.overlap_eta_arpack(ER.tensor(kind, D, dev['params_hist'][step, t]), ER.tensor(kind, D, host['params_hist'][step, t]), np.eye(4))[0])
            assert abs(o - 1.0) < 1e-6, (step, t, o)
        prev = dev['params_hist'][step]
    # rejected full steps: the backtracking points are eigen-solved two at a time, the ladder stops at the first rung that passes
    odd = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=60, tol=1e-12, alphas=(2.5, 1.0, 0.3, 0.05, 0.005))
    ref = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=60, tol=1e-12)
    assert odd['failed_evaluations'] == 0 and np.abs(odd['fun'][-1] - ref['fun'][-1]).max() < 1e-6
    # a call of three steps = three calls of one step (parameters and inverse Hessians carried over by the caller)
    x, hinv, f3 = X0, None, []
    for step in range(n_steps):
        r1 = eng.evolve_bfgs_device(kind, x, WW, n_steps=1, maxiter=40, tol=1e-12, carry_hessian=True, hess_inv=hinv)
        x, hinv = r1['x'], r1['hess_inv']
        f3.append(r1['fun'][0])
    whole = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=40, tol=1e-12, carry_hessian=True)
    assert np.abs(np.array(f3) - whole['fun']).max() < 1e-8
    # ShallowCNOTStateTensor3: nine angles, eighteen neighbours on eight waves
    X9 = rng.standard_normal((3, 9))
    eng9 = engine_factory(D, 3 * 19)
    h9 = eng9.evolve_bfgs(3, X9, WW, n_steps=2, maxiter=40, tol=1e-12)
    d9 = eng9.evolve_bfgs_device(3, X9, WW, n_steps=2, maxiter=40, tol=1e-12)
    assert np.abs(d9['fun'] - h9['fun']).max() < 1e-8 and d9['failed_evaluations'] == 0
    with pytest.raises(Exception, match='ShallowCNOT'):
        eng.evolve_bfgs_device(1, X0, WW)
    # evolve(): the lock-step stays the default at D = 16, options {'device_driver': 'trajectory'} selects this driver
    Hd, info_d = NT.evolve(X0, WW, 2, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-12,
                           options={'maxiter': 30, 'device_driver': 'trajectory'}, return_info=True)
    Hl, info_l = NT.evolve(X0, WW, 2, method='BFGS', D=D, state_tensor=R.ShallowCNOTStateTensor, tol=1e-12, options={'maxiter': 30}, return_info=True)
    assert np.abs(np.array([f[-1] for f in info_d['fun']]) - np.array([f[-1] for f in info_l['fun']])).max() < 1e-8
    assert 'gradient_batches' in info_l['solver'] and np.abs(Hd - Hl).max() < 1e-4


@pytest.mark.parametrize('D,P,T,K', [(8, 6, 21, 3), (16, 8, 10, 4), (4, 4, 9, 2), (2, 8, 7, 2)])
def test_lockstep_groups_are_the_same_trajectories(D, P, T, K, engine_factory):
    """qmps_set_evolve_groups: K lock-step groups (a context and a host thread each inside ONE qmps_evolve_bfgs call) against the one
    lock-step over all trajectories - every trajectory takes the same decisions on the same numbers (ragged split: T not a
    multiple of K), the lock-step count is the maximum over the groups, the counters are sums, a continued call finds the groups'
    resident fixed points, qmps_overlap_stats pools the groups."""
    from qmps_amd import _lib
    rng = np.random.default_rng(77 + D)
    WW = WW_of(0.05)
    X0 = rng.standard_normal((T, P))
    kind = _lib.ANSATZ_SHALLOW_CNOT
    one, many = engine_factory(D, T * (2 * P + 1)), engine_factory(D, T * (2 * P + 1) + 1)        # (the factory caches per capacity: two contexts)
    assert one is not many
    one.set_evolve_groups(1)
    many.set_evolve_groups(K)
    many.overlap_stats(reset=True)
    one.overlap_stats(reset=True)
    kw = dict(n_steps=3, maxiter=30, tol=1e-13, carry_hessian=True)
    a = one.evolve_bfgs(kind, X0, WW, **kw)
    b = many.evolve_bfgs(kind, X0, WW, **kw)
    assert np.array_equal(a['nit'], b['nit'])
    assert np.array_equal(a['params_hist'], b['params_hist']) and np.array_equal(a['fun'], b['fun']) and np.array_equal(a['fun_start'], b['fun_start'])
    assert np.array_equal(a['hess_inv'], b['hess_inv']) and np.array_equal(a['x'], b['x'])
    # (scipy's count charges every trajectory for every batch of its lock-step: a group that finishes early stops counting)
    assert 0 < b['nfev'] <= a['nfev'] and b['gradient_batches'] >= a['gradient_batches'] and b['gradient_ms'] > 0
    sa, sb = one.overlap_stats(), many.overlap_stats()
    assert sb['evaluations'] > 0 and sb['not_converged'] == 0 and sa['not_converged'] == 0
    # a continued evolution (resident fixed points of the groups, carried inverse Hessians)
    a2 = one.evolve_bfgs(kind, a['x'], WW, n_steps=2, maxiter=30, tol=1e-13, carry_hessian=True, warm=True, hess_inv=a['hess_inv'])
    b2 = many.evolve_bfgs(kind, b['x'], WW, n_steps=2, maxiter=30, tol=1e-13, carry_hessian=True, warm=True, hess_inv=b['hess_inv'])
    assert np.array_equal(a2['nit'], b2['nit']) and np.array_equal(a2['x'], b2['x']) and np.array_equal(a2['fun'], b2['fun'])
    # a different split of the same trajectories has no resident fixed points: refused like any cold WARM call ... unless this context has them
    many.set_evolve_groups(K + 1)
    if D >= 4:        # (D = 2 solves every candidate cold: nothing resident to continue from)
        with pytest.raises(_lib.QmpsError, match='QMPS_BFGS_WARM'):
            many.evolve_bfgs(kind, b2['x'], WW, n_steps=1, maxiter=30, tol=1e-13, warm=True)
    c = many.evolve_bfgs(kind, b2['x'], WW, n_steps=1, maxiter=30, tol=1e-13)
    assert c['fun'][-1].mean() < -0.999
    assert many.evolve_groups(T) == K + 1 and one.evolve_groups(T) == 1
    # errors of a group come back with the group named
    with pytest.raises(_lib.QmpsError, match='lock-step group'):
        many.evolve_bfgs(kind, X0, WW, n_steps=1, maxiter=30, gtol=-1.0)
    many.set_evolve_groups(0)
    one.set_evolve_groups(0)
    assert [many.evolve_groups(t) for t in (1, 256, 511, 512, 1024, 4096)] == [1, 1, 1, 2, 4, 4]


@pytest.mark.parametrize('D,P,T', [(8, 6, 600), (16, 8, 520)])
def test_lockstep_groups_automatic_at_full_size(D, P, T, engine_factory):
    """From 512 trajectories on qmps_evolve_bfgs groups by itself (two groups here): the same evolution, bit for bit, as one
    lock-step over all of them (D = 16: 520 iterates in one launch and 260 in another go through the same tensor builder - a
    trajectory's numbers must not depend on how many others share its launch), and every trajectory ends at its minimum."""
    from qmps_amd import _lib
    rng = np.random.default_rng(2024)
    WW = WW_of(0.05)
    X0 = rng.standard_normal((T, P))
    auto, one = engine_factory(D, T * (2 * P + 1)), engine_factory(D, T * (2 * P + 1) + 1)
    auto.set_evolve_groups(0)
    one.set_evolve_groups(1)
    assert auto.evolve_groups(T) == 2 and one.evolve_groups(T) == 1
    kw = dict(n_steps=2, maxiter=40, tol=1e-12, carry_hessian=True)
    a = auto.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, **kw)
    b = one.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, **kw)
    assert np.array_equal(a['params_hist'], b['params_hist']) and np.array_equal(a['fun'], b['fun']) and np.array_equal(a['nit'], b['nit'])
    assert np.all(np.isfinite(a['fun'])) and a['fun'][-1].max() < -0.99
    one.set_evolve_groups(0)


def test_lockstep_groups_edge_cases(engine_factory):
    """More groups than trajectories (one trajectory per group), two trajectories in two groups, a single trajectory (never
    grouped), history pointers left out: the grouped call behaves like the plain one."""
    from qmps_amd import _lib
    D, P = 8, 6
    rng = np.random.default_rng(99)
    WW = WW_of(0.05)
    kind = _lib.ANSATZ_SHALLOW_CNOT
    many, one = engine_factory(D, 5 * (2 * P + 1)), engine_factory(D, 5 * (2 * P + 1) + 1)
    one.set_evolve_groups(1)
    for T, K, expect in ((5, 16, 5), (2, 2, 2), (1, 4, 1)):
        X0 = rng.standard_normal((T, P))
        many.set_evolve_groups(K)
        assert many.evolve_groups(T) == expect
        a = many.evolve_bfgs(kind, X0, WW, n_steps=2, maxiter=30, tol=1e-13)
        b = one.evolve_bfgs(kind, X0, WW, n_steps=2, maxiter=30, tol=1e-13)
        assert np.array_equal(a['x'], b['x']) and np.array_equal(a['fun'], b['fun']) and np.array_equal(a['nit'], b['nit'])
        # counters and histories are optional at the C level
        lib, ctx = many._lib, many._ctx
        Pm = np.array(X0, dtype=np.float64, order='C', copy=True)
        fh = np.empty((2, 2, T))
        al = np.array([1.0, 0.5, 0.25])
        import ctypes
        dp = lambda v: v.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        rc = lib.qmps_evolve_bfgs(ctx, T, kind, P, dp(Pm), dp(np.ascontiguousarray(WW).view(np.float64)), 2, 30, 1e-5, 1e-6, 1e-4, 3, dp(al), 0, 100000, 1e-13,
                                  None, None, dp(fh), None, None)
        assert rc == 0 and np.all(np.isfinite(fh)) and np.all(fh[:, 1] <= fh[:, 0] + 1e-12)
    many.set_evolve_groups(0)
    one.set_evolve_groups(0)


def test_native_bfgs_driver_argument_checks_and_single_rung(engine_factory):
    """Refusals of qmps_evolve_bfgs (batch larger than the context, a warm continuation without resident fixed points)
    and the ladder-free variant (one step length: a rejected full step ends the trajectory's minimisation)."""
    from qmps_amd import _lib
    from qmps_amd._lib import QmpsError
    WW = WW_of(0.05)
    rng = np.random.default_rng(5)
    eng = engine_factory(4, 4 * 9)
    X0 = rng.standard_normal((4, 4))
    with pytest.raises(QmpsError, match='exceed max_batch'):
        eng.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, rng.standard_normal((5, 4)), WW)
    with pytest.raises(QmpsError, match='QMPS_BFGS_WARM'):
        eng.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, warm=True)
    with pytest.raises(ValueError, match='hess_inv'):
        eng.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, warm=True, carry_hessian=True)
    full = eng.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, n_steps=2, maxiter=40, tol=1e-13)
    one = eng.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, n_steps=2, maxiter=40, alphas=(1.0,), tol=1e-13)
    assert one['ladder_batches'] == 0 and full['nfev'] >= one['gradient_batches'] * 4 * 9
    # without a ladder a trajectory whose full step is rejected stops early: never better than the full driver by more than rounding
    assert np.all(full['fun'][-1] <= one['fun'][-1] + 1e-9)
    assert np.all(np.isfinite(full['fun'])) and full['fun'][-1].mean() < -0.999
    # a continued call starts where the previous one stopped
    cont = eng.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, full['x'], WW, n_steps=1, maxiter=40, tol=1e-13, warm=True)
    assert cont['fun'].shape == (1, 4) and cont['fun'][0].mean() < -0.999
    # QMPS_BFGS_TIGHT_GRADIENT: the gradient batches' eigen-solves iterate to tol instead of 1e-8 - the same minima
    tight = eng.evolve_bfgs(_lib.ANSATZ_SHALLOW_CNOT, X0, WW, n_steps=2, maxiter=40, tol=1e-13, tight_gradient=True)
    assert np.abs(tight['fun'] - full['fun']).max() < 1e-9


def test_reference_signature_single_trajectory(engine_factory):
    """evolve(params (P,), ...) keeps the reference's shape: history (n_steps + 1, P); scipy method per trajectory."""
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal(15)
    WW = WW_of(0.02)
    H = NT.evolve(p0, WW, 2, method='Rotosolve', n_sweeps=2)
    assert H.shape == (3, 15)
    A = NT.state_tensor(H[0])
    f = NT.obj(H[1], A, WW)
    assert abs(f - O.overlap_objective(A, NT.state_tensor(H[1]), WW)) < 1e-10
    Hb, info = NT.evolve(p0, WW, 2, method='BFGS', return_info=True)
    assert Hb.shape == (3, 15)
    fb = NT.obj(Hb[1], A, WW)
    assert fb < NT.obj(Hb[0], A, WW) and fb < -0.9999              # BFGS improved on "do not move" and projects W|A A> well
    Hs = NT.evolve(p0, WW, 1, method='Nelder-Mead', options={'maxiter': 60})
    assert Hs.shape == (2, 15)
    # D = 4 through the scalar objective
    p4 = rng.standard_normal(4)
    A4 = NT.state_tensor(p4, D=4)
    assert abs(NT.obj(p4 + 0.01, A4, WW) - O.overlap_objective(A4, NT.state_tensor(p4 + 0.01, D=4), WW)) < 1e-10


@pytest.mark.parametrize('D,P', [(4, 4), (16, 8)])
def test_active_mask_skips_trajectories_and_keeps_their_values(D, P, engine_factory):
    """qmps_overlap_set_active: a one-shot per-trajectory mask - the masked trajectories' candidates / iterates are skipped (no
    solver rounds counted) and their outputs keep the previous launch's values; the others are evaluated as usual."""
    rng = np.random.default_rng(60 + D)
    eng = engine_factory(D, 4096)
    T, G = 6, 3
    WW = WW_of(0.05)
    ref = rng.standard_normal((T, P))
    cand = (ref[:, None, :] + 0.03 * rng.standard_normal((T, G, P))).reshape(T * G, P)
    eng.overlap_set_refs_params(0, ref, WW)
    eng.overlap_set_group(G)
    f0, st0 = eng.overlap_eval_params(0, cand, tol=1e-13)
    moved = cand + 0.01 * rng.standard_normal(cand.shape)
    mask = np.array([1, 0, 1, 1, 0, 1], dtype=bool)
    eng.overlap_stats(reset=True)
    eng.overlap_set_active(mask)
    f1, st1 = eng.overlap_eval_params(0, moved, tol=1e-13)
    assert eng.overlap_stats()['evaluations'] == int(mask.sum()) * G
    f_all, _ = eng.overlap_eval_params(0, moved, tol=1e-13)            # the mask was one-shot: everything is evaluated again
    for t in range(T):
        rows = slice(t * G, (t + 1) * G)
        if mask[t]:
            assert np.abs(f1[rows] - f_all[rows]).max() < 1e-13
        else:
            assert np.array_equal(f1[rows], f0[rows]) and np.abs(f_all[rows] - f0[rows]).max() > 1e-9
    # the gradient entry point
    eng.overlap_set_group(0)
    X = ref + 0.02 * rng.standard_normal((T, P))
    fa, ga, _ = eng.overlap_gradient(0, X, tol=1e-13)
    X2 = X + 0.01 * rng.standard_normal(X.shape)
    eng.overlap_set_active(mask)
    fb, gb, _ = eng.overlap_gradient(0, X2, tol=1e-13)
    fc, gc, _ = eng.overlap_gradient(0, X2, tol=1e-13)
    assert np.array_equal(fb[~mask], fa[~mask]) and np.array_equal(gb[~mask], ga[~mask])
    assert np.abs(fb[mask] - fc[mask]).max() < 1e-12 and np.abs(gb[mask] - gc[mask]).max() < 1e-7
    with pytest.raises(L.QmpsError):
        eng.overlap_set_active(mask[:3])
        eng.overlap_gradient(0, X2, tol=1e-13)                          # three entries for six trajectories
    eng.overlap_set_active(None)


def test_option_struct_entry_points_are_the_positional_ones(engine_factory):
    """ABI 6.1: qmps_evolve_bfgs_opts / qmps_evolve_bfgs_device_opts (versioned structs, defaults from qmps_evolve_opts_init) against the
    positional entry points: the same code, the same numbers; defaults = scipy's BFGS settings and the eight-rung ladder."""
    import ctypes
    from qmps_amd.engine import _f64, _i32
    rng = np.random.default_rng(61)
    T, P = 5, 15
    X0 = rng.standard_normal((T, P))
    WW = np.ascontiguousarray(WW_of(0.05), dtype=np.complex128)
    eng = engine_factory(2, T * (2 * P + 1))
    ref = eng.evolve_bfgs_device(L.ANSATZ_SHALLOW_FULL, X0, WW, n_steps=2, maxiter=200, tol=1e-12)
    lib = eng._lib
    opts = L.EvolveOpts()
    L.check(lib.qmps_evolve_opts_init(ctypes.byref(opts)))
    opts.n_steps = 2                                   # everything else: the defaults (maxiter 200, gtol 1e-5, h 1e-6, c1 1e-4, tol 1e-12, default ladder)
    X = X0.copy()
    ph, fh, nit, cnt = np.empty((2, T, P)), np.empty((2, 2, T)), np.zeros((2, T), dtype=np.int32), np.zeros(4)
    out = L.EvolveOut(size=ctypes.sizeof(L.EvolveOut), params_hist=_f64(ph), f_hist=_f64(fh), nit=_i32(nit), counters=_f64(cnt))
    L.check(lib.qmps_evolve_bfgs_device_opts(eng._ctx, T, L.ANSATZ_SHALLOW_FULL, P, _f64(X), _f64(WW.view(np.float64)), ctypes.byref(opts), ctypes.byref(out)))
    assert np.array_equal(X, ref['x']) and np.array_equal(fh[:, 1], ref['fun']) and np.array_equal(nit, ref['nit'])
    # a caller compiled against a SHORTER struct (only size .. max_rounds): the rest takes the defaults
    short = L.EvolveOpts()
    L.check(lib.qmps_evolve_opts_init(ctypes.byref(short)))
    short.n_steps = 2
    short.size = L.EvolveOpts.gtol.offset
    short.gtol = 123.0                                 # beyond `size`: must be ignored
    X2 = X0.copy()
    L.check(lib.qmps_evolve_bfgs_device_opts(eng._ctx, T, L.ANSATZ_SHALLOW_FULL, P, _f64(X2), _f64(WW.view(np.float64)), ctypes.byref(short), ctypes.byref(out)))
    assert np.array_equal(X2, ref['x'])


def test_option_struct_defaults_at_d16_cap_power_steps_not_squarings(engine_factory):
    """Advisor finding of round 5: qmps_evolve_bfgs_device_opts turned max_rounds = 0 into 60 at every D; at D = 16 max_rounds caps the POWER steps
    of a backtracking point's solve (qmps_evolve_d16.hip), so a C caller keeping the qmps_evolve_opts_init defaults had nearly every rung
    of a rejected full step exhaust its cap and count as rejected.  The default is now the one include/qmps_hip.h documents (60 squarings at
    D = 2, 4; 100 000 power steps at D = 8, 16), as in qmps_evolve_bfgs_opts: the struct call with max_rounds = 0 equals the positional call
    with 100 000 bit for bit, on a ladder that provokes rejected full steps."""
    import ctypes
    from qmps_amd.engine import _f64, _i32
    rng = np.random.default_rng(1617)
    kind, D, P, T, n_steps = 0, 16, 8, 5, 2
    X0 = rng.standard_normal((T, P))
    WW = np.ascontiguousarray(WW_of(0.05), dtype=np.complex128)
    al = np.array([2.5, 1.0, 0.3, 0.05, 0.005])
    eng = engine_factory(D, T * (2 * P + 1))
    ref = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=60, tol=1e-12, alphas=al, max_rounds=100000)
    assert ref['failed_evaluations'] == 0
    lib = eng._lib
    opts = L.EvolveOpts()
    L.check(lib.qmps_evolve_opts_init(ctypes.byref(opts)))
    assert opts.max_rounds == 0
    opts.n_steps, opts.maxiter, opts.n_alphas, opts.alphas = n_steps, 60, len(al), _f64(al)
    X = X0.copy()
    ph, fh, nit, cnt = np.empty((n_steps, T, P)), np.empty((n_steps, 2, T)), np.zeros((n_steps, T), dtype=np.int32), np.zeros(4)
    out = L.EvolveOut(size=ctypes.sizeof(L.EvolveOut), params_hist=_f64(ph), f_hist=_f64(fh), nit=_i32(nit), counters=_f64(cnt))
    L.check(lib.qmps_evolve_bfgs_device_opts(eng._ctx, T, kind, P, _f64(X), _f64(WW.view(np.float64)), ctypes.byref(opts), ctypes.byref(out)))
    assert int(cnt[1]) == 0                                                  # no failed evaluation
    assert np.array_equal(nit, ref['nit']) and np.abs(fh[:, 1] - ref['fun']).max() < 1e-9 and np.abs(X - ref['x']).max() < 1e-6
    assert fh[-1, 1].mean() < -0.999


def test_d2_device_driver_characteristic_polynomial_solve_against_squaring_and_the_oracle(engine_factory, monkeypatch):
    """Round 6: the D = 2 device-resident BFGS driver takes eta of a candidate's 4 x 4 map as the largest root of its characteristic polynomial
    (power sums from one product, Newton's identities, Aberth's iteration with a root per lane of the quad: overlap_quad_charpoly, qmps_evolve_d2.hip)
    instead of squaring the map until rank one (QMPS_EVOLVE_D2_SQUARING=1: rounds 4-5).  Both against the oracle's dense eigen-solve at the
    device's parameters (1e-10 on generic starts: the recorded objective IS -sqrt|eta| of the recorded parameters), on random starts and on starts at the special
    grid (a fifth of them: tied moduli, nilpotent maps, clusters - the cases a root finder could trip over): no NaN that the squaring solve does not
    have, no failed evaluation, the same minima where the two BFGS runs stay in one basin; Aberth needs fewer rounds than the squaring."""
    rng = np.random.default_rng(66)
    WW = WW_of(0.05)
    for kind, P in ((L.ANSATZ_SHALLOW_FULL, 15), (L.ANSATZ_SHALLOW_CNOT, 8), (L.ANSATZ_SHALLOW_CNOT, 2)):
        T, n_steps = 40, 3
        X0 = rng.standard_normal((T, P))
        X0[::5] = np.round(X0[::5] / (np.pi / 4)) * (np.pi / 4)
        eng = engine_factory(2, T * (2 * P + 1 + 8))
        res = {}
        for mode in ('charpoly', 'squaring'):
            if mode == 'squaring':
                monkeypatch.setenv('QMPS_EVOLVE_D2_SQUARING', '1')
            else:
                monkeypatch.delenv('QMPS_EVOLVE_D2_SQUARING', raising=False)
            res[mode] = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=60, tol=1e-13)
        monkeypatch.delenv('QMPS_EVOLVE_D2_SQUARING', raising=False)
        a, b = res['charpoly'], res['squaring']
        assert a['failed_evaluations'] == 0 and int(np.isnan(a['fun']).sum()) <= int(np.isnan(b['fun']).sum())
        assert a['squarings'] < b['squarings']                     # (Aberth iterations against squaring rounds, summed over all evaluations)
        prev = X0
        for step in range(n_steps):
            for t in range(T):
                if not np.isfinite(a['fun'][step, t]):
                    continue
                f_or = ER.objective(kind, 2, ER.tensor(kind, 2, prev[t]), a['params_hist'][step, t], WW)
                # (starts ON the grid can sit on double dominant eigenvalues, where a root of the quartic - and an eigenvalue from numpy's eig, the
                # oracle here - is only good to ~sqrt(eps): the suite's F_TOL there, 1e-10 on the generic starts)
                assert abs(f_or - a['fun'][step, t]) < (F_TOL if t % 5 == 0 else 1e-10), (kind, P, step, t)
            prev = a['params_hist'][step]
        both = np.isfinite(a['fun'][-1]) & np.isfinite(b['fun'][-1])
        close = np.abs(a['fun'][-1] - b['fun'][-1])[both] < 1e-6
        assert close.mean() > 0.8 and a['fun'][-1][both].mean() < -0.99


@pytest.mark.parametrize('D', [2, 4])
def test_device_drivers_on_the_special_grid_against_gelfand(D, engine_factory):
    """Round 6, second pass over the device-resident drivers: starts ON the grid of multiples of pi / 4 and pi / 2 - product states, permutation-like
    tensors, non-injective states.  D = 2: their maps carry what a root finder of the quartic cannot answer (multiple largest roots: eps^(1/m)
    conditioning, linear convergence; nilpotent maps; near-clusters: eps / kappa) or trips over (a centroid that is itself a root: spectrum
    {l, -l, 0, 0}; polynomials with the symmetry of the starting polygon); the kernel hands the first class to its squaring solve (`fallback`,
    overlap_quad_charpoly) and starts the iteration off the polygon.  D = 4 (ABI 6.5): tied dominant eigenvalues (1, 1, -1, -1) used to come back as the
    quotient of a noise-picked direction with status 0 once rounding broke the tie (round ~53 of the driver's 60: |eta| = 1.0008, 0.54 where it is 1,
    0.999) - now their common modulus, and the neighbours of such a point are eigen-solved one by one (no fixed points to expand them round).  Checked at EVERY recorded
    point - the start and the end of each step - against Gelfand's formula (tests/evolve_replay.spectral_radius: good to 1e-13 where numpy's eigvals
    is not), 1e-9 throughout: no NaN, no failed evaluation."""
    rng = np.random.default_rng(606 + D)
    cases = ((L.ANSATZ_SHALLOW_CNOT, 2), (L.ANSATZ_SHALLOW_CNOT, 8), (L.ANSATZ_SHALLOW_FULL, 15)) if D == 2 else ((L.ANSATZ_SHALLOW_CNOT, 4), (L.ANSATZ_SHALLOW_CNOT, 8))
    for dt in (0.0, 0.05, 0.3):
        WW = WW_of(dt)
        for kind, P in cases:
            if P == 2:
                g = np.arange(-4, 5) * (np.pi / 4)
                X0 = np.array([(a, b) for a in g for b in g])
            else:
                X0 = np.concatenate([rng.integers(-4, 5, (60, P)) * (np.pi / 4), rng.integers(-2, 3, (60, P)) * (np.pi / 2)])
            T, n_steps = len(X0), 2
            eng = engine_factory(D, max(4096, T * (2 * P + 1 + 8)))
            res = eng.evolve_bfgs_device(kind, X0, WW, n_steps=n_steps, maxiter=4, tol=1e-13)
            assert res['failed_evaluations'] == 0
            prev = X0
            worst = 0.0
            for step in range(n_steps):
                assert np.isfinite(res['fun'][step]).all() and np.isfinite(res['fun_start'][step]).all()
                for t in range(T):
                    A = ER.tensor(kind, D, prev[t])
                    worst = max(worst, abs(ER.objective_gelfand(kind, D, A, prev[t], WW) - res['fun_start'][step, t]),
                                abs(ER.objective_gelfand(kind, D, A, res['params_hist'][step, t], WW) - res['fun'][step, t]))
                prev = res['params_hist'][step]
            assert worst < 1e-9, (D, dt, kind, P, worst)


def test_d4_device_driver_leaves_a_tied_start_as_scipy_does(engine_factory):
    """ABI 6.5: a D = 4 trajectory STARTED at a non-injective state of the special grid (the map carries 1, 1, -1, -1: QMPS_STATUS_TIED) has an objective -
    the tie's common modulus - but no fixed points for the second-order expansion of its 2 P neighbours: those are eigen-solved one by one
    (solve_tied_neighbour, qmps_evolve_d4.hip), which is what the reference's finite differences over ARPACK see.  From four such starts the device
    reaches the minimum scipy's BFGS reaches on the same objective (Gelfand's formula on the CPU: tests/evolve_replay.objective_gelfand) in about as
    many iterations; a tied point that is stationary (the fifth) stays, as scipy stays."""
    from scipy.optimize import minimize
    grid = np.array([[2, -4, 0, 4], [2, 4, 0, -2], [-2, -4, 0, 2], [-4, 0, 2, 2], [4, 2, -4, 2]]) * (np.pi / 4)
    kind = L.ANSATZ_SHALLOW_CNOT
    eng = engine_factory(4, 4096)
    for dt in (0.05, 0.3):
        WW = WW_of(dt)
        res = eng.evolve_bfgs_device(kind, grid, WW, n_steps=1, maxiter=30, tol=1e-13)
        assert res['failed_evaluations'] == 0
        for t, x0 in enumerate(grid):
            A = ER.tensor(kind, 4, x0)
            f = lambda x: ER.objective_gelfand(kind, 4, A, x, WW)
            assert abs(res['fun_start'][0, t] - f(x0)) < 1e-12 and abs(res['fun'][0, t] - f(res['params_hist'][0, t])) < 1e-12
            sp = minimize(f, x0, method='BFGS', options={'maxiter': 30})
            if t < 4:
                assert res['nit'][0, t] >= 5 and res['fun'][0, t] < res['fun_start'][0, t] - 1e-4, (dt, t, res['nit'][0, t])
                assert abs(res['fun'][0, t] - sp.fun) < 1e-7 and abs(int(res['nit'][0, t]) - sp.nit) <= 4, (dt, t, res['fun'][0, t], sp.fun, res['nit'][0, t], sp.nit)
            else:
                assert res['nit'][0, t] == 0 and sp.nit == 0 and abs(res['fun'][0, t] - sp.fun) < 1e-12
