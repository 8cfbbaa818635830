"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle.

fp64 tolerance (BASELINE.json north_star): energies within 1e-10 of the fp64 restatement of the
reference maths.  Iteration counts may differ from the oracle's by +-1 (FMA contraction changes
the last bit of the convergence test).
"""
import numpy as np
import pytest

from oracle import qmps_oracle as O

pytestmark = pytest.mark.gpu

E_TOL = 1e-10
R_TOL = 1e-10
HANDOFF = {2: 64, 4: 128}        # hand-off points exercised by the 'squaring' variant (library default: 0)
SKIP0 = {2: 3, 4: 6}             # QMPS_SKIP_ROUNDS_D2/D4: untracked squarings when handoff == 0
PERIOD = {2: 0, 4: 4}            # QMPS_MATVEC_PERIOD_D4: D = 4 continues with mat-vecs of T^(2^m), squaring every 4th


SOLVERS = ['plain', 'squaring', 'squaring0']     # squaring0: hand-off after 0 plain steps = squaring from the start


def oracle_handoff(D, solver):
    if solver == 'plain' or D >= 8:
        return None
    return HANDOFF[D] if solver == 'squaring' else 0


def select(eng, solver):
    if solver == 'plain':
        eng.set_solver('plain', handoff=HANDOFF.get(eng.D, 0))
    else:
        eng.set_solver('squaring', handoff=HANDOFF.get(eng.D, 0) if solver == 'squaring' else 0)


@pytest.mark.parametrize('solver', SOLVERS)
@pytest.mark.parametrize('D', [2, 4, 8, 16])
def test_golden_vectors(D, solver, golden, engine_factory):
    """Committed fixtures: reference-generated A, oracle E (closed form == state-vector path)."""
    eng = engine_factory(D)
    select(eng, solver)
    A = golden[f'ref_A_D{D}']
    h = golden['ref_h_tfim']
    E, it, st = eng.energies(A, h)
    assert np.all(st == 0)
    assert np.abs(E[:, 0] - golden[f'oracle_E_closed_D{D}']).max() < E_TOL
    if D <= 8:
        assert np.abs(E[:, 0] - golden[f'oracle_E_statevec_D{D}']).max() < E_TOL
    ho = oracle_handoff(D, solver)
    if ho is None:
        assert np.all(np.abs(it - golden[f'oracle_iters_D{D}']) <= 1)
    else:
        assert eng.handoff == ho
        slow = golden[f'oracle_iters_D{D}'] > ho + 1
        assert np.all(it[slow] > ho) and np.all(np.abs(it - golden[f'oracle_iters_D{D}'])[~slow] <= 1)
        if PERIOD[D] == 0:
            assert np.all(np.log2(it[slow] - ho) % 1 == 0)          # handoff + 2^m
        else:
            exp = np.array([O.env_power_iteration(a, handoff=ho, skip=SKIP0[D] if ho == 0 else 0, period=PERIOD[D])[1]
                            for a in A])
            assert (it == exp).mean() > 0.9 and np.all(np.abs(it - exp)[slow] <= np.minimum(it, exp)[slow] - ho)
    r = eng.environments()
    assert np.abs(r - golden[f'oracle_r_D{D}']).max() < R_TOL
    # same through the unitary input kind (device-side unitary_to_tensor)
    E2, _, _ = eng.energies(golden[f'U_D{D}'], h, kind='unitary')
    assert np.array_equal(E, E2)


@pytest.mark.parametrize('solver', SOLVERS)
@pytest.mark.parametrize('D,B', [(2, 1), (2, 63), (2, 4096), (4, 1), (4, 65), (4, 1000), (4, 5000), (8, 37), (16, 9)])
def test_random_batches_vs_c_oracle(D, B, solver, c_oracle, engine_factory):
    rng = np.random.default_rng(1000 * D + B)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, B))
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}),
                  O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}),
                  rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))])
    eng = engine_factory(D)
    select(eng, solver)
    E, it, st = eng.energies(A, h, max_iter=4000)
    ref = c_oracle.energy_batch(A, h, max_iter=4000, want_r=True, want_rho=True, handoff=oracle_handoff(D, solver), skip=SKIP0[D] if oracle_handoff(D, solver) == 0 else 0,
                                period=PERIOD.get(D, 0))
    ok = (st == 0) & (ref['status'] == 0)
    assert ok.mean() > 0.9
    assert np.array_equal(st == 1, ref['status'] == 1) or np.abs(it - ref['iters']).max() <= 1
    assert np.abs(E - ref['E'])[ok].max() < E_TOL
    # plain steps: +-1 (FMA contraction flips a borderline convergence test); in the D = 2 squaring tail such a
    # flip doubles/halves 2^m; an item on the hand-off boundary may land on either side of it
    ho = oracle_handoff(D, solver)
    di = np.abs(it - ref['iters'])
    good = di <= 1
    if ho is not None:
        lo, hi = np.minimum(it, ref['iters']), np.maximum(it, ref['iters'])
        with np.errstate(divide='ignore', invalid='ignore'):
            ratio = (it - ho) / (ref['iters'] - ho).astype(float)
        good |= (lo > ho) & ((ratio == 2) | (ratio == 0.5))
        if PERIOD[D] > 0:      # a flipped test costs one more product with T^(2^m) (2^m <= steps taken so far)
            good |= (lo > ho) & (di <= lo - ho)
        good |= (lo >= ho - 1) & (hi <= ho + 2)
    assert good[ok].all() and (di[ok] == 0).mean() > 0.95
    # the two solvers agree with each other far inside the tolerance
    plain = c_oracle.energy_batch(A, h, max_iter=4000)
    both = ok & (plain['status'] == 0)
    assert np.abs(E - plain['E'])[both].max() < E_TOL
    assert np.abs(eng.environments() - ref['r'])[ok].max() < R_TOL
    assert np.abs(eng.rdm() - ref['rho'])[ok].max() < R_TOL
    # device-side summed cost == host sum
    assert np.allclose(eng.summed_cost(), E.sum(0), rtol=0, atol=1e-9 * max(1, B))


@pytest.mark.parametrize('D', [2, 4, 8])
def test_warm_start_and_energy_only(D, c_oracle, engine_factory):
    rng = np.random.default_rng(7 + D)
    B = 200
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, B))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 0.7})
    eng = engine_factory(D)
    select(eng, 'squaring')
    E, it, st = eng.energies(A, h)
    r = eng.environments()
    # warm start from the converged environment: converges immediately, same energies
    E2, it2, st2 = eng.energies(A, h, r0=r)
    assert np.all(st2 == 0) and it2.max() <= (8 if D == 4 else 2)   # D = 4: the fixed point of the squared matrix sits ~1e-14 off
    assert np.abs(E2 - E).max() < E_TOL
    # energy-only launch on the resident (A, r)
    eng.launch_energy_only()
    E3, _, _ = eng.results()
    assert np.abs(E3 - E).max() < 1e-13
    # warm start scaled by an arbitrary positive trace and perturbed
    r0 = 3.7 * r + 1e-3 * np.eye(D)
    E4, it4, st4 = eng.energies(A, h, r0=r0)
    assert np.all(st4 == 0) and np.abs(E4 - E).max() < E_TOL and it4.mean() < it.mean()


def test_squaring_tail_heavy_tail_D2(c_oracle, engine_factory):
    """D = 2 iteration counts are heavy-tailed (p99.9 ~ 3000 plain steps): the in-lane squaring tail
    reaches the same fixed point in O(log K) rounds.  Also a near-degenerate transfer matrix."""
    rng = np.random.default_rng(99)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 4, 20000))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    eng = engine_factory(2)
    select(eng, 'squaring')
    E, it, st = eng.energies(A, h, max_iter=100000)
    ref = c_oracle.energy_batch(A, h, max_iter=100000, handoff=64, skip=0)
    ok = (st == 0) & (ref['status'] == 0)
    assert ok.mean() > 0.999 and it.max() > 2000
    assert np.abs(E - ref['E'])[ok].max() < E_TOL
    select(eng, 'plain')
    E2, it2, st2 = eng.energies(A, h, max_iter=100000)
    select(eng, 'squaring0')
    E3, it3, st3 = eng.energies(A, h, max_iter=100000)
    assert np.abs(E - E3)[ok & (st3 == 0)].max() < E_TOL and (st3 == 0).mean() > 0.999
    ok2 = ok & (st2 == 0)
    assert np.abs(E - E2)[ok2].max() < E_TOL


def test_edge_cases(engine_factory):
    eng = engine_factory(4)
    select(eng, 'squaring')
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    # empty batch
    E, it, st = eng.energies(np.zeros((0, 2, 4, 4), complex), h)
    assert E.shape == (0, 1)
    # product state: r has rank 1 -> not positive definite (the reference's LinAlgError branch)
    U = np.eye(8, dtype=complex)[None]
    E, it, st = eng.energies(U, h, kind='unitary')
    assert st[0] in (1, 2)
    assert abs(E[0, 0] - (-1.0)) < 1e-12      # |00..0>: <ZZ> = 1, <X> = 0
    # max_iter cut-off is reported, not hidden
    rng = np.random.default_rng(3)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 10))
    E, it, st = eng.energies(A, h, max_iter=3)
    assert np.all(st == 1) and np.all(it == 3)


def test_argument_errors(engine_factory):
    from qmps_amd._lib import QmpsError
    eng = engine_factory(4)
    with pytest.raises(ValueError):
        eng.energies(np.zeros((3, 2, 2, 2), complex), np.eye(4))
    with pytest.raises(QmpsError):
        eng.energies(np.zeros((3, 2, 4, 4), complex), np.eye(4), max_iter=0)
    from qmps_amd import EnergyEngine
    with pytest.raises(QmpsError):
        EnergyEngine(3, 10)


@pytest.mark.parametrize('D', [2, 4])
def test_non_isometric_tensors_stay_finite(D, engine_factory):
    """Tensors that are not left isometries (dominant transfer eigenvalue eta != 1): the environment is still the
    dominant right eigen-matrix (what xmps TransferMatrix(A).eigs() returns), no overflow/underflow in the
    squared matrices; both solvers agree."""
    rng = np.random.default_rng(77 + D)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, 300))
    A = A * rng.uniform(0.6, 1.5, size=(300, 1, 1, 1)) + 0.05 * (rng.standard_normal(A.shape) + 1j * rng.standard_normal(A.shape))
    eng = engine_factory(D)
    select(eng, 'squaring0')
    r1, it1, st1 = eng.env_batch(A, max_iter=100000)
    select(eng, 'plain')
    r2, it2, st2 = eng.env_batch(A, max_iter=100000)
    ok = (st1 == 0) & (st2 == 0)
    assert ok.mean() > 0.95 and np.isfinite(r1).all()
    assert np.abs(r1 - r2)[ok].max() < 1e-10
    for k in np.flatnonzero(ok)[:30]:
        _, r_ref = O.env_dense_eig(A[k])
        assert np.abs(r1[k] - r_ref).max() < 1e-10


@pytest.mark.parametrize('max_iter', [1, 2, 5, 63, 64, 65, 130, 300])
def test_d4_default_schedule_iteration_caps(max_iter, c_oracle, engine_factory):
    """D = 4 library default (squaring from the start, then mat-vecs with T^(2^m)): iteration counts, status and
    environments follow the oracle's restatement of the schedule for every cap on the number of power steps."""
    rng = np.random.default_rng(4242)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 500))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    eng = engine_factory(4)
    select(eng, 'squaring0')
    assert eng.squaring_schedule == (SKIP0[4], PERIOD[4])
    E, it, st = eng.energies(A, h, max_iter=max_iter)
    ref = c_oracle.energy_batch(A, h, max_iter=max_iter, want_r=True, handoff=0, skip=SKIP0[4], period=PERIOD[4])
    assert (it == ref['iters']).mean() > 0.97 and np.all(it <= max_iter)
    same = it == ref['iters']
    assert np.array_equal(st[same] == 1, ref['status'][same] == 1)
    assert np.abs(eng.environments() - ref['r'])[same].max() < R_TOL
    assert np.abs(E - ref['E'])[same].max() < E_TOL


def test_d4_default_schedule_warm_start(c_oracle, engine_factory):
    """Warm start through the D = 4 matrix kernel (handoff 0): start matrix of any positive trace, not Hermitian
    to rounding; same fixed point, oracle-identical step counts."""
    rng = np.random.default_rng(4243)
    B = 300
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    h = O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})
    eng = engine_factory(4)
    select(eng, 'squaring0')
    E, it, st = eng.energies(A, h)
    r = eng.environments()
    G = rng.standard_normal((B, 4, 4)) + 1j * rng.standard_normal((B, 4, 4))
    r0 = 2.5 * r + 0.05 * (G @ G.conj().transpose(0, 2, 1)) + 1e-9 * G          # PSD perturbation + non-Hermitian dust
    E2, it2, st2 = eng.energies(A, h, r0=r0)
    ref = c_oracle.energy_batch(A, h, r0=r0, want_r=True, handoff=0, skip=SKIP0[4], period=PERIOD[4])
    assert np.all(st2 == 0) and np.all(ref['status'] == 0)
    assert (it2 == ref['iters']).mean() > 0.97
    assert np.abs(E2 - E).max() < E_TOL and np.abs(E2 - ref['E']).max() < E_TOL
    assert np.abs(eng.environments() - ref['r']).max() < R_TOL


def test_d4_scaled_tensors_inside_the_documented_range(engine_factory):
    """include/qmps_hip.h: tensors within a factor ~250 of an isometry keep T^(2^skip) inside the double range; the
    environment (a ray) does not depend on the scale; the energy tr(B r B^+)/tr r with B = A A scales as scale^4."""
    rng = np.random.default_rng(55)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 64))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    eng = engine_factory(4)
    select(eng, 'squaring0')
    E1, it1, st1 = eng.energies(A, h)
    r1 = eng.environments()
    for scale in (1e-2, 30.0, 200.0):
        E2, it2, st2 = eng.energies(scale * A, h)
        assert np.all(st2 == 0) and np.isfinite(E2).all()
        assert np.abs(eng.environments() - r1).max() < 1e-10
        assert np.abs(E2 / scale ** 4 - E1).max() < 1e-9


def test_d16_environment_krylov_fallback(engine_factory, monkeypatch):
    """D = 16 environment (qmps/tools.py:176-182 - the reference: TransferMatrix(A).eigs(), ARPACK): state tensors whose transfer map
    has |lambda_2| = 1 - 3e-3 .. 1 - 3e-6 (tests/overlap_cases.py) cost the power iteration ~ 30 / (1 - |lambda_2|) steps; the
    Krylov fall-back finishes every one of them within a few hundred map applications, energies within 1e-10 of the dense
    eigen-solve; QMPS_NO_KRYLOV runs the power iteration alone - same energies where it converges, status 1 at the same cap where not."""
    import overlap_cases as OC
    rng = np.random.default_rng(160)
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    A = np.stack([OC.slow_environment_tensor(rng, 16, t) for t in (0.3, 0.3, 0.1, 0.1, 0.03, 0.03, 0.01, 0.01)]
                 + list(O.unitary_to_tensor(O.haar_unitaries(rng, 32, 4))))
    ref = np.array([[O.energy_closed_form(a, hh) for hh in h] for a in A])      # dense eigen-solve of the environment
    eng = engine_factory(16, 64)
    E, it, st = eng.energies(A, h, max_iter=3000, tol=1e-13)
    assert np.all(st == 0), (st, it)
    assert it.max() <= 1000, it
    assert np.abs(E - ref).max() < E_TOL, np.abs(E - ref).max()
    r = eng.environments()
    for b in range(len(A)):
        rr = r[b] / np.trace(r[b])
        assert np.abs(rr - rr.conj().T).max() < 1e-12 and np.abs(O.apply_transfer(A[b], rr) - rr).max() < 1e-11
    # the exact in-kernel cost accumulation receives the late arrivals of the finishing pass
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    eng.launch(len(A), max_iter=3000, accumulate_cost=True)
    eng.cost_launch(len(A))
    assert np.abs(eng.get_cost() - ref.sum(axis=0)).max() < 1e-9
    monkeypatch.setenv('QMPS_NO_KRYLOV', '1')
    E2, it2, st2 = eng.energies(A, h, max_iter=3000, tol=1e-13)
    monkeypatch.delenv('QMPS_NO_KRYLOV')
    assert (st2 == 1).sum() >= 4 and np.all(it2[st2 == 1] == 3000)         # the slow ones exhaust the cap without it
    assert np.abs(E2 - ref)[st2 == 0].max() < E_TOL


def test_cost_accumulator_survives_a_change_of_the_number_of_terms():
    """Round 5 (profiles/experiments/r05/stress_api_state.py, a fuzzer of the stateful API): the in-kernel clear of a cost accumulator covers the CURRENT
    number of Hamiltonian terms; after two terms -> one term -> two terms a ring slot that counted as clean still held the arrivals of the second term:
    'cost accumulator: 46 of 23 waves arrived' on an accumulating launch.  qmps_set_hamiltonian now marks the whole ring dirty when the number changes."""
    from qmps_amd import EnergyEngine
    rng = np.random.default_rng(11)
    D, B = 4, 368
    eng = EnergyEngine(D, B)                      # (a context of its own: the ring's history is the point)
    U = O.haar_unitaries(rng, 2 * D, B)
    eng.set_tensors(np.stack([O.unitary_to_tensor(u) for u in U]))
    h1 = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})[None]
    h2 = np.stack([h1[0], O.hamiltonian_matrix({'XX': 1.0, 'YY': 1.0, 'ZZ': 0.5})])
    for h in (h2, h1, h2, h1, h2):
        eng.set_hamiltonian(h)
        for _ in range(12):                       # more than a lap of the eight-slot ring
            eng.launch(B, solver='direct', store_env=False, accumulate_cost=True)
            eng.cost_launch(B)
            E, _, st = eng.results(B)
            cost = eng.get_cost()
            assert np.all(st == 0) and np.abs(cost - E.sum(0)).max() < 1e-9 * np.abs(E).sum()
    eng.close()


@pytest.mark.parametrize('D,solver', [(2, 'plain'), (2, 'squaring'), (4, 'squaring'), (8, 'plain'), (16, 'plain')])
def test_warm_start_without_a_stored_environment_is_a_cold_start(D, solver):
    """Same fuzzer: `warm_start=True` reads the RESIDENT environments - but the flag that says there are any is per context, not per window.  A warm
    launch on a window that never stored one found zeros (or whatever the allocation held), divided by their trace and reported status != 0 for
    every evaluation.  Zeros / NaN are no guess: the power-iteration kernels now take their default start there (the direct D = 4 kernel's
    acceptance step had always rejected them)."""
    from qmps_amd import EnergyEngine
    rng = np.random.default_rng(12 + D)
    B = 96 if D == 16 else 300
    eng = EnergyEngine(D, 2 * B)
    A = np.stack([O.unitary_to_tensor(u) for u in O.haar_unitaries(rng, 2 * D, 2 * B)])
    h = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})[None]
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    eng.set_window(0)
    eng.launch(B, solver=solver, store_env=True)               # window 0 has environments now: the context's flag is up
    E0, _, st0 = eng.results(B)
    eng.set_window(B)
    eng.launch(B, solver=solver, store_env=True, warm_start=True)      # window 1 never stored any
    E1, _, st1 = eng.results(B)
    ref = EnergyEngine(D, B)
    ref.set_solver(solver)
    Er, _, str_ = ref.energies(A[B:], h)
    assert np.array_equal(st1 == 0, str_ == 0) and (st1 == 0).mean() > 0.95
    ok = st1 == 0
    assert np.abs(E1[ok] - Er[ok]).max() < 1e-10
    eng.close()
    ref.close()


def test_d2_squaring_chain_decides_at_the_end_of_its_budget(engine_factory):
    """D = 2 'squaring': the chain compares z_{2^m} with z_{2^(m-1)}, so under max_iter = 10 000 its last comparison is z_8192 against
    z_4096 and an evaluation the plain method finishes in 4 097 .. 10 000 steps used to end with status 1 although z_8192 IS its fixed
    point (found by the randomised stress of round 5).  Now one plain step on the last iterate decides (iterations 2^m + 1).
    Amplitude-damping channels in a random basis, decay rate gamma: the plain method needs ~30 / gamma steps."""
    rng = np.random.default_rng(77)

    def damp(gamma):
        W = O.haar_unitaries(rng, 2, 1)[0]
        K0 = np.diag([1, np.sqrt(1 - gamma)]).astype(complex)
        K1 = np.zeros((2, 2), complex)
        K1[0, 1] = np.sqrt(gamma)
        th = gamma / 2                  # a rotation in front of the damping: the fixed point is mixed (smallest eigenvalue ~0.13), not |0><0|
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]], dtype=complex)
        return np.stack([W @ K0 @ R @ W.conj().T, W @ K1 @ R @ W.conj().T])

    h = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
    eng = engine_factory(2)
    for solver, gammas in (('squaring0', [0.008, 0.007, 0.006]), ('squaring', [0.006, 0.005, 0.004])):
        select(eng, solver)
        ho = oracle_handoff(2, solver)
        A = np.stack([damp(gm) for gm in gammas] + [O.unitary_to_tensor(u) for u in O.haar_unitaries(rng, 4, 5)])
        E, it, st = eng.energies(A, h)
        r = eng.environments()
        ref = [O.env_power_iteration(a, handoff=ho, skip=SKIP0[2] if ho == 0 else 0) for a in A]
        assert np.all(st == 0), (solver, st, it)
        assert np.array_equal(it, [x[1] for x in ref]), (solver, it, [x[1] for x in ref])
        assert np.all(it[:len(gammas)] == ho + 8192 + 1)
        for b, a in enumerate(A):
            r_eig = O.env_dense_eig(a)[1]
            assert np.abs(r[b] - r_eig).max() < 1e-9
            assert np.abs(r[b] - ref[b][0]).max() < 1e-12
            assert abs(E[b, 0] - O.energy_closed_form(a, h, r_eig)) < 1e-9
    # a budget the chain cannot use: max_iter = 5 (skip schedule: 2^3 > 5) - the single plain step does not accept an unconverged iterate
    select(eng, 'squaring0')
    E, it, st = eng.energies(A[:3], h, max_iter=5)
    assert np.all(st == 1)
    eng.set_solver('direct')


def test_d8_environment_krylov_fallback(engine_factory, monkeypatch):
    """D = 8 (round 5): the power loop behind a direct solve that is not accepted - or the whole solve under 'squaring', which is the power method at
    D = 8 - hands evaluations with a long tail ahead to the Arnoldi kernel on the environment map, as D = 16 does.  Slow tensors
    (|lambda_2| = 1 - 3e-3 .. 1 - 3e-6): every one finished within a few hundred map applications, energies within 1e-10 of the dense
    eigen-solve, the in-kernel cost accumulation receives the late arrivals; 'plain' stays the plain iteration (a-13) and QMPS_NO_KRYLOV
    switches the hand-over off.  And the case the second stress campaign flagged: `ShallowCNOT3` at special angles - the unpivoted elimination
    meets a structural zero, gap 1.5e-3 - used to end with status 1 after 10 000 steps; its environment has rank 3: status 2, as the
    reference's Cholesky says."""
    import overlap_cases as OC
    rng = np.random.default_rng(80)
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    A = np.stack([OC.slow_environment_tensor(rng, 8, t) for t in (0.3, 0.3, 0.1, 0.1, 0.03, 0.03, 0.01, 0.01)]
                 + list(O.unitary_to_tensor(O.haar_unitaries(rng, 16, 4))))
    ref = np.array([[O.energy_closed_form(a, hh) for hh in h] for a in A])
    eng = engine_factory(8, 64)
    select(eng, 'squaring')
    E, it, st = eng.energies(A, h, max_iter=3000, tol=1e-13)
    assert np.all(st == 0), (st, it)
    assert it.max() <= 1000, it
    # (a fixed point accepted at ||T r - r|| < 1e-13 is within 1e-13 / gap of the true one: 3e-8 at gap 3e-6 - the energies of the two slowest within 1e-9)
    assert np.abs(E - ref)[:6].max() < E_TOL and np.abs(E - ref).max() < 1e-9, np.abs(E - ref).max()
    r = eng.environments()
    for b in range(len(A)):
        rr = r[b] / np.trace(r[b])
        assert np.abs(rr - rr.conj().T).max() < 1e-12 and np.abs(O.apply_transfer(A[b], rr) - rr).max() < 1e-11
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    eng.launch(len(A), max_iter=3000, solver='squaring')                     # the stepping entry point hands over on request only
    assert (eng.results(len(A))[2] == 1).sum() >= 4
    eng.launch(len(A), max_iter=3000, solver='squaring', accumulate_cost=True, krylov_fallback=True)
    eng.cost_launch(len(A))
    assert np.abs(eng.get_cost() - ref.sum(axis=0)).max() < 1e-9
    monkeypatch.setenv('QMPS_NO_KRYLOV', '1')
    E2, it2, st2 = eng.energies(A, h, max_iter=3000, tol=1e-13)
    monkeypatch.delenv('QMPS_NO_KRYLOV')
    assert (st2 == 1).sum() >= 4 and np.all(it2[st2 == 1] == 3000)
    assert np.abs(E2 - ref)[st2 == 0].max() < 1e-9
    select(eng, 'plain')
    E3, it3, st3 = eng.energies(A, h, max_iter=3000, tol=1e-13)
    assert np.array_equal(st3, st2) and np.array_equal(it3, it2)
    # the direct solver (default): accepted in one step for all of these
    eng.set_solver('direct')
    E4, it4, st4 = eng.energies(A, h, max_iter=3000, tol=1e-13)
    assert np.all(st4 == 0) and np.all(it4 <= 2) and np.abs(E4 - ref).max() < 1e-9
    p = np.array([0.06636560584974034, -np.pi / 2, 0.0, np.pi / 2, -np.pi / 2, 0.0, np.pi, np.pi / 2, np.pi])
    Es, its, sts = eng.energies_from_params(3, p[None], h[:1])
    a = O.unitary_to_tensor(O.shallow_cnot3_unitary(8, p))
    lam = np.linalg.eigvalsh(O.env_dense_eig(a)[1])
    assert lam[0] < 1e-12 < lam[-3]                      # rank 3
    assert sts[0] == 2 and its[0] < 2000, (sts, its)
    rr = eng.environments()[0]
    assert np.abs(O.apply_transfer(a, rr) - rr).max() < 1e-10
    assert abs(Es[0, 0] - O.energy_closed_form(a, h[0])) < 1e-9


def test_plain_power_iteration_d4_row_kernel_against_the_lane_kernel_and_the_oracle(c_oracle, engine_factory, monkeypatch):
    """Round 6: QMPS_ENV_POWER at D = 4 runs env_power_d4_kernel - a 16-lane DPP row per evaluation (the map as a real 16 x 16 matrix in
    registers, a power step = sixteen v_fmac_f64_dpp), persistent waves drawing evaluations from a work counter - then the energy-only
    kernel.  Against the lane-per-evaluation kernel of rounds 1-5 (QMPS_POWER_LANE=1) and the C oracle (plain power iteration): the same
    iterates (1e-12), the same iteration counts (+-1: a borderline test), statuses, energies (1e-10), density matrices, summed costs; batch
    sizes that are no multiple of four, a batch of one, an iteration cap that some evaluations hit (status 1 with the iterate of the last
    step), the caller's guess (symmetrised, trace-normalised; an unusable one = the default start), a resident-batch window, the in-kernel
    cost accumulator, a tensor that is not an isometry and the zero tensor (no fixed point: status 1, not a hang)."""
    rng = np.random.default_rng(606)
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    eng = engine_factory(4, 8192)

    def run(A, lane, **kw):
        if lane:
            monkeypatch.setenv('QMPS_POWER_LANE', '1')
        else:
            monkeypatch.delenv('QMPS_POWER_LANE', raising=False)
        select(eng, 'plain')
        E, it, st = eng.energies(A, h, **kw)
        out = (E.copy(), it.copy(), st.copy(), eng.environments().copy(), eng.rdm().copy(), eng.summed_cost().copy())
        monkeypatch.delenv('QMPS_POWER_LANE', raising=False)
        return out

    for B in (1, 3, 4, 5, 63, 1000, 8191):
        A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
        new, old = run(A, False, max_iter=4000), run(A, True, max_iter=4000)
        ref = c_oracle.energy_batch(A, h, max_iter=4000, want_r=True, want_rho=True)
        assert np.array_equal(new[2], old[2]) and np.all(new[2] == ref['status'])
        assert np.abs(new[1] - old[1]).max() <= 1 and np.abs(new[1] - ref['iters']).max() <= 1 and (new[1] == ref['iters']).mean() > 0.95
        ok = new[2] == 0
        assert np.abs(new[0] - old[0])[ok].max() < 1e-12 and np.abs(new[0] - ref['E'])[ok].max() < E_TOL
        assert np.abs(new[3] - old[3])[ok].max() < 1e-12 and np.abs(new[3] - ref['r'])[ok].max() < R_TOL
        assert np.abs(new[4] - ref['rho'])[ok].max() < R_TOL
        assert np.allclose(new[5], new[0].sum(0), rtol=0, atol=1e-9 * max(1, B))
    # an iteration cap inside the distribution (median ~95 steps): capped evaluations report status 1, iterations = cap, and the iterate of the last step
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 2000))
    new, old = run(A, False, max_iter=90), run(A, True, max_iter=90)
    assert np.array_equal(new[2], old[2]) and 0.2 < (new[2] == 1).mean() < 0.8 and np.all(new[1][new[2] == 1] == 90)
    assert np.abs(new[3] - old[3]).max() < 1e-12 and np.abs(new[0] - old[0]).max() < 1e-12
    # the caller's guess: a perturbed, scaled, slightly non-Hermitian fixed point converges in a few steps to the same environment; zeros / NaN = default start
    full = run(A, False)
    guess = 2.5 * full[3] + 1e-6 * (rng.standard_normal(full[3].shape) + 1j * rng.standard_normal(full[3].shape))
    guess[::7] = 0.0
    guess[3::7] = np.nan
    gn, go = run(A, False, r0=guess), run(A, True, r0=guess)
    assert np.all(gn[2] == 0) and np.abs(gn[1] - go[1]).max() <= 1 and np.abs(gn[3] - full[3]).max() < 1e-10 and np.abs(gn[0] - full[0]).max() < E_TOL
    usable = np.ones(len(A), dtype=bool)
    usable[::7] = False
    usable[3::7] = False
    assert gn[1][usable].mean() < 0.7 * full[1][usable].mean() and np.abs(gn[1][~usable] - full[1][~usable]).max() <= 1
    # resident batches: a window in the middle, the in-kernel cost accumulator
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    eng.set_window(512)
    eng.launch(700, max_iter=4000, tol=1e-13, solver='plain', accumulate_cost=True)
    eng.cost_launch(700)
    cost = eng.get_cost()
    Ew, itw, stw = eng.results(700)
    eng.set_window(0)
    assert np.abs(Ew - full[0][512:1212]).max() < 1e-12 and np.abs(itw - full[1][512:1212]).max() <= 1 and np.all(stw == 0)
    assert np.allclose(cost, Ew.sum(0), rtol=0, atol=1e-9 * 700)
    # not an isometry (the trace is not preserved: the normalisation carries it), and the zero tensor (no fixed point)
    odd = A[:6].copy()
    odd[0] *= 1.7
    odd[1] = 0.0
    on, oo = run(odd, False, max_iter=3000), run(odd, True, max_iter=3000)
    assert np.array_equal(on[2], oo[2]) and on[2][1] == 1 and on[1][1] == 3000 and on[2][0] == 0
    assert abs(on[1][0] - full[1][0]) <= 1 and np.abs(on[3][0] - full[3][0]).max() < 1e-12           # (a scaled tensor has the same normalised fixed point)
    keep = [0, 2, 3, 4, 5]
    assert np.abs(on[0][keep] - oo[0][keep]).max() < 1e-11
