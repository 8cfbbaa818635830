"""GPU parity tests of the DIRECT environment solver (QMPS_ENV_DIRECT, energy_direct_d4_kernel) through the C-ABI:
against the oracle's independent restatement (dense complex transfer matrix + pivoted LAPACK solve), the reference's
own route (dominant eigen-matrix by dense eig), the iterative solvers, and the CPU lock-step emulation of the
kernel's own source.  Tolerance: energies and environments within 1e-10 (BASELINE.json north_star)."""
import numpy as np
import pytest

from oracle import qmps_oracle as O
from tests import direct_emu as EMU

pytestmark = pytest.mark.gpu

E_TOL = 1e-10
R_TOL = 1e-10


def h3(rng):
    return np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}),
                     O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}),
                     rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))])


def test_direct_is_the_default_at_d4_and_matches_golden(golden, engine_factory):
    eng = engine_factory(4)
    eng.set_solver('direct')
    E, it, st = eng.energies(golden['ref_A_D4'], golden['ref_h_tfim'])
    assert np.all(st == 0) and np.all(it == 1)
    assert np.abs(E[:, 0] - golden['oracle_E_closed_D4']).max() < E_TOL
    assert np.abs(E[:, 0] - golden['oracle_E_statevec_D4']).max() < E_TOL
    assert np.abs(eng.environments() - golden['oracle_r_D4']).max() < R_TOL
    E2, _, _ = eng.energies(golden['U_D4'], golden['ref_h_tfim'], kind='unitary')
    assert np.array_equal(E, E2)
    from qmps_amd import EnergyEngine
    with EnergyEngine(4, 64) as fresh:          # a fresh context: the library default
        E3, it3, st3 = fresh.energies(golden['ref_A_D4'], golden['ref_h_tfim'])
    assert np.array_equal(E3, E) and np.all(it3 == 1)


@pytest.mark.parametrize('B', [1, 3, 15, 16, 17, 65, 1000, 5000])
def test_random_batches(B, c_oracle, engine_factory):
    rng = np.random.default_rng(4000 + B)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    h = h3(rng)
    eng = engine_factory(4)
    eng.set_solver('direct')
    E, it, st = eng.energies(A, h, max_iter=4000)
    assert np.all(st == 0) and np.all(it == 1)
    # the C oracle's plain power iteration (a different algorithm, same fixed point)
    ref = c_oracle.energy_batch(A, h, max_iter=4000, want_r=True, want_rho=True)
    ok = ref['status'] == 0
    assert ok.mean() > 0.9
    assert np.abs(E - ref['E'])[ok].max() < E_TOL
    r = eng.environments()
    assert np.abs(r - ref['r'])[ok].max() < R_TOL
    assert np.abs(eng.rdm() - ref['rho'])[ok].max() < R_TOL
    assert np.allclose(eng.summed_cost(), E.sum(0), rtol=0, atol=1e-9 * max(1, B))
    # the oracle's direct restatement and the reference's dense-eig route, item by item
    for b in range(0, B, max(1, B // 40)):
        rd, itd, std = O.env_direct(A[b], max_iter=4000)
        assert (itd, std) == (1, 0) and np.abs(r[b] - rd).max() < 1e-12
        assert np.abs(r[b] - O.env_dense_eig(A[b])[1]).max() < R_TOL
        for t in range(3):
            assert abs(E[b, t] - O.energy_closed_form(A[b], h[t], rd)) < 1e-12
    # the CPU lock-step emulation of the kernel's own source: same arithmetic up to FMA contraction / reciprocals
    emu = EMU.energies_d4(A, h, max_iter=4000)
    assert np.array_equal(emu['status'], st) and np.array_equal(emu['iters'], it)
    assert np.abs(E - emu['E']).max() < 1e-12 and np.abs(r - emu['r']).max() < 1e-12
    # asynchronous launch without the environment store: same energies, environments no longer resident
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    eng.launch(max_iter=4000, solver='direct', store_env=False)
    E2, it2, st2 = eng.results()
    assert np.array_equal(E2, E) and np.array_equal(st2, st)
    from qmps_amd._lib import QmpsError
    with pytest.raises(QmpsError):
        eng.environments()
    eng.cost_launch()
    assert np.allclose(eng.get_cost(), E.sum(0), rtol=0, atol=1e-9 * max(1, B))


def test_agrees_with_the_iterative_solvers(engine_factory):
    rng = np.random.default_rng(77)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 3000))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 0.7})
    eng = engine_factory(4)
    eng.set_solver('direct')
    E, it, st = eng.energies(A, h)
    r = eng.environments()
    eng.set_solver('squaring', handoff=0)
    E2, it2, st2 = eng.energies(A, h)
    r2 = eng.environments()
    eng.set_solver('plain')
    E3, it3, st3 = eng.energies(A, h)
    ok = (st == 0) & (st2 == 0) & (st3 == 0)
    assert ok.mean() > 0.99
    assert np.abs(E - E2)[ok].max() < E_TOL and np.abs(E - E3)[ok].max() < E_TOL
    assert np.abs(r - r2)[ok].max() < R_TOL
    eng.set_solver('direct')


def test_fallback_paths(engine_factory):
    """Evaluations the direct solve cannot accept continue inside the same launch with the power method 2^m steps at
    a time: tensors that are not isometries (dominant eigenvalue != 1), a degenerate product state, the iteration cap."""
    rng = np.random.default_rng(78)
    B = 300
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    mix = A.copy()
    bad = rng.random(B) < 0.3                     # a minority per wave: quads that fall back next to quads that do not
    mix[bad] = mix[bad] * rng.uniform(0.6, 1.5, size=(bad.sum(), 1, 1, 1)) \
        + 0.05 * (rng.standard_normal(mix[bad].shape) + 1j * rng.standard_normal(mix[bad].shape))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    eng = engine_factory(4)
    eng.set_solver('direct')
    r, it, st = eng.env_batch(mix, max_iter=100000)
    assert np.all(st == 0) and np.all(it[~bad] == 1) and np.all(it[bad] > 1) and np.isfinite(r).all()
    for b in range(B):
        rd, itd, std = O.env_direct(mix[b], max_iter=100000)
        assert std == 0 and itd == it[b]
        assert np.abs(r[b] - rd).max() < 1e-11
    for b in np.flatnonzero(bad)[:30]:
        assert np.abs(r[b] - O.env_dense_eig(mix[b])[1]).max() < R_TOL
    emu = EMU.energies_d4(mix, h, max_iter=100000)
    assert np.array_equal(emu['iters'], it) and np.abs(emu['r'] - r).max() < 1e-11
    # the cap is reported, not hidden
    E, it, st = eng.energies(mix[bad][:20], h, max_iter=3)
    assert np.all(st == 1) and np.all(it == 3)
    E, it, st = eng.energies(mix[bad][:20], h, max_iter=1)
    assert np.all(st == 1) and np.all(it == 1)
    # product state: rank-one environment, not positive definite (the reference's LinAlgError branch)
    U = np.eye(8, dtype=complex)[None]
    E, it, st = eng.energies(U, h, kind='unitary')
    assert st[0] == 2 and abs(E[0, 0] + 1.0) < 1e-12
    # empty batch
    E, it, st = eng.energies(np.zeros((0, 2, 4, 4), complex), h)
    assert E.shape == (0, 1)


def test_direct_flag_errors(engine_factory):
    from qmps_amd._lib import QmpsError
    eng8 = engine_factory(8, 64)
    rng = np.random.default_rng(5)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 16, 8))
    eng8.set_tensors(A)
    eng8.set_hamiltonian(O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
    with pytest.raises(QmpsError):
        eng8.launch(solver='direct', store_env=False)      # the environment store can only be skipped by the fused D = 4 kernel
    eng2 = engine_factory(2, 64)
    A2 = O.unitary_to_tensor(O.haar_unitaries(rng, 4, 8))
    E, it, st = eng2.energies(A2, O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
    assert np.all(st == 0) and np.all(it == 1)              # D = 2: 4 x 4 solve in the lane, accepted by one power step
    eng2.launch(solver='squaring')
    E2, it2, st2 = eng2.results()
    assert np.all(st2 == 0) and np.all(it2 > 1) and np.abs(E - E2).max() < E_TOL
    eng16 = engine_factory(16, 64)
    A16 = O.unitary_to_tensor(O.haar_unitaries(rng, 32, 4))
    E16, it16, st16 = eng16.energies(A16, O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
    eng16.launch(solver='direct')                           # documented: D = 16 iterates
    _, it16b, st16b = eng16.results()
    assert np.all(st16b == 0) and np.all(it16b > 1)


@pytest.mark.parametrize('B', [1, 37, 768, 3000])
def test_direct_d8(B, c_oracle, engine_factory):
    """D = 8 (BASELINE.json configs[3]): wave-per-evaluation direct solve of the real 64 x 64 system, accepted by the
    first power step of the iteration kernel."""
    rng = np.random.default_rng(8000 + B)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 16, B))
    h = np.stack([O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}), O.hamiltonian_matrix({'ZZ': -1, 'X': 1})])
    eng = engine_factory(8, 4096)
    eng.set_solver('direct')
    E, it, st = eng.energies(A, h)
    assert np.all(st == 0) and np.all(it == 1)
    ref = c_oracle.energy_batch(A, h, want_r=True, want_rho=True)
    ok = ref['status'] == 0
    assert ok.mean() > 0.9
    assert np.abs(E - ref['E'])[ok].max() < E_TOL
    r = eng.environments()
    assert np.abs(r - ref['r'])[ok].max() < R_TOL
    assert np.abs(eng.rdm() - ref['rho'])[ok].max() < R_TOL
    for b in range(0, B, max(1, B // 10)):
        rd, itd, std = O.env_direct(A[b])
        assert (itd, std) == (1, 0) and np.abs(r[b] - rd).max() < 1e-12
        assert np.abs(r[b] - O.env_dense_eig(A[b])[1]).max() < R_TOL
    # the plain power iteration of the same kernel agrees
    eng.set_solver('plain')
    E2, it2, st2 = eng.energies(A, h)
    both = (st2 == 0)
    assert np.abs(E - E2)[both].max() < E_TOL and it2.mean() > 10
    eng.set_solver('direct')
    if B == 37:
        # fall-back: tensors that are not isometries converge by the plain iteration from the (wrong) direct start;
        # a product state (zero pivot in the elimination) restarts from 1/8
        bad = A * rng.uniform(0.7, 1.4, size=(B, 1, 1, 1)) + 0.03 * (rng.standard_normal(A.shape) + 1j * rng.standard_normal(A.shape))
        rb, itb, stb = eng.env_batch(bad, max_iter=20000)
        assert np.all(stb == 0) and np.all(itb > 1) and np.isfinite(rb).all()
        for k in range(0, B, 6):
            assert np.abs(rb[k] - O.env_dense_eig(bad[k])[1]).max() < R_TOL
        U = np.eye(16, dtype=complex)[None]
        Ep, itp, stp = eng.energies(U, h[1], kind='unitary')
        assert stp[0] in (1, 2) and abs(Ep[0, 0] + 1.0) < 1e-12


@pytest.mark.parametrize('with_comm', [False, True])
def test_in_kernel_cost_accumulation(with_comm):
    """QMPS_FLAG_ACCUMULATE_COST: the fused kernel sums the batch itself (fixed-point integers: exact, independent of
    the order the waves finish in); qmps_cost_launch then launches nothing.  Every step of a long sequence (the ring of
    accumulators wraps several times), grouped exchanges, partly filled groups, a dropped accumulation, the overflow
    path for tensors that are not isometries, and bit-identical repeats."""
    from qmps_amd import EnergyEngine
    from qmps_amd._lib import QmpsError
    rng = np.random.default_rng(91)
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    batches = [O.unitary_to_tensor(O.haar_unitaries(rng, 8, 777)) for _ in range(5)]
    with EnergyEngine(4, 5 * 777) as eng:
        if with_comm:
            eng.comm_init(EnergyEngine.comm_unique_id(), 0, 1)
        sums = []
        for A in batches:
            E, _, st = eng.energies(A, h)
            assert np.all(st == 0)
            sums.append(E.sum(0))
        eng.set_tensors(np.concatenate(batches))
        eng.set_hamiltonian(h)
        for period in (1, 3):
            eng.set_exchange_period(period)
            for k in range(23):
                eng.set_window((k % 5) * 777)
                eng.launch(777, accumulate_cost=True, store_env=(k % 2 == 0))
                eng.cost_launch(777)
                if k % 4 == 3 or k == 22:
                    c = eng.get_cost()
                    assert np.abs(c - sums[k % 5]).max() < 1e-10, (period, k)
        eng.set_exchange_period(1)
        # many laps of the accumulator ring with the host running far ahead of the GPU (no read-back in between): a finish
        # kernel must never mistake the previous lap's words (full arrival count) for this step's
        for k in range(150):
            eng.set_window((k % 5) * 777)
            eng.launch(777, accumulate_cost=True, store_env=False)
            eng.cost_launch(777)
            if k % 37 == 36 or k == 149:
                assert np.abs(eng.get_cost() - sums[k % 5]).max() < 1e-10, k
        # the fixed-point sum does not depend on the order the waves finish in: bit-identical repeats
        seen = set()
        for _ in range(5):
            eng.set_window(0)
            eng.launch(777, accumulate_cost=True)
            eng.cost_launch(777)
            seen.add(eng.get_cost().tobytes())
        assert len(seen) == 1
        # contract: an accumulating launch must be consumed before the next one
        eng.launch(777, accumulate_cost=True)
        with pytest.raises(QmpsError):
            eng.launch(777, accumulate_cost=True)
        eng.launch(777)                       # an ordinary launch drops the pending accumulation ...
        eng.cost_launch(777)                  # ... and this is the two-kernel path again
        assert np.abs(eng.get_cost() - sums[0]).max() < 1e-10
        eng.launch(777, accumulate_cost=True)
        eng.cost_launch(777)
        assert np.abs(eng.get_cost() - sums[0]).max() < 1e-10
        # tensors far from isometries: energies beyond 16 ||h||_F per wave bypass the fixed-point sum (double adds)
        big = 30.0 * batches[0]
        Eb, _, stb = eng.energies(big, h, max_iter=100000)
        assert np.all(stb == 0) and np.abs(Eb).max() > 1e4
        eng.launch(777, max_iter=100000, accumulate_cost=True)
        eng.cost_launch(777)
        assert np.allclose(eng.get_cost(), Eb.sum(0), rtol=1e-12, atol=0)
        if with_comm:
            eng.comm_destroy()


@pytest.mark.parametrize('D,B', [(2, 4096), (2, 100), (4, 1000), (8, 768), (16, 40)])
def test_in_kernel_cost_on_the_other_kernels(D, B, engine_factory):
    """QMPS_FLAG_ACCUMULATE_COST on the lane kernels (D = 2; D = 4 plain power iteration), the D = 8 block kernel behind the
    direct solve and the D = 16 MFMA kernel: the exact in-kernel sum equals the host sum of the energies; the two-kernel
    D = 4 squaring path refuses the flag."""
    from qmps_amd._lib import QmpsError
    rng = np.random.default_rng(50 + D + B)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, B))
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    eng = engine_factory(D, 4096)
    for solver in (['plain', 'direct'] if D == 4 else ['direct', 'plain']):
        eng.set_solver(solver)
        E, _, st = eng.energies(A, h)
        eng.set_tensors(A)
        eng.set_hamiltonian(h)
        for rep in range(11):                       # laps of the accumulator ring
            eng.launch(B, solver=solver, accumulate_cost=True)
            eng.cost_launch(B)
        c = eng.get_cost()
        E2, _, _ = eng.results(B)
        assert np.array_equal(E2, E) and np.abs(c - E.sum(0)).max() < 1e-9 * max(1, B / 100), (D, solver)
    if D == 4:
        with pytest.raises(QmpsError):
            eng.launch(B, solver='squaring', accumulate_cost=True)
    eng.set_solver('direct')


@pytest.mark.parametrize('kind,P', [('cnot', 4), ('cnot', 6), ('qaoa', 4), ('cnot3', 6)])
def test_ansatz_fused_in_front_of_the_solve(kind, P, engine_factory):
    """at D = 4 the direct kernel builds the state tensor in LDS from the ansatz parameters (8 P bytes per
    evaluation instead of 512) - same energies / environments as the tensors of the stand-alone builder, for whole
    batches, windows into the resident parameters and ragged batch sizes; against the oracle's circuits."""
    from qmps_amd import _lib
    code = {'cnot': _lib.ANSATZ_SHALLOW_CNOT, 'qaoa': _lib.ANSATZ_SHALLOW_QAOA, 'cnot3': _lib.ANSATZ_SHALLOW_CNOT3}[kind]
    build = {'cnot': O.shallow_cnot_unitary, 'qaoa': O.shallow_qaoa_unitary, 'cnot3': O.shallow_cnot3_unitary}[kind]
    rng = np.random.default_rng(31)
    B = 1000 + 7                                     # not a multiple of 16: the last wave is ragged
    prm = rng.standard_normal((B, P))
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    eng = engine_factory(4, 2048)
    eng.set_hamiltonian(h)
    eng.set_ansatz_params(code, prm)
    eng.launch(B, solver='direct', store_env=True)            # fused: no tensor has been built yet
    E1, it1, st1 = eng.results(B)
    r1 = eng.environments(B)
    A = eng.tensors(B)                                         # materialised on demand by the stand-alone builder
    for b in range(0, B, 83):
        assert np.abs(A[b] - O.unitary_to_tensor(build(4, prm[b])[None])[0]).max() < 1e-14
    # a window into the resident parameters
    eng.set_window(496)
    eng.launch(300, solver='direct', store_env=True)
    Ew, itw, stw = eng.results(300)
    assert np.array_equal(Ew, E1[496:796]) and np.array_equal(stw, st1[496:796])
    eng.set_window(0)
    # the same tensors through the HBM path
    eng.set_tensors(A)
    eng.launch(B, solver='direct', store_env=True)
    E2, it2, st2 = eng.results(B)
    r2 = eng.environments(B)
    ok = (st1 == 0) & (st2 == 0)
    assert np.array_equal(st1, st2) and ok.mean() > 0.9
    assert np.abs(E1 - E2)[ok].max() < 1e-12 and np.abs(r1 - r2)[ok].max() < 1e-12
    for b in np.flatnonzero(ok)[::97]:
        for t in range(2):
            assert abs(E1[b, t] - O.energy_closed_form(A[b], h[t])) < 1e-10


def test_warm_start_from_resident_environments(engine_factory):
    """environments carried over between evaluations.  An evaluation whose resident environment passes the
    acceptance test (one power step moves it by less than tol) skips the matrix build and the elimination (iterations 1);
    a stale environment is rejected and the evaluation solved from scratch (iterations 2) - same energies either way."""
    rng = np.random.default_rng(41)
    B = 2000 + 5
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    A2 = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    eng = engine_factory(4, 4096)
    eng.set_hamiltonian(h)
    eng.set_tensors(A)
    eng.launch(B, solver='direct', store_env=True)
    E0, it0, st0 = eng.results(B)
    r0 = eng.environments(B)
    assert np.all(st0 == 0) and np.all(it0 == 1)
    # (1) unchanged tensors, resident environments: every evaluation takes the fast path
    eng.launch(B, solver='direct', store_env=True, warm_start=True)
    E1, it1, st1 = eng.results(B)
    assert np.all(st1 == 0) and np.all(it1 == 1)
    assert np.abs(E1 - E0).max() < 1e-12 and np.abs(eng.environments(B) - r0).max() < 1e-12
    # ... also without storing anything, and summed inside the kernel
    eng.launch(B, solver='direct', store_env=False, accumulate_cost=True, warm_start=True)
    eng.cost_launch(B)
    acc = eng.get_cost()
    E1b, _, _ = eng.results(B)
    assert np.abs(E1b - E1).max() < 1e-13 and np.allclose(acc, E1b.sum(0), rtol=0, atol=1e-9)
    # (2) half of the tensors replaced: their resident environments are stale -> solved from scratch, the others accepted
    mix = A.copy()
    mix[::2] = A2[::2]
    eng.set_tensors(mix)            # (keeps the resident environments: they are the guess)
    eng.set_env_guess(r0)
    eng.launch(B, solver='direct', store_env=True)          # a guess set by the host is used the same way
    E2, it2, st2 = eng.results(B)
    assert np.all(st2 == 0) and np.all(it2[1::2] == 1) and np.all(it2[::2] == 2)
    eng.set_tensors(mix)
    eng.launch(B, solver='direct', store_env=True)
    Ec, itc, stc = eng.results(B)
    assert np.all(itc == 1) and np.abs(E2 - Ec).max() < 1e-12
    # (3) a guess of a different trace / not Hermitian-exact is normalised; garbage is rejected, never believed
    eng.set_tensors(A)
    junk = r0 * 3.7
    junk[5] = np.eye(4) * 0.25
    junk[6] = np.nan
    eng.set_env_guess(junk)
    eng.launch(B, solver='direct', store_env=True)
    E3, it3, st3 = eng.results(B)
    assert np.all(st3 == 0) and it3[5] == 2 and it3[6] == 2 and np.all(np.delete(it3, [5, 6]) == 1)
    assert np.abs(E3 - E0).max() < 1e-12
    # (4) the flag without resident environments is an error, not a silent cold start
    eng.set_tensors(A)
    eng.launch(B, solver='direct', store_env=False)
    with pytest.raises(Exception):
        eng.launch(B, solver='direct', warm_start=True)


@pytest.mark.parametrize('D', [4, 8])
def test_structured_angles_stress(D, engine_factory):
    """ShallowCNOT depth-2 parameters on the grid of multiples of pi/4 (product states, degenerate and unimodular transfer
    spectra: everything the rare paths exist for), tiny perturbations of those points, and random angles.  Wherever the
    kernel says OK the environment IS a fixed point and - if the dominant eigenvalue is separated - the energy is the
    dense-eigensolver oracle's; nothing non-finite is ever reported as OK; a separated spectrum is never 'not converged'."""
    import itertools
    from qmps_amd import _lib
    rng = np.random.default_rng(5)
    grid = np.array([0, np.pi / 4, np.pi / 2, np.pi, -np.pi / 2, 3 * np.pi / 4])
    P = {4: 4, 8: 6}[D]
    base = np.array(list(itertools.product(grid, repeat=P)))[::{4: 1, 8: 59}[D]]
    prm = np.concatenate([base, base + 1e-9 * rng.standard_normal(base.shape), base + 1e-5 * rng.standard_normal(base.shape),
                          base + 1e-2 * rng.standard_normal(base.shape), 2 * rng.standard_normal(({4: 3000, 8: 800}[D], P))])
    B = len(prm)
    N = D * D
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    eng = engine_factory(D, 8192)
    eng.set_hamiltonian(h)
    eng.set_ansatz_params(_lib.ANSATZ_SHALLOW_CNOT, prm)
    eng.launch(B, max_iter=100000 if D == 4 else 20000, solver='direct', store_env=True)
    E, it, st = eng.results(B)
    r = eng.environments(B)
    A = eng.tensors(B)
    ok = st == 0
    assert np.isfinite(E[ok]).all() and np.isfinite(r[ok]).all() and ok.mean() > 0.8
    Tr = np.einsum('bsij,bjk,bslk->bil', A, r, A.conj())
    Tr /= np.trace(Tr, axis1=1, axis2=2)[:, None, None]
    assert np.abs(Tr - r)[ok].max() < 1e-11
    T = np.einsum('bsij,bskl->bikjl', A, A.conj()).reshape(B, N, N)
    w = np.sort(np.abs(np.linalg.eigvals(T)), axis=1)[:, ::-1]
    gap = w[:, 0] - w[:, 1]
    assert not np.any((st == 1) & (gap > 1e-3))
    sel = np.flatnonzero(ok & (gap > 1e-6))[::{4: 7, 8: 23}[D]]
    for b in sel:
        for t in range(2):
            assert abs(E[b, t] - O.energy_closed_form(A[b], h[t])) < 1e-9 * max(1.0, 1e-6 / gap[b])
    # the fall-back is taken (singular systems) and reported: D = 4: iterations 1 + 2^m
    fb = it > 1
    assert fb.any() and (D != 4 or np.all(np.log2(it[fb] - 1) % 1 == 0))


def test_warm_start_with_the_fused_ansatz(engine_factory):
    """Both prologues together: tensors built from ansatz parameters in LDS AND resident environments accepted as they are."""
    from qmps_amd import _lib
    rng = np.random.default_rng(43)
    B = 777
    prm = rng.standard_normal((B, 4))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    eng = engine_factory(4, 2048)
    eng.set_hamiltonian(h)
    for code in (_lib.ANSATZ_SHALLOW_CNOT, _lib.ANSATZ_SHALLOW_QAOA):
        eng.set_ansatz_params(code, prm)
        eng.launch(B, solver='direct', store_env=True)
        E0, it0, st0 = eng.results(B)
        eng.launch(B, solver='direct', store_env=True, warm_start=True)
        E1, it1, st1 = eng.results(B)
        ok = (st0 == 0) & (st1 == 0)
        # (an environment with a zero eigenvalue to rounding may flip between OK and NOT_PD from one power step to the next)
        assert (st0 == st1).mean() > 0.99 and ok.mean() > 0.9
        assert np.all(it1[ok & (it0 == 1)] == 1) and np.abs(E1 - E0)[ok].max() < 1e-12
        # new parameters, stale environments: rejected, solved from scratch
        prm2 = prm + 0.3
        eng.set_ansatz_params(code, prm2)
        eng.launch(B, solver='direct', store_env=True)
        Ec, itc, stc = eng.results(B)
        eng.set_ansatz_params(code, prm)
        eng.launch(B, solver='direct', store_env=True)
        r_old = eng.environments(B)
        eng.set_ansatz_params(code, prm2)
        eng.set_env_guess(r_old)
        eng.launch(B, solver='direct', store_env=True)
        Ew, itw, stw = eng.results(B)
        both = (stc == 0) & (stw == 0)
        assert np.abs(Ew - Ec)[both].max() < 1e-12 and np.all(itw[both & (itc == 1)] == 2)


@pytest.mark.parametrize('n_terms', [1, 2, 3])
def test_energy_only_contraction_chain_d4(n_terms, engine_factory):
    """qmps_energy_only_launch at D = 4 (the A - Abar - h - A - Abar chain with resident environments): the density-matrix-
    free route (one or two terms) and the rho route (more terms, the density matrix itself) against the oracle's closed
    form with the SAME environments; ragged batch, a window, non-Hermitian h, unnormalised environments."""
    rng = np.random.default_rng(51)
    B = 3000 + 11
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    hs = [O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)),
          O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})][:n_terms]
    h = np.stack(hs)
    eng = engine_factory(4, 8192)
    eng.set_hamiltonian(h)
    eng.set_tensors(A)
    eng.launch(B, solver='direct', store_env=True)
    E0, _, st = eng.results(B)
    r = eng.environments(B)
    assert np.all(st == 0)
    # resident environments replaced by scaled copies: the pass normalises by the trace
    eng.set_env_guess(r * rng.uniform(0.5, 3.0, size=(B, 1, 1)))
    eng.launch_energy_only(B)
    E1, _, _ = eng.results(B)
    assert np.abs(E1 - E0).max() < 1e-12
    for b in range(0, B, 211):
        for t in range(n_terms):
            want = np.einsum('st,ts->', h[t], O.two_site_rdm(A[b], r[b]))
            assert abs(E1[b, t] - want.real) < 1e-12
    # a window
    eng.set_window(1024)
    eng.launch_energy_only(500)
    Ew, _, _ = eng.results(500)
    assert np.array_equal(Ew, E1[1024:1524])
    eng.set_window(0)
    # the density matrix (rho route whatever the number of terms)
    rho = eng.rdm(B)
    for b in range(0, B, 307):
        assert np.abs(rho[b] - O.two_site_rdm(A[b], r[b])).max() < 1e-13


def test_direct_d2(c_oracle, engine_factory):
    """D = 2 (BASELINE.json configs[0], [1]): the 4 x 4 fixed-point solve in the lane.  Iteration counts of the power
    method are heavy-tailed at D = 2 (p99.9 ~ 3000 plain steps, some Haar items do not converge in 10 000): the direct
    solve accepts every one of them in one step.  Non-isometric tensors and product states take the squaring path."""
    rng = np.random.default_rng(99)
    B = 20000
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 4, B))
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    eng = engine_factory(2, 32768)
    eng.set_solver('direct')
    E, it, st = eng.energies(A, h)
    r = eng.environments()
    assert np.all(st == 0) and np.all(it == 1)
    for b in range(0, B, 331):
        rd, itd, std = O.env_direct(A[b])
        assert (itd, std) == (1, 0) and np.abs(r[b] - rd).max() < 1e-12
        assert np.abs(r[b] - O.env_dense_eig(A[b])[1]).max() < 1e-10
        for t in range(2):
            assert abs(E[b, t] - O.energy_closed_form(A[b], h[t], rd)) < 1e-12
    eng.set_solver('squaring')
    E2, it2, st2 = eng.energies(A, h)
    both = st2 == 0
    assert both.mean() > 0.99 and np.abs(E - E2)[both].max() < 1e-10 and it2.max() > 100
    eng.set_solver('direct')
    # tensors that are not isometries: rejected, the squaring tail takes over (iterations > 1), same fixed point as dense eig
    bad = A[:200] * rng.uniform(0.6, 1.5, size=(200, 1, 1, 1)) + 0.05 * (rng.standard_normal((200, 2, 2, 2)) + 1j * rng.standard_normal((200, 2, 2, 2)))
    rb, itb, stb = eng.env_batch(bad, max_iter=100000)
    assert np.all(stb == 0) and np.all(itb > 1)
    for b in range(0, 200, 9):
        assert np.abs(rb[b] - O.env_dense_eig(bad[b])[1]).max() < 1e-10
    # product state |00>: rank-one environment -> NOT_PD, energy -1 (ZZ = -1 on |00>)
    Ep, itp, stp = eng.energies(np.eye(4, dtype=complex)[None], h[0], kind='unitary')
    assert stp[0] == 2 and abs(Ep[0, 0] + 1.0) < 1e-12
    # the device-resident rotosolve (every sweep of every restart in one kernel) runs the same solve
    from qmps_amd import _lib
    P0 = rng.standard_normal((64, 4))
    eng.set_hamiltonian(h)
    hist, pf = eng.rotosolve(_lib.ANSATZ_SHALLOW_CNOT, P0, 3)
    for b in np.flatnonzero(~np.isnan(hist[-1]))[::7]:
        Ab = O.unitary_to_tensor(O.shallow_cnot_unitary(2, pf[b])[None])[0]
        assert abs(hist[-1][b] - sum(O.energy_closed_form(Ab, h[t]) for t in range(2))) < 1e-9


def test_fallback_chain_decides_at_the_end_of_its_budget(engine_factory):
    """The fused D = 4 kernel's fall-back compares z_(2^m) with z_(2^(m-1)): under max_iter = 10 000 a tensor whose solve is not accepted (here: not
    an isometry, A -> 1.3 A) and whose plain iteration needs 4 097 .. 9 998 steps used to end with status 1 at 1 + 8 192 although z_8192 is its
    fixed point (fourth stress campaign of round 5; the D = 2 chain had the same effect).  One plain step on the last iterate now decides:
    iterations 1 + 8 192 + 1, the kernel, its CPU emulation (tests/csrc/direct_emu.cpp) and the oracle alike."""
    import overlap_cases as OC
    rng = np.random.default_rng(44)
    A = np.stack([OC.slow_environment_tensor(rng, 4, t) * 1.3 for t in (0.22, 0.22, 0.25, 0.25, 0.28, 0.28, 0.3, 0.3)])
    h = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
    ref = [O.env_direct(a) for a in A]
    its = np.array([x[1] for x in ref])
    assert (its == 8194).sum() >= 5 and all(x[2] == 0 for x in ref)
    eng = engine_factory(4)
    eng.set_solver('direct')
    E, it, st = eng.energies(A, h)
    assert np.all(st == 0) and np.array_equal(it, its), (st, it, its)
    r = eng.environments()
    for b, a in enumerate(A):
        rr = r[b] / np.trace(r[b])
        assert np.abs(rr - ref[b][0]).max() < 1e-11
        assert abs(E[b, 0] - O.energy_closed_form(a, h, ref[b][0])) < 1e-9 * 1.3 ** 4
    emu = EMU.energies_d4(A, h)
    assert np.array_equal(emu['iters'], it) and np.all(emu['status'] == 0) and np.abs(emu['E'][:, 0] - E[:, 0]).max() < 1e-11
    E2, it2, st2 = eng.energies(A, h, max_iter=8193)          # no room for the extra step: reported, not hidden
    assert np.all(st2[its == 8194] == 1)
