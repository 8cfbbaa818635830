"""GPU: the brick-wall kernels (qmps_bw_*), through the reference-named classes of qmps_amd.new_tdvp, against
outputs of the reference's own code (tests/golden/brickwall_golden.npz), its exact known answers, and the oracle
on larger seeded batches."""
import os

import numpy as np
import pytest

from oracle import brickwall_oracle as BW
from qmps_amd import new_tdvp as NT

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bw():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'brickwall_golden.npz'))


def dag(u):
    return np.conj(np.swapaxes(u, -1, -2))


def test_reference_outputs(bw):
    oc, re_, le, mo = NT.OverlapCalculator(), NT.RightEnvironment(), NT.LeftEnvironment(), NT.ManifoldOverlap()
    U1, U2, U1p, U2p = bw['U1'], bw['U2'], bw['U1p'], bw['U2p']
    assert np.abs(oc.qbt2_exp_val(U1, U2, bw['O2']) - bw['ref_qbt2']).max() < 1e-12
    assert np.abs(oc.mqbt2_exp_val(U1, U2, bw['O2']) - bw['ref_mqbt2']).max() < 1e-12
    assert np.abs(oc.qbt4_exp_val(U1, U2, bw['O4']) - bw['ref_qbt4']).max() < 1e-11
    assert np.abs(oc.mqbt4_exp_val(U1, U2, bw['O4']) - bw['ref_mqbt4']).max() < 1e-11
    # single-item calls with the reference's (2,2,2,2) tensor form
    k = 2
    assert abs(oc.expectation_value(U1[k].reshape(2, 2, 2, 2), U2[k].reshape(2, 2, 2, 2), bw['O2'][k].reshape(2, 2, 2, 2))
               - bw['ref_qbt2'][k]) < 1e-12
    assert np.abs(re_.exact_environment_circuit(U1, U2, U1p, U2p) - bw['ref_RE_matrix']).max() < 1e-13
    assert np.abs(le.exact_environment_circuit(U1, U2, U1p, U2p) - bw['ref_LE_matrix']).max() < 1e-13
    assert np.abs(re_.circuit(U1, U2, U1p, U2p, bw['M']) - bw['ref_RE_circuit']).max() < 1e-13
    for env, eta_ref, vec_ref in ((re_, bw['ref_RE_eta_dag'], bw['ref_RE_vec_dag']), (le, bw['ref_LE_eta_dag'], bw['ref_LE_vec_dag'])):
        eta, vec = env.exact_environment(U1, U2, dag(U1), dag(U2))
        assert np.abs(eta - eta_ref).max() < 1e-11
        for a, b in zip(vec, vec_ref):
            assert abs(abs(np.vdot(b.reshape(-1), a.reshape(-1))) - 1) < 1e-9
    ov = mo.circuit(U1, U2, U1p, U2p, bw['M'], bw['Ml'], bw['W'])
    assert np.abs(ov - bw['ref_manifold']).max() < 1e-12 and np.abs(ov - bw['ref_mmanifold']).max() < 1e-12


def test_reference_known_answers():
    """new_tdvp/testTDVPStripped.py:71-170."""
    I, Z, X = np.eye(2), np.diag([1.0, -1.0]), np.array([[0, 1.0], [1.0, 0]])
    Had = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    kr = np.kron
    oc = NT.OverlapCalculator()
    t4 = lambda m: m.reshape(2, 2, 2, 2)                                              # noqa: E731
    assert np.isclose(oc.expectation_value(t4(kr(I, I)), t4(kr(I, I)), t4(kr(Z, Z))), 1)
    assert np.isclose(oc.expectation_value(t4(kr(X, X)), t4(kr(I, I)), t4(kr(Z, Z))), 1)
    assert np.isclose(oc.expectation_value(t4(kr(X, X)), t4(kr(I, I)), t4(kr(I, Z))), -1)
    assert np.isclose(oc.expectation_value(t4(kr(Had, Had)), t4(kr(I, I)), t4(kr(X, X))), 1)
    assert np.isclose(oc.expectation_value(t4(kr(Had, Had)), t4(kr(X, X)), t4(kr(X, I))), -1)
    Z4 = kr(kr(Z, Z), kr(Z, Z)).reshape((2,) * 8)
    assert np.isclose(oc.qbt4_exp_val(t4(kr(I, I)), t4(kr(I, I)), Z4), 1)
    assert np.isclose(oc.qbt4_exp_val(t4(kr(X, X)), t4(kr(I, I)), kr(kr(I, Z), kr(Z, Z)).reshape((2,) * 8)), -1)
    assert np.isclose(oc.qbt4_exp_val(t4(kr(Had, Had)), t4(kr(X, X)), kr(kr(X, I), kr(I, I)).reshape((2,) * 8)), -1)
    RE = NT.RightEnvironment()
    U1, U2 = kr(X, X), kr(I, I)
    assert np.allclose(RE.circuit(t4(U1), t4(U2), t4(U1.conj().T), t4(U2.conj().T), Z), I)
    assert np.allclose(RE.exact_environment_circuit(t4(U1), t4(U2), t4(U1.conj().T), t4(U2.conj().T)),
                       [[1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [1, 0, 0, 0]])
    eta, vec = RE.exact_environment(t4(U1), t4(U2), t4(U1.conj().T), t4(U2.conj().T))
    assert abs(eta - 1) < 1e-12 and np.allclose(vec, np.eye(2) / np.sqrt(2))


def test_batches_vs_oracle():
    from scipy.stats import unitary_group
    B = 200
    U = unitary_group.rvs(4, size=4 * B, random_state=5)
    U1, U2, U1p, U2p = U[:B], U[B:2 * B], U[2 * B:3 * B], U[3 * B:]
    rng = np.random.default_rng(2)
    O2 = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
    O4 = rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16))
    oc, re_, le, mo = NT.OverlapCalculator(), NT.RightEnvironment(), NT.LeftEnvironment(), NT.ManifoldOverlap()
    e2 = oc.mqbt2_exp_val(U1, U2, O2)                       # shared operator
    assert np.abs(e2 - [BW.exp_val_2(a, b, O2) for a, b in zip(U1, U2)]).max() < 1e-12
    e4 = oc._expval(U1[:40], U2[:40], O4, 4)
    assert np.abs(e4 - [BW.exp_val_4(a, b, O4) for a, b in zip(U1[:40], U2[:40])]).max() < 1e-11
    # generic (non state-like) environments: eigenpair with the largest REAL part, like eta[np.argmax(eta)]
    for env, fn in ((re_, BW.right_env_matrix), (le, BW.left_env_matrix)):
        mats = env.exact_environment_circuit(U1, U2, U1p, U2p)
        ref = np.stack([fn(a, b, c, d) for a, b, c, d in zip(U1, U2, U1p, U2p)])
        assert np.abs(mats - ref).max() < 1e-13
        lib_eta, lib_vec, st = env._env(U1, U2, U1p, U2p, False)[1:4]
        good = st == 0
        assert good.mean() > 0.9
        for k in np.flatnonzero(good):
            w = np.linalg.eigvals(ref[k])
            order = np.argsort(-w.real)
            if w.real[order[0]] - w.real[order[1]] < 1e-3:
                continue                                     # near-tie in the real part: skip
            eta_o, vec_o = BW.dominant(ref[k])
            assert abs(lib_eta[k] - eta_o) < 1e-9
            assert abs(abs(np.vdot(vec_o.reshape(-1), lib_vec[k].reshape(-1))) - 1) < 1e-8
    Mr = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
    Ml = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
    W = unitary_group.rvs(16, random_state=9)
    ov = mo.circuit(U1[:60], U2[:60], U1p[:60], U2p[:60], Mr, Ml, W)
    assert np.abs(ov - [BW.manifold_overlap(a, b, c, d, Mr, Ml, W) for a, b, c, d in zip(U1, U2, U1p, U2p)][:60]).max() < 1e-12
    # Represent.exact_env of a state with itself: both environments have eta = 1
    Mr_, Ml_ = NT.Represent().exact_env(U1[:10], U2[:10], dag(U1[:10]), dag(U2[:10]))
    assert Mr_.shape == (10, 2, 2) and Ml_.shape == (10, 2, 2)


# ---- driver classes (callers of the contractions): new_tdvp/testTDVPStripped.py:236-330 ----------------
X0 = np.array([[0, 1], [1, 0]], dtype=complex)
Z0 = np.diag([1.0, -1.0]).astype(complex)
I2 = np.eye(2, dtype=complex)


def test_represent_known_answer():
    """testTDVPStripped.py:242-251: U1 = X x X, U1' = U1^+, U2 = U2' = 1: the hand-rolled gradient descent on
    ||eta M - RE(M)|| reaches the exact environment."""
    RE1 = NT.Represent()
    U1 = np.kron(X0, X0).reshape(2, 2, 2, 2)
    U1_ = U1.reshape(4, 4).conj().T.reshape(2, 2, 2, 2)
    U2 = np.kron(I2, I2).reshape(2, 2, 2, 2)
    U2_ = np.kron(I2, I2).reshape(2, 2, 2, 2)
    M_exact = RE1.exact_env(U1, U2, U1_, U2_)
    res = RE1.optimize_by_hand([U1, U2, U1_, U2_])
    assert np.allclose(M_exact, RE1.M(res.x[1:]))
    # the Nelder-Mead variant finds a fixed point of the same map: eta M == RE(M)
    res2 = RE1.optimize(U1, U2, U1_, U2_)
    assert res2.fun < 1e-6
    M = RE1.M(res2.x[1:])
    assert np.allclose(res2.x[0] * M, NT.RightEnvironment().circuit(U1, U2, U1_, U2_, M), atol=1e-6)


def test_represent_random_unitaries_fixed_point():
    rng = np.random.default_rng(5)
    from scipy.stats import unitary_group
    U1 = unitary_group.rvs(4, random_state=rng)
    U2 = unitary_group.rvs(4, random_state=rng)
    R = NT.Represent()
    res = R.optimize(U1, U2, U1.conj().T, U2.conj().T)
    Mr, Ml = R.exact_env(U1, U2, U1.conj().T, U2.conj().T)
    M = R.M(res.x[1:])
    assert res.fun < 1e-5
    # same ray as the exact environment (the variational M has unit Frobenius norm by construction at a = pi/4 only)
    ov = abs(np.vdot(Mr, M)) / (np.linalg.norm(Mr) * np.linalg.norm(M))
    assert ov > 1 - 1e-6


def test_optimize_reaches_ground_state_of_IZ():
    """testTDVPStripped.py:311-329: <1 x Z> can be driven to -1 from random parameters."""
    O = np.kron(I2, Z0).reshape(2, 2, 2, 2)
    rng = np.random.default_rng(11)
    for _ in range(2):
        OP = NT.Optimize()
        res = OP.optimize(O, initial_params=rng.random(22))
        assert np.allclose(res.fun, -1, atol=1e-4)
        U1, U2 = OP.paramU(res.x)
        assert np.allclose(OP.mcost_function(res.x), res.fun, atol=1e-10)
        assert len(OP.energy_opt) > 10 and OP.energy_opt[-1] <= OP.energy_opt[0] + 1e-12
    # population evaluation == one-by-one evaluation
    P = rng.random((9, 22))
    OP.O = O
    assert np.allclose(OP.batch_cost_function(P), [OP.cost_function(p) for p in P], atol=1e-12)


def test_evolve_identity_and_short_step():
    """W = 1 at the state's own parameters: both environments are 1/sqrt2 (unit 2-norm eigenvectors, as
    `exact_environment` returns them in the reference too), so the overlap is 1/2 * <psi|psi> and the cost -1/4 - the
    reference's normalisation (with M = 1 the same circuit gives 1, testTDVPStripped.py:180-191).  A short-time
    propagator is followed by the projection step (:331-375 without the plots)."""
    from scipy.linalg import expm
    rng = np.random.default_rng(21)
    p = rng.random(22)
    EV = NT.Evolve()
    U1, U2 = EV.paramU(p)
    EV.W, EV.U1, EV.U2 = np.eye(16), U1, U2
    Mr, Ml = EV.RE.exact_env(U1, U2, U1.conj().T, U2.conj().T)
    assert np.allclose(np.abs(Mr), np.eye(2) / np.sqrt(2), atol=1e-9)
    assert abs(EV.exact_cost_function(p) + 0.25) < 1e-9
    assert abs(EV.mcost_function(p) + 0.25) < 1e-9
    assert abs(NT.ManifoldOverlap().circuit(U1, U2, U1.conj().T, U2.conj().T, np.eye(2), np.eye(2), np.eye(16)) - 1) < 1e-9
    assert EV.exact_cost_function(rng.random(22)) > -0.25 + 1e-3
    H = NT.tensor([X0, X0, X0, X0])
    W = expm(-1j * H * 1e-3)
    res = EV.time_evolve(2, W, init_params=p)
    assert len(res) == 2 and all(-0.25 - 1e-6 < r.fun < -0.2499 for r in res)
    OPT = NT.Optimizer()
    assert isinstance(OPT.evolve, NT.Evolve) and isinstance(OPT.represent, NT.Represent) and isinstance(OPT.optimize, NT.Optimize)


def test_environment_eigenpair_at_special_unitaries_found_by_the_stress():
    """Round 5 (profiles/experiments/r05/stress_brickwall.py): with U1' = Z x X and U2' = X x 1 the environment matrix has a DOUBLE eigenvalue 0 whose
    eigenvectors are columns of exp(M) itself.  The solve used to accept the largest column of the power as soon as it was an eigenvector of
    ANYTHING - the kernel vector, residual 0 at round 0 - and returned eta = 0 where the reference's rule eta[np.argmax(eta)] names an
    eigenvalue with a positive real part (4 of 9 360 separated cases of the stress).  Now the power must be rank one - or, for a DEGENERATE
    leading eigenvalue (the double 0 itself leads in a third of these seeds: any of its eigenvectors is an answer), the eigenvalue must be the
    one with the largest real part as measured by the growth rate of the power."""
    from scipy.stats import unitary_group
    Z, X, I = np.diag([1.0, -1.0]), np.array([[0, 1.0], [1.0, 0]]), np.eye(2)
    n = 12
    U1 = np.stack([unitary_group.rvs(4, random_state=100 + s) for s in range(n)])
    U2 = np.stack([unitary_group.rvs(4, random_state=200 + s) for s in range(n)])
    U1p = np.stack([np.kron(Z, X).astype(complex)] * n)
    U2p = np.stack([np.kron(X, I).astype(complex)] * n)
    seen = set()
    for env, fn in ((NT.RightEnvironment(), BW.right_env_matrix), (NT.LeftEnvironment(), BW.left_env_matrix)):
        mats, eta, vec, st = env._env(U1, U2, U1p, U2p, True)[:4]
        assert np.all(st == 0)
        for k in range(n):
            M = fn(U1[k], U2[k], U1p[k], U2p[k])
            assert np.abs(mats[k] - M).max() < 1e-13
            w = np.linalg.eigvals(M)
            w = w[np.argsort(-w.real)]
            assert int((np.abs(w) < 1e-12).sum()) == 2                                   # the double zero
            lead_is_zero = abs(w[0]) < 1e-12
            seen.add(lead_is_zero)
            assert abs(eta[k] - w[0]) < 1e-9, (k, eta[k], w)
            x = vec[k].reshape(-1)
            assert np.abs(M @ x - eta[k] * x).max() < 1e-9 and abs(np.linalg.norm(x) - 1.0) < 1e-12
    assert seen == {True, False}                                                         # both situations occur among the seeds
