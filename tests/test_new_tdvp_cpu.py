"""CPU: the host-only parts of the new_tdvp driver mirror (gate matrices, parameter maps, descent loop, finite
brick-wall states).  The contractions themselves need the GPU (tests/test_brickwall_gpu.py)."""
import numpy as np

from qmps_amd import new_tdvp as NT


def test_circuit_solver_gates():
    cs = NT.CircuitSolver()
    for t in (0.0, 0.3, 1.7):
        X, Z = cs.X(t), cs.Z(t)
        assert np.allclose(X @ X.conj().T, np.eye(2)) and np.allclose(Z @ Z.conj().T, np.eye(2))
        assert np.allclose(cs.D(t).sum(0), 1) and np.allclose(np.trace(cs.D2(t)), 1)
        assert np.isclose(np.linalg.norm(cs.D3(t)), 1) and np.isclose(np.linalg.norm(cs.D1(t)), 1)
    assert np.allclose(cs.X(1.0), -1j * np.array([[0, 1], [1, 0]]))            # X(theta) = exp(-i pi theta X / 2)
    M = cs.M([np.pi / 4, 0.2, 0.4, -0.3, 0.9, 0.1])
    assert np.isclose(np.linalg.norm(M), 1)                                      # unitary . diag(cos, sin) . unitary
    assert np.allclose(cs.M([np.pi / 4, 0, 0, 0, 0, 0]), np.eye(2) / np.sqrt(2))


def test_parameter_maps():
    rng = np.random.default_rng(0)
    cs = NT.CircuitSolver()
    U1, U2 = cs.paramU(rng.random(22))
    assert np.allclose(U1 @ U1.conj().T, np.eye(4)) and np.allclose(U2 @ U2.conj().T, np.eye(4))
    B1, B2 = cs.batch_paramU(rng.random((3, 22)))
    assert B1.shape == B2.shape == (3, 4, 4)
    lam = NT.OO_lambdas()
    assert lam.shape == (7, 4, 4) and all(np.allclose(g, g.conj().T) and abs(np.trace(g)) < 1e-15 for g in lam)
    assert all(np.abs(g[:, 0]).sum() > 0 for g in lam)
    # the seven generators reach every normalised first column: tangent space at the identity has real rank 7
    T = np.stack([np.concatenate([(-1j * g)[:, 0].real, (-1j * g)[:, 0].imag]) for g in lam])
    assert np.linalg.matrix_rank(T) == 7
    assert np.allclose(NT.OO_unitary(np.zeros(7)), np.eye(4))


def test_gradient_descent_and_states():
    res = NT.gradient_descent(lambda x: float(((x - 1) ** 2).sum()), lambda x: 2 * (x - 1), np.zeros(3))
    assert res.fun < 1e-6 and np.allclose(res.x, 1, atol=1e-2) and res.message.lower().startswith('answer')
    res = NT.gradient_descent(lambda x: 1.0 + float((x ** 2).sum()), lambda x: 2 * x, np.ones(2))
    assert res.message == 'CF stopped changing' and abs(res.fun - 1) < 1e-6
    p = np.random.default_rng(3).random(22)
    for l in (2, 3):
        psi = NT.state_from_params(p, l)
        assert psi.shape == (4 ** l,) and np.isclose(np.vdot(psi, psi).real, 1)
    hist = NT.optimize_2layer_bwmps(np.kron(np.diag([1.0, -1.0]), np.diag([1.0, -1.0])), initial_params=p, maxiter=40)
    assert len(hist) > 5 and hist[-1] <= hist[0] + 1e-12
