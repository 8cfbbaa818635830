"""CPU: the brick-wall oracle against OUTPUTS OF THE REFERENCE'S OWN CODE (new_tdvp/ClassicalTDVPStripped.py run in
the build container, tests/golden/brickwall_golden.npz) and its exact known answers (testTDVPStripped.py:71-170)."""
import os

import numpy as np
import pytest

from oracle import brickwall_oracle as BW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bw():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'brickwall_golden.npz'))


def dag(u):
    return u.conj().T


def test_against_reference_outputs(bw):
    for k in range(len(bw['U1'])):
        U1, U2, U1p, U2p = bw['U1'][k], bw['U2'][k], bw['U1p'][k], bw['U2p'][k]
        assert np.allclose(BW.bw_state(U1, U2, 2), bw['ref_bwmps_state_l2'][k], atol=1e-14)
        e2 = BW.exp_val_2(U1, U2, bw['O2'][k])
        assert abs(e2.real - bw['ref_qbt2'][k]) < 1e-12 and abs(e2 - bw['ref_mqbt2'][k]) < 1e-12
        e4 = BW.exp_val_4(U1, U2, bw['O4'][k])
        assert abs(e4.real - bw['ref_qbt4'][k]) < 1e-11 and abs(e4.real - bw['ref_mqbt4'][k]) < 1e-11
        assert np.allclose(BW.right_env_circuit(U1, U2, U1p, U2p, bw['M'][k]), bw['ref_RE_circuit'][k], atol=1e-13)
        assert np.allclose(BW.right_env_matrix(U1, U2, U1p, U2p), bw['ref_RE_matrix'][k], atol=1e-13)
        assert np.allclose(BW.left_env_matrix(U1, U2, U1p, U2p), bw['ref_LE_matrix'][k], atol=1e-13)
        assert np.allclose(BW.right_env_matrix(U1, U2, dag(U1), dag(U2)), bw['ref_RE_matrix_dag'][k], atol=1e-13)
        assert np.allclose(BW.left_env_matrix(U1, U2, dag(U1), dag(U2)), bw['ref_LE_matrix_dag'][k], atol=1e-13)
        for mat, eta_ref, vec_ref in ((bw['ref_RE_matrix_dag'][k], bw['ref_RE_eta_dag'][k], bw['ref_RE_vec_dag'][k]),
                                      (bw['ref_LE_matrix_dag'][k], bw['ref_LE_eta_dag'][k], bw['ref_LE_vec_dag'][k])):
            eta, vec = BW.dominant(mat)
            assert abs(eta - eta_ref) < 1e-12 and abs(eta - 1) < 1e-12      # a state's transfer matrix: eta = 1
            assert abs(abs(np.vdot(vec_ref.reshape(-1), vec.reshape(-1))) - 1) < 1e-10   # same ray
        ov = BW.manifold_overlap(U1, U2, U1p, U2p, bw['M'][k], bw['Ml'][k], bw['W'][k])
        assert abs(ov - bw['ref_manifold'][k]) < 1e-12 and abs(ov - bw['ref_mmanifold'][k]) < 1e-12


def test_reference_known_answers():
    """new_tdvp/testTDVPStripped.py:71-170, exact +-1 cases."""
    I, Z, X = np.eye(2), np.diag([1.0, -1.0]), np.array([[0, 1.0], [1.0, 0]])
    Had = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    kr = np.kron
    assert np.isclose(BW.exp_val_2(kr(I, I), kr(I, I), kr(Z, Z)), 1)
    assert np.isclose(BW.exp_val_2(kr(X, X), kr(I, I), kr(Z, Z)), 1)
    assert np.isclose(BW.exp_val_2(kr(X, X), kr(I, I), kr(I, Z)), -1)
    assert np.isclose(BW.exp_val_2(kr(Had, Had), kr(I, I), kr(X, X)), 1)
    assert np.isclose(BW.exp_val_2(kr(Had, Had), kr(X, X), kr(X, I)), -1)
    Z4 = kr(kr(Z, Z), kr(Z, Z))
    assert np.isclose(BW.exp_val_4(kr(I, I), kr(I, I), Z4), 1)
    assert np.isclose(BW.exp_val_4(kr(X, X), kr(I, I), Z4), 1)
    assert np.isclose(BW.exp_val_4(kr(X, X), kr(I, I), kr(kr(I, Z), kr(Z, Z))), -1)
    assert np.isclose(BW.exp_val_4(kr(Had, Had), kr(X, X), kr(kr(X, I), kr(I, I))), -1)
    U1, U2 = kr(X, X), kr(I, I)
    assert np.allclose(BW.right_env_circuit(U1, U2, dag(U1), dag(U2), Z), I)
    Mx = BW.right_env_matrix(U1, U2, dag(U1), dag(U2))
    assert np.allclose(Mx, [[1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [1, 0, 0, 0]])
    eta, vec = BW.dominant(Mx)
    assert abs(eta - 1) < 1e-14 and np.allclose(vec, np.eye(2) / np.sqrt(2))
    # manifold overlap of a state with itself under W = 1 and the trivial environments of a product state
    assert np.isclose(BW.manifold_overlap(kr(I, I), kr(I, I), kr(I, I), kr(I, I), I, I, np.eye(16)), 1)
