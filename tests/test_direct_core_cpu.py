"""The mathematics of the D = 4 direct-solve kernel (qmps_amd/csrc/qmps_direct_core.h - the very source the GPU
kernel is compiled from) run on the CPU in a four-lane lock-step emulation and compared with the oracle.
CPU only: the GPU parity tests proper are in test_direct_gpu.py."""
import numpy as np
import pytest

from oracle import qmps_oracle as O
from tests import direct_emu as EMU

H3 = lambda rng: np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}),
                           O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}),
                           rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))])


def test_golden_vectors_d4(golden):
    out = EMU.energies_d4(golden['ref_A_D4'], golden['ref_h_tfim'])
    assert np.all(out['status'] == 0) and np.all(out['iters'] == 1)
    assert np.abs(out['E'][:, 0] - golden['oracle_E_closed_D4']).max() < 1e-12
    assert np.abs(out['E'][:, 0] - golden['oracle_E_statevec_D4']).max() < 1e-12
    assert np.abs(out['r'] - golden['oracle_r_D4']).max() < 1e-12


def test_haar_batch_against_the_oracle():
    rng = np.random.default_rng(11)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 400))
    h = H3(rng)
    out = EMU.energies_d4(A, h)
    assert np.all(out['status'] == 0) and np.all(out['iters'] == 1) and out['resid'].max() < 1e-14
    assert np.abs(out['E_lean'] - out['E']).max() < 1e-13          # the density-matrix-free route of the energy-only kernel
    for b in range(0, 400, 7):
        r, it, st = O.env_direct(A[b])
        assert (it, st) == (1, 0)
        assert np.abs(out['r'][b] - r).max() < 1e-13
        assert np.abs(out['rho'][b] - O.two_site_rdm(A[b], r)).max() < 1e-13
        for t in range(3):
            assert abs(out['E'][b, t] - O.energy_closed_form(A[b], h[t], r)) < 1e-13
        _, rr = O.env_dense_eig(A[b])          # and the reference's own route: the dominant eigen-matrix
        assert np.abs(out['r'][b] - rr).max() < 1e-12


def test_ansatz_family_d4():
    """ShallowCNOT depth-2 tensors (the optimisers' default ansatz, represent.py:288-310): structured, far from Haar."""
    rng = np.random.default_rng(12)
    A = np.stack([O.unitary_to_tensor(O.shallow_cnot_unitary(4, rng.standard_normal(4))) for _ in range(300)])
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    out = EMU.energies_d4(A, h)
    ok = out['status'] == 0
    assert ok.mean() > 0.95
    for b in np.flatnonzero(ok)[::5]:
        assert abs(out['E'][b, 0] - O.energy_closed_form(A[b], h)) < 1e-11


def test_fallback_non_isometric_and_degenerate():
    rng = np.random.default_rng(13)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 60))
    A = A * rng.uniform(0.6, 1.5, size=(60, 1, 1, 1)) + 0.05 * (rng.standard_normal(A.shape) + 1j * rng.standard_normal(A.shape))
    h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    out = EMU.energies_d4(A, h, max_iter=100000)
    assert np.all(out['status'] == 0) and np.all(out['iters'] > 1)
    for b in range(60):
        r, it, st = O.env_direct(A[b], max_iter=100000)
        assert st == 0 and it == out['iters'][b]
        assert np.abs(out['r'][b] - r).max() < 1e-12
        assert np.abs(out['r'][b] - O.env_dense_eig(A[b])[1]).max() < 1e-10
    # the iteration cap is reported, not hidden: 1 verification step + 2^1 squared steps <= 3
    cut = EMU.energies_d4(A[:5], h, max_iter=3)
    assert np.all(cut['status'] == 1) and np.all(cut['iters'] == 3)
    # product state |00..0>: rank-one environment -> not positive definite (the reference's LinAlgError branch)
    U = np.eye(8, dtype=complex)[None]
    out = EMU.energies_d4(O.unitary_to_tensor(U), h)
    assert out['status'][0] == 2 and abs(out['E'][0, 0] + 1.0) < 1e-12
    E, it, st = O.energy_direct(O.unitary_to_tensor(U)[0], h)
    assert st == 2 and abs(E + 1.0) < 1e-12
