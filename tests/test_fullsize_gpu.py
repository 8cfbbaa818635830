"""GPU, BASELINE.json's full size (D = 4, batch 65536; D = 2 batch 4096; D = 8 / 16 configs): the oracle
cannot cover 65536 items in seconds with numpy, so parity at full size goes through size-independent
properties of the path (plus the C oracle on a strided sample)."""
import numpy as np
import pytest

from oracle import qmps_oracle as O

pytestmark = pytest.mark.gpu
E0 = -4 / np.pi


def haar_tensors(seed, D, B):
    rng = np.random.default_rng(seed)
    out = np.empty((B, 2, D, D), dtype=np.complex128)
    for lo in range(0, B, 8192):
        n = min(8192, B - lo)
        out[lo:lo + n] = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, n))
    return out


@pytest.mark.parametrize('D,B,solver', [(4, 65536, 'direct'), (4, 65536, 'squaring'), (2, 4096, 'direct'), (2, 4096, 'squaring'), (8, 768, 'direct'),
                                        (8, 768, 'squaring'), (16, 768, 'squaring')])
def test_full_size_properties(D, B, solver, c_oracle, engine_factory):
    """solver = 'direct': the library default at D = 4 and 8 (exact fixed-point solve, accepted by one power step);
    'squaring': the iterative path of round 1 (and what D = 2, 16 run)."""
    A = haar_tensors(20241022, D, B)
    h1 = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
    h2 = O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})
    eng = engine_factory(D, 65536)
    eng.set_solver(solver, handoff=0)
    E, it, st = eng.energies(A, np.stack([h1, h2, h1 + 2 * h2, np.eye(4)]))
    ok = st == 0
    assert ok.mean() > 0.999
    if solver == 'direct':
        assert np.all(it == 1)                 # every Haar evaluation is accepted without a fall-back round
    # (1) linearity in the Hamiltonian and normalisation <1> = 1
    assert np.abs(E[:, 2] - E[:, 0] - 2 * E[:, 1])[ok].max() < 1e-12
    assert np.abs(E[:, 3] - 1)[ok].max() < 1e-12
    # (2) variational bound (tests/test_ground_state.py:218) and the two-site RDM: Hermitian, trace one, PSD
    assert E[ok, 0].min() >= E0
    rho = eng.rdm()
    assert np.abs(rho - rho.conj().transpose(0, 2, 1))[ok].max() < 1e-13
    assert np.abs(np.trace(rho, axis1=1, axis2=2) - 1)[ok].max() < 1e-12
    assert np.linalg.eigvalsh(rho[ok][::97]).min() > -1e-12
    # (3) fixed point: the returned environment is reproduced by one application of the transfer map
    r = eng.environments()
    sel = np.flatnonzero(ok)[::max(1, B // 512)]
    Tr = np.einsum('bsij,bjk,bslk->bil', A[sel], r[sel], A[sel].conj())
    assert np.abs(Tr - r[sel]).max() < 1e-11
    assert np.abs(np.trace(r, axis1=1, axis2=2) - 1)[ok].max() < 1e-13
    # (4) gauge invariance: A_s -> G A_s G^+ leaves the energy unchanged (r -> G r G^+)
    rng = np.random.default_rng(1)
    G = O.haar_unitaries(rng, D, 1)[0]
    Ag = np.einsum('ij,bsjk,lk->bsil', G, A, G.conj())
    Eg, _, stg = eng.energies(Ag, np.stack([h1, h2]))
    both = ok & (stg == 0)
    assert np.abs(Eg - E[:, :2])[both].max() < 1e-11
    # (5) checksum of checksums: device-side summed cost == host sum == sum over two independently run halves
    E1, _, _ = eng.energies(A, np.stack([h1, h2]))
    total = eng.summed_cost()
    assert np.allclose(total, E1.sum(0), rtol=0, atol=1e-8)
    Ea, _, _ = eng.energies(A[:B // 2], np.stack([h1, h2]))
    Eb, _, _ = eng.energies(A[B // 2:], np.stack([h1, h2]))
    if D == 16:     # half batches (<= 512) run two waves per evaluation: the partial maps are summed in another order
        assert np.abs(np.concatenate([Ea, Eb]) - E1).max() < 1e-12
    else:
        assert np.array_equal(np.concatenate([Ea, Eb]), E1)      # results do not depend on wave-mates
    # (6) the plain power iteration (the literal `krylov` algorithm) reaches the same energies
    eng.set_solver('plain')
    Ep, itp, stp = eng.energies(A, np.stack([h1, h2]))
    both = ok & (stp == 0)
    assert np.abs(Ep - E1)[both].max() < 1e-10
    eng.set_solver(solver, handoff=0)
    if solver == 'direct' and D == 4:
        # (6b) the exact in-kernel cost (fixed-point accumulation inside the fused kernel, no reduction kernel)
        eng.set_tensors(A)
        eng.set_hamiltonian(np.stack([h1, h2]))
        eng.launch(B, solver='direct', store_env=False, accumulate_cost=True)
        eng.cost_launch(B)
        acc = eng.get_cost()
        E2, _, _ = eng.results(B)
        assert np.array_equal(E2, E1) and np.allclose(acc, E1.sum(0), rtol=0, atol=1e-8)
    # (7) strided sample against the C oracle (plain algorithm)
    samp = np.arange(0, B, max(1, B // 1024))
    ref = c_oracle.energy_batch(A[samp], np.stack([h1, h2]), threads=8)
    g = (ref['status'] == 0) & ok[samp]
    assert np.abs(ref['E'] - E1[samp])[g].max() < 1e-10
