#!/usr/bin/env python3
"""Generate the committed golden fixtures (tests/golden/*.npz).

Run ONLY in the build container, where the reference checkout is mounted read-only at
/root/reference:   python tests/golden/make_golden.py

Two kinds of vectors are written:

 (1) ``ref_*`` arrays - outputs of the REFERENCE'S OWN pure-numpy functions
     (qmps/tools.py unitary_to_tensor / tensor_to_unitary / environment_to_unitary,
     qmps/time_evolve_tools.py merge, qmps/ground_state.py Hamiltonian.to_matrix), imported
     from /root/reference with throw-away stub modules standing in for the un-installable
     third-party imports (cirq, xmps, tqdm-less, matplotlib-less).  The stubs provide NO
     arithmetic except ``xmps.spin.paulis`` (the Pauli matrices - pinned by the reference's own
     4x4 TFIM known answer, tests/test_ground_state.py:26-38).
 (2) ``oracle_*`` arrays - outputs of this repo's CPU oracle (oracle/qmps_oracle.py) on the same
     seeded inputs; they freeze the double restatement (state-vector vs closed-form) so that
     the GPU box - which has no /root/reference - can check the HIP path against data.

No reference source text is stored: the fixtures are inputs and numeric outputs only.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def _stub_modules():
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Any()

        def __getattr__(self, name):
            return _Any()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__getattr__ = lambda n: _Any()  # type: ignore
        sys.modules[name] = m
        return m

    cirq = mod('cirq', Gate=object)
    sx = np.array([[0, 1], [1, 0]], dtype=complex)
    sy = np.array([[0, -1j], [1j, 0]], dtype=complex)
    sz = np.array([[1, 0], [0, -1]], dtype=complex)
    mod('xmps')
    mod('xmps.spin', paulis=lambda s: (sx, sy, sz))
    mod('xmps.iMPS')
    mod('xmps.tensor')
    mod('xmps.iOptimize')
    mod('tqdm', tqdm=lambda x, *a, **k: x, tqdm_notebook=lambda x, *a, **k: x)
    mod('jax', device_put=lambda x: x, jit=lambda f, *a, **k: f)
    mod('jax.numpy')
    for name in ('matplotlib', 'matplotlib.pyplot'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod(name)
    return cirq


def main():
    assert os.path.isdir(REF), 'reference checkout not mounted - fixtures can only be regenerated in the build container'
    _stub_modules()
    sys.path.insert(0, REF)
    from qmps import tools as rtools                      # noqa: E402
    from qmps import time_evolve_tools as rtet            # noqa: E402
    from qmps import ground_state as rgs                  # noqa: E402
    from oracle import qmps_oracle as O                   # noqa: E402

    out = {}
    # ---- (1a) Hamiltonian.to_matrix  (ground_state.py:73-88)
    out['ref_h_tfim'] = rgs.Hamiltonian({'ZZ': -1, 'X': 1}).to_matrix()
    out['ref_h_tfim_split'] = rgs.Hamiltonian({'ZZ': -1, 'IX': 0.5, 'XI': 0.5}).to_matrix()
    out['ref_h_xxz'] = rgs.Hamiltonian({'XX': 1, 'YY': 1, 'ZZ': 0.5}).to_matrix()
    out['ref_h_xy'] = rgs.Hamiltonian({'XX': 1, 'YY': 1}).to_matrix()
    out['ref_h_tfim_g07'] = rgs.Hamiltonian({'ZZ': -1, 'X': 0.7}).to_matrix()

    rng = np.random.default_rng(20241022)
    for D in (2, 4, 8, 16):
        n = 4 if D <= 8 else 2
        U = O.haar_unitaries(rng, 2 * D, n)
        out[f'U_D{D}'] = U
        # ---- (1b) unitary_to_tensor (tools.py:151-154)
        A = np.stack([rtools.unitary_to_tensor(u) for u in U])
        out[f'ref_A_D{D}'] = A
        # ---- (2) oracle energies on these inputs, both restatements
        h = out['ref_h_tfim']
        r = np.stack([O.env_dense_eig(a)[1] for a in A])
        out[f'oracle_r_D{D}'] = r
        out[f'oracle_E_closed_D{D}'] = np.array([O.energy_closed_form(a, h, rr) for a, rr in zip(A, r)])
        if D <= 8:
            # ---- (1c) environment_to_unitary on the Cholesky factor (tools.py:97-108, 181-182)
            L = np.stack([O.env_cholesky(rr) for rr in r])
            V = np.stack([rtools.environment_to_unitary(l) for l in L])
            out[f'oracle_L_D{D}'] = L
            out[f'ref_V_D{D}'] = V
            # state-vector energy using the REFERENCE's V completion
            out[f'oracle_E_statevec_D{D}'] = np.array(
                [O.energy_statevector(u, h, v) for u, v in zip(U, V)])
            out[f'oracle_psi_D{D}'] = np.stack([O.state_vector(u, v, 2) for u, v in zip(U, V)])
        pw = [O.energy_power(a, h) for a in A]
        out[f'oracle_E_power_D{D}'] = np.array([p[0] for p in pw])
        out[f'oracle_iters_D{D}'] = np.array([p[1] for p in pw], dtype=np.int32)

    # ---- (1d) tensor_to_unitary round trip with its 5 internal checks (tools.py:123-148; D=2 only)
    A2 = out['ref_A_D2']
    t2u = [rtools.tensor_to_unitary(a, testing=True) for a in A2]
    out['ref_t2u_passed'] = np.array([bool(p) for _, p in t2u])
    out['ref_t2u_U'] = np.stack([u for u, _ in t2u])
    out['ref_t2u_roundtrip'] = np.stack([rtools.unitary_to_tensor(u) for u, _ in t2u])
    # ---- (1e) merge (time_evolve_tools.py:20-23; hard-codes D=2)
    out['ref_merge_D2'] = np.stack([rtet.merge(A2[i], A2[(i + 1) % len(A2)]) for i in range(len(A2))])
    # ---- (1f) to/from_real_vector, environment_from_unitary, direct_sum helpers
    v = rng.standard_normal(8)
    out['realvec_in'] = v
    out['ref_from_real_vector'] = rtools.from_real_vector(v)
    out['ref_to_real_vector'] = rtools.to_real_vector(out['ref_V_D2'][0])
    out['ref_env_from_unitary'] = rtools.environment_from_unitary(out['ref_V_D2'][0])

    # ---- (3) the reference's golden INPUT fixtures/A.npy (xmps flat format: [d, D, n_sites] + data)
    flat = np.load(os.path.join(REF, 'fixtures', 'A.npy'))
    out['ref_fixture_A_flat'] = flat

    # ---- (2b) ansatz family fixtures (represent.py:288-310, 393-401) through the oracle
    p_cnot = rng.standard_normal((6, 4))
    out['cnot_params_D2'] = p_cnot[:, :2]
    out['cnot_params_D4'] = p_cnot
    for D, P in ((2, p_cnot[:, :2]), (4, p_cnot)):
        Us = np.stack([O.shallow_cnot_unitary(D, p) for p in P])
        out[f'oracle_cnot_U_D{D}'] = Us
        res = []
        for u in Us:
            a = O.unitary_to_tensor(u)
            _, rr = O.env_dense_eig(a)
            res.append(O.energy_closed_form(a, out['ref_h_tfim'], rr))
        out[f'oracle_cnot_E_D{D}'] = np.array(res)
    p_full = rng.standard_normal((4, 15))
    out['full_params'] = p_full
    out['oracle_full_U'] = np.stack([O.shallow_full_unitary(p) for p in p_full])

    # ---- (2c) two-site unit cell (ground_state.py:291-331), D=2
    U1, U2 = O.haar_unitaries(rng, 4, 3), O.haar_unitaries(rng, 4, 3)
    out['cell_U1'], out['cell_U2'] = U1, U2
    out['oracle_cell_E'] = np.array([O.two_site_cell_energy(a, b, out['ref_h_tfim']) for a, b in zip(U1, U2)])

    path = os.path.join(HERE, 'qmps_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')
    brickwall_golden()


def brickwall_golden():
    """(4) new_tdvp brick-wall contractions: the reference's OWN classical code
    (new_tdvp/ClassicalTDVPStripped.py), which is pure numpy/scipy once its unused jax/xmps/cirq imports are
    stubbed, run on seeded random unitaries.  These are reference OUTPUTS, the strongest parity anchor."""
    from scipy.stats import unitary_group
    sys.path.insert(0, os.path.join(REF, 'new_tdvp'))
    import ClassicalTDVPStripped as ref
    out = {}
    n = 6
    U = [unitary_group.rvs(4, random_state=100 + k) for k in range(4 * n)]
    U1, U2, U1p, U2p = (np.stack(U[k::4]) for k in range(4))
    out['U1'], out['U2'], out['U1p'], out['U2p'] = U1, U2, U1p, U2p
    rng = np.random.default_rng(7)

    def herm(m):
        x = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
        return x + x.conj().T

    O2 = np.stack([herm(4) for _ in range(n)])
    O4 = np.stack([herm(16) for _ in range(n)])
    M = np.stack([rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)) for _ in range(n)])
    Ml = np.stack([rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)) for _ in range(n)])
    W = np.stack([unitary_group.rvs(16, random_state=200 + k) for k in range(n)])
    out['O2'], out['O4'], out['M'], out['Ml'], out['W'] = O2, O4, M, Ml, W
    oc, re_, le, mo = ref.OverlapCalculator(), ref.RightEnvironment(), ref.LeftEnvironment(), ref.ManifoldOverlap()
    t4 = lambda u: u.reshape(2, 2, 2, 2)                                    # noqa: E731
    dag = lambda u: u.conj().T.reshape(2, 2, 2, 2)                          # noqa: E731
    out['ref_qbt2'] = np.array([oc.qbt2_exp_val(t4(a), t4(b), t4(o)) for a, b, o in zip(U1, U2, O2)])
    out['ref_mqbt2'] = np.array([oc.mqbt2_exp_val(a, b, o) for a, b, o in zip(U1, U2, O2)])
    out['ref_qbt4'] = np.array([oc.qbt4_exp_val(t4(a), t4(b), o.reshape((2,) * 8)) for a, b, o in zip(U1, U2, O4)])
    out['ref_mqbt4'] = np.array([oc.mqbt4_exp_val(a, b, o) for a, b, o in zip(U1, U2, O4)])
    out['ref_RE_circuit'] = np.stack([re_.circuit(t4(a), t4(b), t4(c), t4(d), m)
                                      for a, b, c, d, m in zip(U1, U2, U1p, U2p, M)])
    out['ref_RE_matrix'] = np.stack([re_.exact_environment_circuit(t4(a), t4(b), t4(c), t4(d))
                                     for a, b, c, d in zip(U1, U2, U1p, U2p)])
    out['ref_LE_matrix'] = np.stack([le.exact_environment_circuit(t4(a), t4(b), t4(c), t4(d))
                                     for a, b, c, d in zip(U1, U2, U1p, U2p)])
    # state-like case U' = U^dagger: dominant eigenvalue 1
    out['ref_RE_matrix_dag'] = np.stack([re_.exact_environment_circuit(t4(a), t4(b), dag(a), dag(b)) for a, b in zip(U1, U2)])
    out['ref_LE_matrix_dag'] = np.stack([le.exact_environment_circuit(t4(a), t4(b), dag(a), dag(b)) for a, b in zip(U1, U2)])
    ee = [re_.exact_environment(t4(a), t4(b), dag(a), dag(b)) for a, b in zip(U1, U2)]
    out['ref_RE_eta_dag'] = np.array([e for e, _ in ee])
    out['ref_RE_vec_dag'] = np.stack([v for _, v in ee])
    ee = [le.exact_environment(t4(a), t4(b), dag(a), dag(b)) for a, b in zip(U1, U2)]
    out['ref_LE_eta_dag'] = np.array([e for e, _ in ee])
    out['ref_LE_vec_dag'] = np.stack([v for _, v in ee])
    out['ref_manifold'] = np.array([mo.circuit(t4(a), t4(b), t4(c), t4(d), mr, ml, w.reshape((2,) * 8))
                                    for a, b, c, d, mr, ml, w in zip(U1, U2, U1p, U2p, M, Ml, W)])
    out['ref_mmanifold'] = np.array([mo.mcircuit(a, b, c, d, mr, ml, w)
                                     for a, b, c, d, mr, ml, w in zip(U1, U2, U1p, U2p, M, Ml, W)])
    out['ref_bwmps_state_l2'] = np.stack([ref.bwMPS([b, a], 2).state() for a, b in zip(U1, U2)])
    path = os.path.join(HERE, 'brickwall_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
