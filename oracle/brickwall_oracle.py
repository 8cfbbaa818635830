"""CPU oracle for the brick-wall ("new_tdvp") classical contractions - TEST INFRASTRUCTURE ONLY.

Restates, with explicit state vectors and operator chains (no einsum index strings), what
new_tdvp/ClassicalTDVPStripped.py computes by tensor-network einsums:

  bwMPS.state                      :179-191   psi = (1 x U1^(l-1) x 1)(U2^l)|0..0>  on 2 l qubits
  OverlapCalculator.qbt2_exp_val   :511-544   <psi_2| 1 x O x 1 |psi_2>            (mqbt2: :546-555)
  OverlapCalculator.qbt4_exp_val   :464-496   <psi_3| 1 x O x 1 |psi_3>            (mqbt4: :498-507)
  RightEnvironment.circuit         :355-379   <j,0,0| U2'_bc U1'_ab M_c U1_ab U2_bc |i,0,0>
  RightEnvironment.exact_environment_circuit  :399-422  the 4 x 4 matrix of that map, exact_environment :424-431
  LeftEnvironment.exact_environment_circuit   :316-338, exact_environment :340-347
  ManifoldOverlap.circuit / mcircuit          :239-285

U1, U2 (and the primed U1', U2') are 4 x 4 matrices acting on two neighbouring qubits, big-endian;
`U.reshape(2,2,2,2)` in the reference is [out0, out1, in0, in1].  PINNED against outputs of the reference's own
code run in the build container (tests/golden/brickwall_golden.npz, generator tests/golden/make_golden.py) and
the reference's exact known answers (new_tdvp/testTDVPStripped.py:71-170).
"""
import numpy as np

from .qmps_oracle import _on


def bw_state(U1, U2, l):
    n = 2 * l
    psi = np.zeros(2 ** n, dtype=complex)
    psi[0] = 1
    for k in range(l):
        psi = _on(n, U2, [2 * k, 2 * k + 1]) @ psi
    for k in range(l - 1):
        psi = _on(n, U1, [2 * k + 1, 2 * k + 2]) @ psi
    return psi


def exp_val_2(U1, U2, O):
    """<O> on the two middle qubits of the 4-qubit brick-wall state (complex; the reference's qbt2 takes .real)."""
    psi = bw_state(U1, U2, 2)
    return psi.conj() @ (_on(4, O, [1, 2]) @ psi)


def exp_val_4(U1, U2, O):
    psi = bw_state(U1, U2, 3)
    return psi.conj() @ (_on(6, O, [1, 2, 3, 4]) @ psi)


def _chain3(U1, U2, U1p, U2p, mid):
    """8 x 8 operator U2'_bc U1'_ab mid U1_ab U2_bc on wires (a, b, c)."""
    return _on(3, U2p, [1, 2]) @ _on(3, U1p, [0, 1]) @ mid @ _on(3, U1, [0, 1]) @ _on(3, U2, [1, 2])


def right_env_circuit(U1, U2, U1p, U2p, M):
    op = _chain3(U1, U2, U1p, U2p, _on(3, M, [2]))
    # result[j][i] = <j,0,0| op |i,0,0>
    return np.array([[op[4 * j, 4 * i] for i in range(2)] for j in range(2)])


def right_env_matrix(U1, U2, U1p, U2p):
    """Mmat[(i,i'),(j,j')] = <i',0,0| U2' U1' (|j'><j|)_c U1 U2 |i,0,0>."""
    out = np.zeros((4, 4), dtype=complex)
    for j in range(2):
        for jp in range(2):
            E = np.zeros((2, 2), dtype=complex)
            E[jp, j] = 1
            op = _chain3(U1, U2, U1p, U2p, _on(3, E, [2]))
            for i in range(2):
                for ip in range(2):
                    out[2 * i + ip, 2 * j + jp] = op[4 * ip, 4 * i]
    return out


def left_env_matrix(U1, U2, U1p, U2p):
    """Mirror image on wires (a, b, c): U2 on (a,b) from |00>, U1 on (b,c) with the c input open;
    Mmat[(c_in,c_out),(a,a')] = <0,0,c_out| U2'_ab U1'_bc (|a'><a|)_a U1_bc U2_ab |0,0,c_in>."""
    out = np.zeros((4, 4), dtype=complex)
    for a in range(2):
        for ap in range(2):
            E = np.zeros((2, 2), dtype=complex)
            E[ap, a] = 1
            op = _on(3, U2p, [0, 1]) @ _on(3, U1p, [1, 2]) @ _on(3, E, [0]) @ _on(3, U1, [1, 2]) @ _on(3, U2, [0, 1])
            for ci in range(2):
                for co in range(2):
                    out[2 * ci + co, 2 * a + ap] = op[co, ci]
    return out


def dominant(Mmat):
    """eig + argmax exactly as the reference (:427-431): numpy's argmax on the complex eigenvalues
    (lexicographic: real part first).  Eigenvector normalised to unit 2-norm with the phase that makes its
    largest-magnitude entry real positive (the reference returns LAPACK's arbitrary phase)."""
    w, v = np.linalg.eig(Mmat)
    k = int(np.argmax(w))
    vec = v[:, k]
    vec = vec / np.linalg.norm(vec)
    big = vec[np.argmax(np.abs(vec))]
    return w[k], (vec * np.conj(big) / abs(big)).reshape(2, 2)


def manifold_overlap(U1, U2, U1p, U2p, Mr, Ml, W):
    n = 6
    psi = np.zeros(2 ** n, dtype=complex)
    psi[0] = 1
    for k in range(3):
        psi = _on(n, U2, [2 * k, 2 * k + 1]) @ psi
    for k in range(2):
        psi = _on(n, U1, [2 * k + 1, 2 * k + 2]) @ psi
    psi = _on(n, Ml, [0]) @ psi
    psi = _on(n, W, [1, 2, 3, 4]) @ psi
    psi = _on(n, Mr, [5]) @ psi
    for k in range(2):
        psi = _on(n, U1p, [2 * k + 1, 2 * k + 2]) @ psi
    for k in range(3):
        psi = _on(n, U2p, [2 * k, 2 * k + 1]) @ psi
    return psi[0]
