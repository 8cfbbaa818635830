"""Hard inputs for the D = 8 / 16 fixed-point solves (test infrastructure): candidates unrelated to the reference state and
pairs with a PRESCRIBED ratio |eta_2 / eta_1| of the two dominant eigenvalues of the mixed transfer map."""
import numpy as np
from scipy.linalg import expm

from oracle import qmps_oracle as O


def dense_map(A, B, WW):
    """the D^2 x D^2 matrix of x -> sum_s (WW . merge(A, A))_s x merge(B, B)_s^+ (row-major vec)"""
    C = np.tensordot(WW, O.merge(A, A), [1, 0])
    return O.transfer_matrix(C, O.merge(B, B))


def near_tie_pair(rng, D, ratio, WW, eps1=0.4):
    """Reference / candidate tensors (2, D, D) whose map has |eta_2 / eta_1| = ratio.  Both are direct sums of two sectors of
    bond dimension D/2, so the map decomposes into the four sector pairs; the candidate of sector 2 is moved along a path
    exp(t K) U_2 until the dominant eigenvalue of its sector has the wanted modulus (bisection with dense eigen-solves), then both
    tensors are hidden behind random unitary gauges (the spectrum does not change, the block structure is gone)."""
    d = D // 2

    def tens(U):
        return O.unitary_to_tensor(U)

    def kick(U, e, K):
        return expm(e * K / np.linalg.norm(K)) @ U

    def antiherm(n):
        K = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return K - K.conj().T

    def top(Aa, Bb):
        return np.abs(np.linalg.eigvals(dense_map(Aa, Bb, WW))).max()

    while True:
        U1, U2 = O.haar_unitaries(rng, 2 * d, 2)
        A1, A2 = tens(U1), tens(U2)
        B1 = tens(kick(U1, eps1, antiherm(2 * d)))
        K2 = antiherm(2 * d)
        e11 = top(A1, B1)
        f = lambda t: top(A2, tens(kick(U2, t, K2))) - ratio * e11
        lo, hi = 0.0, 0.6          # the sector's |eta| falls from ~1 as its candidate moves away
        if f(lo) > 0 > f(hi):
            break
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if f(mid) > 0:
            lo = mid
        else:
            hi = mid
        if hi - lo < 1e-16:
            break
    B2 = tens(kick(U2, 0.5 * (lo + hi), K2))

    def dsum(X, Y):
        Z = np.zeros((2, D, D), dtype=complex)
        Z[:, :d, :d] = X
        Z[:, d:, d:] = Y
        return Z
    Gg, Hg = O.haar_unitaries(rng, D, 2)
    A = np.einsum('ij,sjk,lk->sil', Gg, dsum(A1, A2), Gg.conj())
    B = np.einsum('ij,sjk,lk->sil', Hg, dsum(B1, B2), Hg.conj())
    return A, B


def dominant(A, B, WW):
    """(eta_1, |eta_2 / eta_1|) by the dense eigen-solve"""
    w = np.linalg.eigvals(dense_map(A, B, WW))
    w = w[np.argsort(-np.abs(w))]
    return w[0], abs(w[1]) / abs(w[0])


def slow_environment_tensor(rng, D, t):
    """A left-isometric state tensor (2, D, D) whose transfer map has |lambda_2| = 1 - O(t^2): the direct sum of two sectors of
    bond dimension D/2 (two fixed points) behind a random gauge, its unitary then kicked by exp(t K) - the sectors couple weakly.
    t = 0.3 / 0.1 / 0.03 / 0.01 give |lambda_2| ~ 0.997 / 0.9997 / 0.99996 / 0.999997."""
    d = D // 2
    U1, U2 = O.haar_unitaries(rng, 2 * d, 2)
    A = np.zeros((2, D, D), dtype=complex)
    A[:, :d, :d] = O.unitary_to_tensor(U1)
    A[:, d:, d:] = O.unitary_to_tensor(U2)
    G = O.haar_unitaries(rng, D, 1)[0]
    A = np.einsum('ij,sjk,lk->sil', G, A, G.conj())
    K = rng.standard_normal((2 * D, 2 * D)) + 1j * rng.standard_normal((2 * D, 2 * D))
    K = K - K.conj().T
    return O.unitary_to_tensor((expm(t * K / np.linalg.norm(K)) @ O.tensor_to_unitary(A))[None])[0]
