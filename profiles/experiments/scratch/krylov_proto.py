"""Prototype (numpy) of the Krylov fall-back of the D = 8 / 16 overlap fixed-point solves: thick-restart Arnoldi in its
Davidson form (basis V, images W = T V, projected G = V^H T V; restart = any orthonormal k-dim subspace, no Krylov relation
to preserve), the small problem solved by SQUARING G (the D = 4 kernel's machinery on one 16 x 16 tile).
Counts map applications on Haar-far candidates.  usage: krylov_proto.py D n [m] [k] [variant]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import expm, schur
from oracle import qmps_oracle as O


def dense_map(A, B, WW):
    C = np.tensordot(WW, O.merge(A, A), [1, 0])
    return O.transfer_matrix(C, O.merge(B, B))


def squaring_pair(G, rounds=48, tol=1e-14, snap=4):
    """dominant right eigenvector of G by squaring; also a snapshot of an early power for the restart subspace"""
    n2 = np.linalg.norm(G)
    if n2 == 0:
        return None, None, 0
    M = G / n2
    S = None
    for r in range(rounds):
        Q = M @ M
        t = np.trace(M)
        q2 = np.linalg.norm(Q)
        if r == snap:
            S = M.copy()
        if q2 == 0:
            break
        if np.linalg.norm(Q - t * M) < tol * q2:
            break
        M = Q / q2
    if S is None:
        S = M
    c = int(np.argmax(np.linalg.norm(M, axis=0)))
    y = M[:, c] / np.linalg.norm(M[:, c])
    return y, S, r


def pivoted_basis(y, S, k):
    """orthonormal [y, pivoted Gram-Schmidt of the columns of S] (k columns)"""
    m = len(y)
    Q = np.zeros((m, k), dtype=complex)
    Q[:, 0] = y
    R = S - np.outer(y, y.conj() @ S)
    for j in range(1, k):
        nrm = np.linalg.norm(R, axis=0)
        c = int(np.argmax(nrm))
        if nrm[c] < 1e-14:
            # numerically rank-deficient: fill with anything orthogonal
            v = np.random.default_rng(j).standard_normal(m) + 0j
            v -= Q[:, :j] @ (Q[:, :j].conj().T @ v)
            q = v / np.linalg.norm(v)
        else:
            q = R[:, c] / nrm[c]
            q -= Q[:, :j] @ (Q[:, :j].conj().T @ q)      # (re-orthogonalise)
            q /= np.linalg.norm(q)
        Q[:, j] = q
        R = R - np.outer(q, q.conj() @ R)
    return Q


def schur_basis(G, k):
    T, Z = schur(G, output='complex')
    # reorder: largest modulus first (selection by repeated sort through scipy's sort callable is awkward: use eig + QR instead)
    w, v = np.linalg.eig(G)
    idx = np.argsort(-np.abs(w))[:k]
    Q, _ = np.linalg.qr(v[:, idx])
    return Q[:, 0] * 1.0, Q


def second_pair(G, y, V, W):
    """second Ritz value of G (deflated squaring), its Schur vector and the residual vector of that"""
    m = len(y)
    P = np.eye(m) - np.outer(y, y.conj())
    G2 = P @ G @ P
    y2, _, r2 = squaring_pair(G2)
    if y2 is None:
        return 0.0, None, None, 0
    y2 = P @ y2
    y2 /= np.linalg.norm(y2)
    th2 = y2.conj() @ G @ y2
    g = y.conj() @ G @ y2
    res2 = W @ y2 - (V @ y) * g - th2 * (V @ y2)
    return th2, y2, res2, r2


def krylov_solve(E, x0, tol=1e-12, m=16, k=5, max_mv=3000, variant='snap', snap=4, certify=True, margin=100.0, stats=None, rounds_cap=48):
    """thick-restart Arnoldi, Davidson form: any orthonormal k-dim restart subspace, expansion by the residual of the targeted
    Ritz pair, then the Arnoldi chain T v_j within the cycle"""
    N = len(x0)
    V = np.zeros((N, m), dtype=complex)
    W = np.zeros((N, m), dtype=complex)
    G = np.zeros((m, m), dtype=complex)
    V[:, 0] = x0 / np.linalg.norm(x0)
    j0 = 0
    mv = 0
    eta = 0
    cycles = 0
    refreshed = False
    u = V[:, 0]
    while mv + (m - j0) <= max_mv:
        for j in range(j0, m):
            w = E @ V[:, j]
            mv += 1
            W[:, j] = w
            G[:j + 1, j] = V[:, :j + 1].conj().T @ w
            if j > 0:
                G[j, :j] = V[:, j].conj() @ W[:, :j]
            if j + 1 < m:
                h = V[:, :j + 1].conj().T @ w
                w = w - V[:, :j + 1] @ h
                h = V[:, :j + 1].conj().T @ w
                w = w - V[:, :j + 1] @ h
                nw = np.linalg.norm(w)
                V[:, j + 1] = w / nw if nw > 1e-300 else 0
        cycles += 1
        if variant == 'schur':
            w_, v_ = np.linalg.eig(G)
            idx = np.argsort(-np.abs(w_))
            y = v_[:, idx[0]]
            y = y / np.linalg.norm(y)
            Q, _ = np.linalg.qr(np.column_stack([y, v_[:, idx[1:k]]]))
        elif variant in ('defl', 'defl2', 'defl3'):
            Q = np.zeros((m, k), dtype=complex)
            Gd = G
            for i in range(k):
                yi, _, rr = squaring_pair(Gd, rounds=rounds_cap)
                if stats is not None:
                    stats.setdefault('rounds', []).append(rr)
                for _ in range(2):
                    yi = yi - Q[:, :i] @ (Q[:, :i].conj().T @ yi)
                yi /= np.linalg.norm(yi)
                Q[:, i] = yi
                P = np.eye(m) - np.outer(yi, yi.conj())
                Gd = P @ Gd @ P
        else:
            y, S, rr = squaring_pair(G, snap=snap)
            Q = pivoted_basis(y, S, k)
        y = Q[:, 0]
        u = V @ y
        Tu = W @ y
        eta = y.conj() @ G @ y
        res = Tu - eta * u
        rn = np.linalg.norm(res)
        t = res
        if variant in ('defl2', 'defl3'):
            # strong certificate: EVERY kept Schur pair must lie below the first, residual-aware
            Vn_ = V @ Q
            Wn_ = W @ Q
            R = Q.conj().T @ G @ Q
            resid = []
            for i_ in range(k):
                ri_ = Wn_[:, i_] - Vn_[:, :i_ + 1] @ R[:i_ + 1, i_]
                resid.append(ri_)
            rn = np.linalg.norm(resid[0])
            eta = R[0, 0]
            t = resid[0]
            ok = False
            if rn < tol:
                bad = [i_ for i_ in range(1, k) if not (abs(R[i_, i_]) + margin * (np.linalg.norm(resid[i_]) + rn) < abs(eta))]
                if not bad:
                    ok = True
                else:
                    t = resid[bad[0]]
                    if stats is not None:
                        stats['tie_cycles'] = stats.get('tie_cycles', 0) + 1
            if ok and variant == 'defl3' and not refreshed:
                # one more cycle whose new direction is the ORIGINAL start vector (it holds every dominant component): a dominant
                # eigenvalue that was discarded at a restart before it was resolved shows up again and fails the certificate
                refreshed = True
                ok = False
                t = x0 / np.linalg.norm(x0)
                if stats is not None:
                    stats['refresh'] = stats.get('refresh', 0) + 1
            if ok:
                Tu2 = E @ u
                mv += 1
                nu = np.linalg.norm(u)
                u, Tu2 = u / nu, Tu2 / nu
                eta = u.conj() @ Tu2
                if np.linalg.norm(Tu2 - eta * u) < tol * 1.5:
                    return eta, u, mv, cycles, 0
        elif rn < tol:
            ok = True
            if certify:
                if variant == 'defl':
                    y2 = Q[:, 1]
                    th2 = y2.conj() @ G @ y2
                    res2 = W @ y2 - u * (y.conj() @ G @ y2) - th2 * (V @ y2)
                else:
                    th2, y2, res2, r2 = second_pair(G, y, V, W)
                if y2 is not None:
                    ok = abs(th2) + margin * (np.linalg.norm(res2) + rn) < abs(eta)
                    if not ok:
                        if stats is not None:
                            stats['tie_cycles'] = stats.get('tie_cycles', 0) + 1
                        # keep [y, y2, ...] and aim the expansion at the second pair
                        Q2 = np.column_stack([y, y2, Q[:, 1:k - 1]])
                        Q, _ = np.linalg.qr(Q2)
                        t = res2
            if ok:
                Tu2 = E @ u
                mv += 1
                nu = np.linalg.norm(u)
                u, Tu2 = u / nu, Tu2 / nu
                eta = u.conj() @ Tu2
                if np.linalg.norm(Tu2 - eta * u) < tol * 1.5:
                    return eta, u, mv, cycles, 0
        Vn = V @ Q
        Wn = W @ Q
        Gn = Q.conj().T @ G @ Q
        for _ in range(2):
            t = t - Vn @ (Vn.conj().T @ t)
        nt = np.linalg.norm(t)
        V[:, :k] = Vn
        W[:, :k] = Wn
        G[:] = 0
        G[:k, :k] = Gn
        V[:, k] = t / nt
        j0 = k
    return eta, u, mv, cycles, 1


def power(E, x0, tol, cap):
    x = x0 / np.linalg.norm(x0)
    for kk in range(1, cap + 1):
        xn = E @ x
        eta = x.conj() @ xn
        if np.linalg.norm(xn - eta * x) < tol:
            return eta, x, kk, 0
        x = xn / np.linalg.norm(xn)
    return eta, x, cap, 1


if __name__ == '__main__':
    D = int(sys.argv[1]); n = int(sys.argv[2])
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    variant = sys.argv[5] if len(sys.argv) > 5 else 'snap'
    snap = int(sys.argv[6]) if len(sys.argv) > 6 else 4
    pcap = int(sys.argv[7]) if len(sys.argv) > 7 else 64
    rng = np.random.default_rng(int(os.environ.get('SEED', '0')))
    WW = expm(-1j * 0.2 * O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}))
    tot = []
    bad = 0
    fail = 0
    for b in range(n):
        A = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, 1)[0])
        B = O.unitary_to_tensor(O.haar_unitaries(rng, 2 * D, 1)[0])
        E = dense_map(A, B, WW)
        w = np.linalg.eigvals(E)
        ws = np.sort(np.abs(w))[::-1]
        ref = w[np.argmax(np.abs(w))]
        x0 = np.eye(D, dtype=complex).reshape(-1)
        eta, x, steps, st = power(E, x0, 1e-12, pcap)
        mv = steps
        cyc = 0
        if st:
            eta, x, mv2, cyc, st = krylov_solve(E, x, m=m, k=k, max_mv=3000 - steps, variant=variant, snap=snap)
            mv += mv2
        tot.append(mv)
        if st:
            fail += 1
            print('FAIL', b, 'ratio', ws[1] / ws[0], ws[2] / ws[0])
        elif abs(eta - ref) > 1e-10:
            bad += 1
            print('WRONG', b, abs(eta - ref), 'ratio', ws[1] / ws[0])
    tot = np.array(tot)
    print(f'D={D} n={n} m={m} k={k} {variant} snap={snap}: fail {fail} wrong {bad}; map applications mean {tot.mean():.0f} median {np.median(tot):.0f} p99 {np.percentile(tot, 99):.0f} max {tot.max()}')


def near_tie_pair(rng, D, ratio, WW, eps1=0.4):
    """reference / candidate tensors whose mixed transfer map has |eta_2 / eta_1| = ratio: block-diagonal sectors of bond
    dimension D/2 (the map decomposes into the four sector pairs), the candidate of sector 2 moved along a path until its
    dominant eigenvalue has the wanted modulus, then hidden behind random gauges"""
    d = D // 2
    def tens(U):
        return O.unitary_to_tensor(U)
    def kick(U, e, K):
        return expm(e * K / np.linalg.norm(K)) @ U
    def antiherm(n):
        K = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return K - K.conj().T
    def top(Aa, Bb):
        w = np.linalg.eigvals(dense_map(Aa, Bb, WW))
        return np.sort(np.abs(w))[::-1]
    while True:
        U1, U2 = O.haar_unitaries(rng, 2 * d, 2)
        A1, A2 = tens(U1), tens(U2)
        B1 = tens(kick(U1, eps1, antiherm(2 * d)))
        K2 = antiherm(2 * d)
        e11 = top(A1, B1)[0]
        lo, hi = 0.0, 0.6          # |eta22| decreases from ~1 as the candidate moves away
        f = lambda t: top(A2, tens(kick(U2, t, K2)))[0] - ratio * e11
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
