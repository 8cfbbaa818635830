"""Round 5: randomised stress of the brick-wall kernels (qmps_bw_*; new_tdvp/ClassicalTDVPStripped.py:239-275, 316-431, 464-555) against the numpy oracle:
Haar-random two-qubit unitaries AND special ones (1, SWAP, CNOT, CZ, H x H, X x X, products and tiny perturbations of them).  Expectation values and
environment matrices are closed contractions (1e-12); the environment EIGENPAIR follows the reference's rule eta[np.argmax(eta)] (largest real part):
status 0 must carry that eigenvalue when it is separated in real part, status != 0 only at ties / defective matrices.
Usage: python profiles/experiments/r05/stress_brickwall.py [n_batches] [seed]"""
import sys, json, time
import numpy as np
from scipy.stats import unitary_group
sys.path.insert(0, '.')
from oracle import brickwall_oracle as BW
from qmps_amd import new_tdvp as NT

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
I, X, Z = np.eye(2), np.array([[0, 1.0], [1.0, 0]]), np.diag([1.0, -1.0])
Hd = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
SW = np.eye(4)[[0, 2, 1, 3]]
CN = np.eye(4)[[0, 1, 3, 2]]
SPECIAL = [np.eye(4), SW, CN, np.diag([1, 1, 1, -1.0]), np.kron(Hd, Hd), np.kron(X, X), np.kron(X, I), np.kron(Hd, I), CN @ np.kron(Hd, I), SW @ CN, np.kron(Z, X)]


def draw(n):
    out = []
    for _ in range(n):
        r = rng.random()
        if r < 0.5:
            U = unitary_group.rvs(4, random_state=int(rng.integers(1 << 31)))
        else:
            U = SPECIAL[rng.integers(len(SPECIAL))].astype(complex)
            if rng.random() < 0.5:
                U = U @ SPECIAL[rng.integers(len(SPECIAL))]
            if r > 0.85:
                G = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
                from scipy.linalg import expm
                U = U @ expm(1j * 10.0 ** rng.uniform(-9, -2) * (G + G.conj().T))
        out.append(U)
    return np.stack(out)


oc, re_, le = NT.OverlapCalculator(), NT.RightEnvironment(), NT.LeftEnvironment()
tot = {'status_nonzero_separated': 0, 'status_nonzero_degenerate_leading': 0, 'items': 0, 'max_d_expval2': 0.0, 'max_d_envmat': 0.0, 'status0': 0, 'status_nonzero': 0, 'separated': 0, 'max_d_eta_status0': 0.0}
bad, t0 = [], time.time()
for batch in range(n_batches):
    B = 256
    U1, U2, U1p, U2p = draw(B), draw(B), draw(B), draw(B)
    if rng.random() < 0.5:
        U1p, U2p = np.conj(np.swapaxes(U1, -1, -2)), np.conj(np.swapaxes(U2, -1, -2))       # a state with itself
    O2 = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
    e2 = oc.mqbt2_exp_val(U1, U2, O2)
    d2 = np.abs(e2 - [BW.exp_val_2(a, b, O2) for a, b in zip(U1, U2)]).max()
    tot['max_d_expval2'] = max(tot['max_d_expval2'], float(d2))
    for side, (env, fn) in enumerate(((re_, BW.right_env_matrix), (le, BW.left_env_matrix))):
        mats, eta, vec, st = env._env(U1, U2, U1p, U2p, True)[:4]
        ref = np.stack([fn(a, b, c, d) for a, b, c, d in zip(U1, U2, U1p, U2p)])
        tot['max_d_envmat'] = max(tot['max_d_envmat'], float(np.abs(mats - ref).max()))
        for k in range(B):
            w, v = np.linalg.eig(ref[k])
            order = np.argsort(-w.real)
            gap = w.real[order[0]] - w.real[order[1]]
            tot['items'] += 1
            tot['status0'] += int(st[k] == 0)
            tot['status_nonzero'] += int(st[k] != 0)
            if gap < 1e-6:
                if st[k] != 0 and abs(w[order[0]] - w[order[1]]) < 1e-9:
                    tot['status_nonzero_degenerate_leading'] += 1       # one eigenvalue, several eigenvectors: any of them would do
                # (a DEFECTIVE cluster - a triple zero - carries eps^(1/3) ~ 1e-8 .. 1e-7 of noise in numpy's own eigenvalues and in the kernel's: 1e-6 here)
                elif st[k] == 0 and abs(w[order[0]] - w[order[1]]) < 1e-9 and abs(eta[k] - w[order[0]]) > 1e-6:
                    bad.append({'batch': batch, 'side': side, 'k': k, 'what': 'degenerate leading eigenvalue, status 0, another eigenvalue returned', 'eta': [eta[k].real, eta[k].imag], 'w': [[x_.real, x_.imag] for x_ in w[order]]})
                continue
            tot['separated'] += 1
            if st[k] == 0:
                d = abs(eta[k] - w[order[0]])
                tot['max_d_eta_status0'] = max(tot['max_d_eta_status0'], float(d))
                x = vec[k].reshape(-1)
                res = float(np.abs(ref[k] @ x - eta[k] * x).max())
                if not d < 1e-8 or not res < 1e-8:
                    bad.append({'batch': batch, 'side': side, 'k': k, 'what': 'status 0, wrong eigenpair', 'd_eta': float(d), 'residual': res, 'gap': float(gap), 'w': [[x_.real, x_.imag] for x_ in w[order]]})
            else:
                tot['status_nonzero_separated'] += 1
            if st[k] != 0 and gap > 1e-3:
                cond = float(np.linalg.cond(v))
                bad.append({'batch': batch, 'side': side, 'k': k, 'what': f'status {int(st[k])} although the real parts are separated', 'gap': float(gap), 'cond_eigvecs': cond, 'w': [[x_.real, x_.imag] for x_ in w[order]]})
print(json.dumps({'batches': n_batches, 'seed': seed, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:8]}))
