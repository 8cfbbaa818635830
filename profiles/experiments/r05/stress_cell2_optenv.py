"""Round 5: randomised stress of the two-site-unit-cell energy (a-9, qmps_cell2_energy_batch) and of the variational-environment objective (a-12,
qmps_opt_env_objective) against the oracle: Haar-random and special two-qubit unitaries (1, SWAP, CNOT, CZ, H x H, X x X, products, tiny perturbations)
for the cell; random and grid angles for the 30-parameter objective.  Silent errors = status 0 with an energy off the oracle's where the oracle's
environments are unique and satisfy their fixed-point equations.
Usage: python profiles/experiments/r05/stress_cell2_optenv.py [n_batches] [seed]"""
import sys, json, time
import numpy as np
from scipy.stats import unitary_group
from scipy.linalg import expm
sys.path.insert(0, '.')
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
I, X, Z = np.eye(2), np.array([[0, 1.0], [1.0, 0]]), np.diag([1.0, -1.0])
Hd = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
SW, CN = np.eye(4)[[0, 2, 1, 3]], np.eye(4)[[0, 1, 3, 2]]
SPECIAL = [np.eye(4), SW, CN, np.diag([1, 1, 1, -1.0]), np.kron(Hd, Hd), np.kron(X, X), np.kron(X, I), np.kron(Hd, I), CN @ np.kron(Hd, I), SW @ CN, np.kron(Z, X)]
GRID = np.array([0.0, np.pi / 4, -np.pi / 4, np.pi / 2, -np.pi / 2, np.pi])
H = {'tfim': O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0}), 'xxz': O.hamiltonian_matrix({'XX': 1.0, 'YY': 1.0, 'ZZ': 0.5})}


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
            if r > 0.8:
                G = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
                U = U @ expm(1j * 10.0 ** rng.uniform(-9, -1) * (G + G.conj().T))
        out.append(U)
    return np.stack(out)


def polished_env(A):
    """(r, unique, ok): dominant fixed point of the d = 4 transfer map of the merged tensor by dense eig + power polish (see env_dense_eig)."""
    w = np.linalg.eigvals(O.transfer_matrix(A))
    w = np.sort(np.abs(w))[::-1]
    if w[1] > (1 - 1e-6) * w[0]:
        return None, False, False
    r = np.eye(A.shape[1], dtype=complex) / A.shape[1]
    res = 1.0
    for _ in range(4000):
        rn = O.apply_transfer(A, r)
        rn = rn / np.trace(rn)
        res = np.abs(rn - r).max()
        r = rn
        if res < 1e-15:
            break
    return (r + r.conj().T) / 2, True, res < 1e-12


eng = EnergyEngine(2, 4096)
tot = {'cell_items': 0, 'cell_status0': 0, 'cell_status_nonzero': 0, 'cell_checked': 0, 'cell_max_dE': 0.0, 'optenv_items': 0, 'optenv_max_d': 0.0}
bad, t0 = [], time.time()
for batch in range(n_batches):
    B = 200
    U1, U2 = draw(B), draw(B)
    hname = str(rng.choice(['tfim', 'xxz']))
    E, it, st = eng.cell2_energies(U1, U2, H[hname])
    E = E[:, 0]
    for k in range(B):
        tot['cell_items'] += 1
        tot['cell_status0'] += int(st[k] == 0)
        tot['cell_status_nonzero'] += int(st[k] != 0)
        A1, A2 = O.unitary_to_tensor(U1[k]), O.unitary_to_tensor(U2[k])
        r12, u12, ok12 = polished_env(O.merge(A1, A2))
        r21, u21, ok21 = polished_env(O.merge(A2, A1))
        if not (u12 and u21 and ok12 and ok21):
            continue
        lam = min(np.linalg.eigvalsh(r12).min(), np.linalg.eigvalsh(r21).min())
        e_or = O.two_site_cell_energy_closed(A1, A2, H[hname], r12, r21)
        if st[k] == 0:
            tot['cell_checked'] += 1
            d = abs(E[k] - e_or)
            tot['cell_max_dE'] = max(tot['cell_max_dE'], float(d))
            if not d < 1e-8:
                bad.append({'what': 'cell: status 0, wrong energy', 'batch': batch, 'k': k, 'E': float(E[k]), 'oracle': float(e_or), 'lam_min': float(lam)})
        elif st[k] == 2 and lam > 1e-9:
            bad.append({'what': 'cell: status 2 although both environments are positive definite', 'batch': batch, 'k': k, 'lam_min': float(lam)})
        elif st[k] == 1:
            bad.append({'what': 'cell: status 1 although both environments are unique', 'batch': batch, 'k': k, 'iters': int(it[k])})
    # variational-environment objective: 30 angles, half of them on the grid
    P = rng.standard_normal((B, 30))
    P = np.where(rng.random((B, 30)) < 0.5, GRID[rng.integers(0, len(GRID), size=(B, 30))], P)
    kpen = float(rng.choice([0.0, 1.0, 3.0]))
    f = eng.opt_env_objective(P, H[hname], k=kpen)
    f_or = np.array([O.opt_environment_objective(p, H[hname], k=kpen)[0] for p in P])
    d = np.abs(f - f_or)
    tot['optenv_items'] += B
    tot['optenv_max_d'] = max(tot['optenv_max_d'], float(d.max()))
    for k in np.flatnonzero(~(d < 1e-10)):
        bad.append({'what': 'opt_env objective differs', 'batch': batch, 'k': int(k), 'f': float(f[k]), 'oracle': float(f_or[k])})
print(json.dumps({'batches': n_batches, 'seed': seed, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:10]}))
