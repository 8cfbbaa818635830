"""Differential stress of the D = 4 direct kernel (incl. its matrix-core fall-back) on structured ansatz angles: grids of
multiples of pi/4 (degenerate / product-state transfer spectra), tiny perturbations of them, and random angles."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import qmps_oracle as O, c_oracle
from qmps_amd import EnergyEngine, _lib as L
rng = np.random.default_rng(5)
grid = np.array([0, np.pi / 4, np.pi / 2, np.pi, -np.pi / 2, 3 * np.pi / 4])
base = np.array(list(itertools.product(grid, repeat=4)))                       # 1296 special points
prm = np.concatenate([base, base + 1e-9 * rng.standard_normal(base.shape), base + 1e-5 * rng.standard_normal(base.shape),
                      base + 1e-2 * rng.standard_normal(base.shape), rng.standard_normal((20000, 4)) * 2])
h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
B = len(prm)
eng = EnergyEngine(4, B); eng.set_hamiltonian(h)
eng.set_ansatz_params(L.ANSATZ_SHALLOW_CNOT, prm)
eng.launch(B, max_iter=100000); E, it, st = eng.results(B); r = eng.environments(B)
A = eng.tensors(B)
print('B', B, 'status counts', np.bincount(st, minlength=3), 'fallback', int((it > 1).sum()), 'max iters', it.max())
assert np.isfinite(E[st == 0]).all()
# ground truth: dense eig of the transfer matrix where the dominant eigenvalue is well separated
bad = 0; checked = 0; worst = 0.0
idx = np.concatenate([np.arange(0, 4 * 1296), 4 * 1296 + np.arange(0, 20000, 10)])
for b in idx:
    T = np.einsum('sij,skl->ikjl', A[b], A[b].conj()).reshape(16, 16)
    w = np.linalg.eigvals(T); w = w[np.argsort(-abs(w))]
    gap = abs(w[0]) - abs(w[1])
    if st[b] == 0 and gap > 1e-6:
        e = [O.energy_closed_form(A[b], h[t]) for t in range(2)]
        err = max(abs(E[b, t] - e[t]) for t in range(2)); worst = max(worst, err); checked += 1
        if err > 1e-9 * max(1.0, 1e-6 / gap): bad += 1; print('MISMATCH', b, prm[b], err, gap, it[b])
    if st[b] == 1 and gap > 1e-3: bad += 1; print('NOT CONVERGED despite gap', b, prm[b], gap, it[b])
    # status 0 means the returned environment IS a fixed point to the criterion
    if st[b] == 0:
        Tr = np.einsum('sij,jk,slk->il', A[b], r[b], A[b].conj())
        if not np.abs(Tr / np.trace(Tr) - r[b]).max() < 1e-11: bad += 1; print('NOT A FIXED POINT', b, prm[b], np.abs(Tr - r[b]).max())
print('checked', checked, 'worst |dE|', worst, 'bad', bad)
# the same batch through the iterative solvers agrees wherever both converge
eng.set_tensors(A); eng.launch(B, max_iter=100000, solver='squaring'); E2, it2, st2 = eng.results(B)
both = (st == 0) & (st2 == 0)
print('squaring: status counts', np.bincount(st2, minlength=3), 'max |dE| where both ok', np.abs(E - E2)[both].max(), 'status differs', int((st != st2).sum()))
d = np.abs(E - E2).max(1); d[~both] = 0
for b in np.argsort(-d)[:8]:
    T = np.einsum('sij,skl->ikjl', A[b], A[b].conj()).reshape(16, 16)
    w = np.linalg.eigvals(T); w = w[np.argsort(-abs(w))]
    print('item', b, 'params/pi', np.round(prm[b] / np.pi, 4), 'dE', d[b], '|w|', np.round(abs(w[:4]), 8), 'iters', it[b], it2[b])
diff = np.flatnonzero(st != st2)
gaps = []
for b in diff[:400]:
    T = np.einsum('sij,skl->ikjl', A[b], A[b].conj()).reshape(16, 16)
    w = np.sort(abs(np.linalg.eigvals(T)))[::-1]; gaps.append(w[0] - w[1])
gaps = np.array(gaps); print('status differs:', len(diff), 'of which gap < 1e-6:', int((gaps < 1e-6).sum()), 'largest gap among them', gaps.max())
import collections
print(collections.Counter(zip(st[diff].tolist(), st2[diff].tolist())))
big = diff[np.argsort(-gaps)[:5]]
for b in big:
    print('  ', b, np.round(prm[b] / np.pi, 5), 'st', st[b], st2[b], 'it', it[b], it2[b])
