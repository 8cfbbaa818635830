import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine, _lib as L
rng = np.random.default_rng(6)
grid = np.array([0, np.pi / 4, np.pi / 2, np.pi, -np.pi / 2])
base = np.array(list(itertools.product(grid, repeat=6)))[::13]
prm = np.concatenate([base, base + 1e-9 * rng.standard_normal(base.shape), base + 1e-4 * rng.standard_normal(base.shape), 2 * rng.standard_normal((2000, 6))])
h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
B = len(prm)
eng = EnergyEngine(8, B); eng.set_hamiltonian(h)
eng.set_ansatz_params(L.ANSATZ_SHALLOW_CNOT, prm)
for solver in ('direct', 'plain'):
    eng.launch(B, max_iter=20000, solver=solver); E, it, st = eng.results(B); r = eng.environments(B); A = eng.tensors(B)
    ok = st == 0
    print(solver, 'B', B, 'status', np.bincount(st, minlength=3), 'iters>1', int((it > 1).sum()), 'max it', it.max(), 'finite', np.isfinite(E[ok]).all())
    Tr = np.einsum('bsij,bjk,bslk->bil', A, r, A.conj()); Tr /= np.trace(Tr, axis1=1, axis2=2)[:, None, None]
    print('  fixed point residual (ok items)', np.abs(Tr - r)[ok].max())
    T = np.einsum('bsij,bskl->bikjl', A, A.conj()).reshape(B, 64, 64)
    w = np.sort(np.abs(np.linalg.eigvals(T)), axis=1)[:, ::-1]; gap = w[:, 0] - w[:, 1]
    print('  not converged despite gap > 1e-3:', int(((st == 1) & (gap > 1e-3)).sum()))
    worst = 0
    for b in np.flatnonzero(ok & (gap > 1e-6))[::9]:
        worst = max(worst, max(abs(E[b, t] - O.energy_closed_form(A[b], h[t])) * min(1.0, gap[b] / 1e-6) for t in range(2)))
    print('  worst scaled |dE| vs dense eig', worst)
    if solver == 'direct': E0, st0 = E, st
print('direct vs plain: status differs', int((st0 != st).sum()), 'max |dE| both ok', np.abs(E0 - E)[(st0 == 0) & (st == 0)].max())
