import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine, _lib as L
rng = np.random.default_rng(8)
grid = np.array([0, np.pi / 4, np.pi / 2, np.pi, -np.pi / 2])
base = np.array(list(itertools.product(grid, repeat=8)))[::997]
for D, P in ((16, 8), (2, 2)):
    b0 = base[:, :P]
    prm = np.concatenate([b0, b0 + 1e-4 * rng.standard_normal(b0.shape), 2 * rng.standard_normal((600, P))])
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    B = len(prm)
    eng = EnergyEngine(D, B); eng.set_hamiltonian(h)
    eng.set_ansatz_params(L.ANSATZ_SHALLOW_CNOT, prm)
    eng.launch(B, max_iter=20000); E, it, st = eng.results(B); r = eng.environments(B); A = eng.tensors(B)
    ok = st == 0
    print('D', D, 'B', B, 'status', np.bincount(st, minlength=3), 'max it', it.max(), 'finite', np.isfinite(E[ok]).all())
    Tr = np.einsum('bsij,bjk,bslk->bil', A, r, A.conj()); Tr /= np.trace(Tr, axis1=1, axis2=2)[:, None, None]
    print('  fixed point residual (ok items)', np.abs(Tr - r)[ok].max())
    worst, n, nc = 0, 0, 0
    for b in np.flatnonzero(ok)[::5]:
        T = np.einsum('sij,skl->ikjl', A[b], A[b].conj()).reshape(D * D, D * D)
        w = np.sort(np.abs(np.linalg.eigvals(T)))[::-1]; gap = w[0] - w[1]
        if gap > 1e-4:
            worst = max(worst, max(abs(E[b, t] - O.energy_closed_form(A[b], h[t])) for t in range(2))); n += 1
    for b in np.flatnonzero(st == 1)[:200]:
        T = np.einsum('sij,skl->ikjl', A[b], A[b].conj()).reshape(D * D, D * D)
        w = np.sort(np.abs(np.linalg.eigvals(T)))[::-1]
        if w[0] - w[1] > 1e-2: nc += 1
    print('  checked', n, 'worst |dE| vs dense eig', worst, '; not converged despite gap > 1e-2:', nc)
    eng.close()
