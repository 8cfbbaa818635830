"""Replay one case of stress_overlap.py and look at one candidate: the dense spectrum, and what the device returns with and without the Krylov
fall-back, from tensors and from parameters.  Usage: python profiles/experiments/r05/stress_overlap_repro.py <seed> <case> <b>"""
import os, sys, json
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
import evolve_replay as ER

seed, target, bsel = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
Hm = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
GRID = np.array([0.0, np.pi / 4, -np.pi / 4, np.pi / 2, -np.pi / 2, np.pi])
for case in range(target + 1):
    D = int(rng.choice([2, 4, 8, 16]))
    kind = int(rng.choice([0, 3] if D > 2 else [0, 1, 2, 3]))
    depth = int(rng.integers(1, 5))
    P = {0: 2 * depth, 1: 2 * depth, 2: 15, 3: 3 * depth}[kind]
    B = int(rng.integers(1, 12)) if D == 16 else int(rng.integers(1, 60))
    ref = rng.standard_normal((B, P))
    mode = rng.choice(['near', 'far', 'grid'])
    if mode == 'near':
        cand = ref + 10.0 ** rng.uniform(-6, -1) * rng.standard_normal((B, P))
    elif mode == 'far':
        cand = rng.standard_normal((B, P))
    else:
        cand = GRID[rng.integers(0, len(GRID), size=(B, P))] + (1e-9 * rng.standard_normal((B, P)) if rng.random() < 0.3 else 0.0)
        if rng.random() < 0.5:
            ref = GRID[rng.integers(0, len(GRID), size=(B, P))].astype(float)
    dt = 0.0
    if rng.random() < 0.3:
        WW = np.eye(4, dtype=complex)
    else:
        dt = float(rng.choice([0.02, 0.05, 0.1, 0.3]))
        WW = expm(-1j * dt * Hm)
print(json.dumps({'D': D, 'kind': kind, 'P': P, 'B': B, 'mode': str(mode), 'ref': ref[bsel].tolist(), 'cand': cand[bsel].tolist(), 'dt': dt}))
A = ER.tensor(kind, D, ref[bsel])
Bt = ER.tensor(kind, D, cand[bsel])
C = np.tensordot(WW, O.merge(A, A), [1, 0])
Bm = O.merge(Bt, Bt)
w, v = np.linalg.eig(O.transfer_matrix(C, Bm))
order = np.argsort(-np.abs(w))
print('top eigenvalues', [(complex(np.round(x, 6)), float(np.round(abs(x), 6))) for x in w[order][:8]])
x0 = np.eye(D).reshape(-1) / np.sqrt(D)
vinv = np.linalg.pinv(v)
print('|components of the identity along the top eigenvectors|', np.abs(vinv @ x0)[order][:8])
for label, env in (('default', {}), ('QMPS_NO_KRYLOV', {'QMPS_NO_KRYLOV': '1'}), ('QMPS_NO_DEFLATION', {'QMPS_NO_DEFLATION': '1'}), ('no Krylov, no deflation', {'QMPS_NO_KRYLOV': '1', 'QMPS_NO_DEFLATION': '1'})):
    for k in ('QMPS_NO_KRYLOV', 'QMPS_NO_DEFLATION'):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = EnergyEngine(D, 1024)
    for how in ('params', 'tensor'):
        c_in = cand[bsel][None] if how == 'params' else Bt[None]
        eta, rounds, st = eng.overlaps(A[None], c_in, WW, kind=how, ansatz=kind if how == 'params' else None, tol=1e-12, max_rounds=200000)
        print(f'{label:26s} {how:7s}: eta {complex(eta[0]):.9f} |eta| {abs(eta[0]):.9f} rounds {int(rounds[0])} status {int(st[0])}')
