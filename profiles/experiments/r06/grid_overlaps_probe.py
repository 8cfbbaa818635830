"""Probe (GPU box): the overlap entry points (EnergyEngine.overlaps -> qmps_overlap_*) on PAIRS of the special grid (reference and candidate both at
multiples of pi / 4 or pi / 2), D = 2 .. 16, against Gelfand's formula.  A status that claims an answer (0, or 4 = tied) must carry |eta| = the
spectral radius; status 1 is counted (honest).  Prints per case: pairs, status counts, the worst deviation over answered pairs, and up to three
offenders."""
import os, sys, json
R = os.environ.get('GRAFT_REPO_ROOT', os.path.abspath(os.path.join(os.path.dirname(__file__), '../../..')))
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import numpy as np
from scipy.linalg import expm
import evolve_replay as ER
from oracle import qmps_oracle as O
from qmps_amd import _lib as L
from qmps_amd.engine import EnergyEngine

H = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
out = []
ONLY = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else None
for D, P in ((2, 2), (2, 8), (4, 4), (4, 8), (8, 6), (16, 8)):
    if ONLY and D not in ONLY:
        rng.random(1)
        continue
    kind = L.ANSATZ_SHALLOW_CNOT
    n = N if D <= 8 else N // 2
    for dt in (0.0, 0.05, 0.3):
        WW = expm(-1j * dt * H)
        step = np.where(rng.random(n) < 0.5, np.pi / 4, np.pi / 2)[:, None]
        a = rng.integers(-4, 5, (n, P)) * step
        same = rng.random(n) < 0.3                                       # a third of the pairs: candidate = reference (the start of a time step)
        b = np.where(same[:, None], a, rng.integers(-4, 5, (n, P)) * step)
        A = np.stack([ER.tensor(kind, D, v) for v in a])
        eng = EnergyEngine(D, max_batch=max(1024, n))
        for max_rounds in ((40, 60) if D <= 4 else (None,)):
            eta, rounds, st = eng.overlaps(A, b, WW, kind='params', ansatz=kind, tol=1e-13, max_rounds=max_rounds)
            worst, bad, nbad_local = 0.0, [], []
            for t in range(n):
                if st[t] not in (0, 4):
                    continue
                B = ER.tensor(kind, D, b[t])
                rho = ER.spectral_radius(O.transfer_matrix(np.tensordot(WW, O.merge(A[t], A[t]), [1, 0]), O.merge(B, B)))
                dev = abs(abs(eta[t]) - rho)
                worst = max(worst, dev)
                nbad_local.append(dev > 1e-8)
                if dev > 1e-8 and len(bad) < 3:
                    bad.append({'a': (a[t] / (np.pi / 4)).round(2).tolist(), 'b': (b[t] / (np.pi / 4)).round(2).tolist(), 'eta': abs(eta[t]), 'rho': rho, 'status': int(st[t]), 'rounds': int(rounds[t])})
            rec = {'D': D, 'P': P, 'dt': dt, 'max_rounds': max_rounds, 'pairs': n, 'n_bad': int(sum(nbad_local)), 'status': {int(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))}, 'worst': worst, 'bad': bad}
            print(json.dumps(rec), flush=True)
            out.append(rec)
        del eng
json.dump(out, open(os.path.join(R, 'gpurun_out', 'grid_overlaps_probe.json'), 'w'), indent=1)
