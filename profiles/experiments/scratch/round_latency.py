"""Per-round latency of the D = 16 pair kernel (right + left power iteration, four waves per solve): every solve of a gradient batch
runs exactly R rounds (tolerance out of reach, Krylov hand-over off): wall time of the gradient call against R.
usage: QMPS_NO_KRYLOV=1 python profiles/experiments/scratch/round_latency.py [D]"""
import os, sys, time
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from qmps_amd import EnergyEngine, _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = 8 if D == 16 else 6
WW = expm(-1j * 0.05 * bench.tfim_h(1.0))
for T in (32, 256, 1024):
    X = np.random.default_rng(1).standard_normal((T, P))
    with EnergyEngine(D, T * (2 * P + 1)) as eng:
        eng.overlap_set_refs_params(_lib.ANSATZ_SHALLOW_CNOT, X, WW)
        eng.overlap_gradient(_lib.ANSATZ_SHALLOW_CNOT, X + 0.01, max_rounds=50, tol=1e-300, two_sided_f=True)
        res = []
        for R in (20, 100, 200):
            ts = []
            for rep in range(6):
                t0 = time.perf_counter()
                eng.overlap_gradient(_lib.ANSATZ_SHALLOW_CNOT, X + 0.01, max_rounds=R, tol=1e-300, warm=True, two_sided_f=True)
                ts.append(time.perf_counter() - t0)
            res.append((R, min(ts) * 1e6))
        per = (res[2][1] - res[0][1]) / (res[2][0] - res[0][0])
        print('D %d T %d: gradient call us at R = 20 / 100 / 200: %.0f / %.0f / %.0f -> %.2f us per round, fixed part %.0f us' %
              (D, T, res[0][1], res[1][1], res[2][1], per, res[0][1] - 20 * per))
