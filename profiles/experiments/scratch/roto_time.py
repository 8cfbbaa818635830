import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from qmps_amd import EnergyEngine, _lib as L
from qmps_amd.ground_state import Hamiltonian
h = Hamiltonian({'ZZ': -1, 'X': 1}).to_matrix()
for D, R, P in ((2, 1365, 2), (2, 1365, 8), (4, 21845, 4)):
    eng = EnergyEngine(D, 3 * R)
    eng.set_hamiltonian(h)
    P0 = np.random.default_rng(0).standard_normal((R, P))
    eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, P0, 1)
    t = time.perf_counter(); hist, _ = eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, P0, 20); dt = time.perf_counter() - t
    print(f'D={D} R={R} P={P} graph={"QMPS_NO_GRAPH" not in os.environ}: {dt / (20 * P) * 1e6:.1f} us per parameter update (3R = {3*R} evals), best E {np.nanmin(hist):.6f}')
    eng.close()
