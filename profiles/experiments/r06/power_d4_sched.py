"""Work distribution of env_power_d4_kernel: waves per SIMD x static first block x chunk size (temporary knobs of a tuning build).
  (Historical: QMPS_TMP_FB / QMPS_TMP_CHUNK existed in the tuning build of that afternoon only; the shipped kernel has first block = half the batch, chunk = 8.  QMPS_POWER_WAVES needs a -DQMPS_DEBUG_KNOBS build.)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import qmps_oracle as O          # noqa: E402
from qmps_amd import EnergyEngine            # noqa: E402
rng = np.random.default_rng(20241022)
B = 65536
A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
out = {}
with EnergyEngine(4, B) as eng:
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    for w in (2, 3, 4, 5, 6, 8):
        for fb in (4, 8, 16):
            for ch in (4, 8, 16):
                os.environ['QMPS_POWER_WAVES'] = str(w); os.environ['QMPS_TMP_FB'] = str(fb); os.environ['QMPS_TMP_CHUNK'] = str(ch)
                ts = []
                for rep in range(3):
                    eng.launch(B, max_iter=10000, tol=1e-13, solver='plain', store_env=True)
                    eng.sync()
                    eng.timer_begin()
                    for _ in range(10):
                        eng.launch(B, max_iter=10000, tol=1e-13, solver='plain', store_env=True)
                    ts.append(eng.timer_end() / 10)
                out[f'w{w}_fb{fb}_ch{ch}'] = round(float(np.median(ts)), 4)
print(json.dumps(out))
