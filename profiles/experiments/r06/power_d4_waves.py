"""Persistent-grid size of env_power_d4_kernel: waves per SIMD x batch size, five timed blocks of ten launches each (median, min).
  (QMPS_POWER_WAVES is a tuning knob: -DQMPS_DEBUG_KNOBS builds only; the shipped default is five waves per SIMD.)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import qmps_oracle as O          # noqa: E402
from qmps_amd import EnergyEngine            # noqa: E402
rng = np.random.default_rng(20241022)
Bfull = 65536
A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, Bfull))
h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
out = {}
with EnergyEngine(4, Bfull) as eng:
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    for B in (65536, 16384, 4096):
        for w in os.environ.get('SWEEP_WAVES', '2,3,4,5,6').split(','):
            os.environ['QMPS_POWER_WAVES'] = w
            ts = []
            for rep in range(5):
                eng.launch(B, max_iter=10000, tol=1e-13, solver='plain', store_env=True)
                eng.sync()
                eng.timer_begin()
                for _ in range(10):
                    eng.launch(B, max_iter=10000, tol=1e-13, solver='plain', store_env=True)
                ts.append(eng.timer_end() / 10)
            out[f'B{B}_w{w}'] = {'median_ms': float(np.median(ts)), 'min_ms': float(min(ts)), 'max_ms': float(max(ts))}
print(json.dumps(out, indent=1))
