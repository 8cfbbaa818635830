"""Where does a launch of env_power_d4_kernel spend its time?  (i) capped iteration count: pure throughput; (ii) a handful of evaluations: the
latency of one power step."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import qmps_oracle as O          # noqa: E402
from qmps_amd import EnergyEngine            # noqa: E402
rng = np.random.default_rng(20241022)
Bfull = 65536
A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, Bfull))
h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
out = []
with EnergyEngine(4, Bfull) as eng:
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    for B, max_iter in ((65536, 10000), (65536, 32), (65536, 64), (65536, 128), (65536, 256), (4, 10000), (4, 1000), (64, 1000), (1024, 1000), (4096, 1000), (16384, 1000), (16384, 100)):
        for _ in range(2):
            eng.launch(B, max_iter=max_iter, tol=1e-13 if max_iter == 10000 else 1e-300, solver='plain', store_env=True)
        eng.sync()
        eng.timer_begin()
        n = 10
        for _ in range(n):
            eng.launch(B, max_iter=max_iter, tol=1e-13 if max_iter == 10000 else 1e-300, solver='plain', store_env=True)
        ms = eng.timer_end() / n
        _, it, st = eng.results(B)
        out.append({'B': B, 'max_iter': max_iter, 'ms': ms, 'mean_iters': float(it.mean()), 'max_iters': int(it.max()),
                    'us_per_step_of_the_longest': 1e3 * ms / it.max(), 'ns_per_eval_step': 1e6 * ms / it.sum()})
print(json.dumps(out, indent=1))
