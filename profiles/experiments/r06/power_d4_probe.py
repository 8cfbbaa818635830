"""A/B of the D = 4 plain power iteration (QMPS_ENV_POWER): the persistent kernel of round 6 (a 16-lane DPP row per evaluation; labelled 'quad' below after its first version) against the lane-per-evaluation
kernel of rounds 1-5 (QMPS_POWER_LANE=1) on the headline's tensors: iterates, iteration counts, statuses, energies, time per launch."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import c_oracle, qmps_oracle as O          # noqa: E402
from qmps_amd import EnergyEngine                      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(20241022)
A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
out = {}
res = {}
with EnergyEngine(4, B) as eng:
    eng.set_tensors(A)
    eng.set_hamiltonian(h)
    names = ['quad', 'lane'] + [f'quad_w{w}' for w in os.environ.get('SWEEP_WAVES', '').split(',') if w]
    for name in names:
        os.environ.pop('QMPS_POWER_WAVES', None)
        if name == 'lane':
            os.environ['QMPS_POWER_LANE'] = '1'
        else:
            os.environ.pop('QMPS_POWER_LANE', None)
            if name.startswith('quad_w'):
                os.environ['QMPS_POWER_WAVES'] = name[6:]
        for _ in range(2):
            eng.launch(B, max_iter=10000, tol=1e-13, solver='plain', store_env=True)
        eng.sync()
        eng.timer_begin()
        n = 10
        for _ in range(n):
            eng.launch(B, max_iter=10000, tol=1e-13, solver='plain', store_env=True)
        ms = eng.timer_end() / n
        E, it, st = eng.results(B)
        r = eng.environments(B) if hasattr(eng, 'environments') else None
        res[name] = (E.copy(), it.copy(), st.copy(), None if r is None else r.copy())
        out[name] = {'ms_per_launch': ms, 'evals_per_s': B / (ms * 1e-3), 'mean_iters': float(it.mean()), 'max_iters': int(it.max()), 'status_nonzero': int((st != 0).sum())}
    os.environ.pop('QMPS_POWER_LANE', None)
Eq, iq, sq, rq = res['quad']
El, il, sl, rl = res['lane']
out['max_abs_dE_quad_vs_lane'] = float(np.abs(Eq - El).max())
out['iters_differ'] = int((iq != il).sum())
out['max_abs_diters'] = int(np.abs(iq - il).max())
out['status_differ'] = int((sq != sl).sum())
if rq is not None:
    out['max_abs_dr'] = float(np.abs(rq - rl).max())
n_or = min(B, 4096)
ref = c_oracle.energy_batch(A[:n_or], h)
out['max_abs_dE_quad_vs_oracle'] = float(np.abs(Eq[:n_or, 0] - np.asarray(ref['E']).reshape(n_or, -1)[:, 0]).max())
out['iters_differ_vs_oracle'] = int((iq[:n_or] != ref['iters']).sum())
print(json.dumps(out, indent=1))
