"""Round 5 (verdict r04 item 7, a bounded attempt): the D = 4 energy-only kernel reading the environment as its 16 real Hermitian
coordinates (128 B per evaluation, 32 contiguous bytes per lane) instead of the stored complex 4 x 4 (256 B, a row AND a column per lane).
Needs profiles/experiments/r05/packed_env.patch applied (git apply) and a tuning build (make EXTRA=-DQMPS_DEBUG_KNOBS): QMPS_PACKED_ENV selects the
packed read inside qmps_energy_only_launch and the warm-started qmps_energy_launch.  NOT in the shipped library: measured, below the bar.
Prints parity of the energies and us per launch of both variants over 9 rotating batches of 65 536 evaluations."""
import os, sys, json
import numpy as np
sys.path.insert(0, '.')
import bench
from qmps_amd import EnergyEngine

D, B, R = 4, 65536, 9
A_all = np.concatenate([bench.haar_tensors(100 + k, D, B) for k in range(R)])
h = bench.tfim_h(1.0)
eng = EnergyEngine(D, R * B)
eng.set_tensors(A_all)
eng.set_hamiltonian(h)
for k in range(R):
    eng.set_window(k * B)
    eng.launch(B, max_iter=10000, tol=1e-13, solver='direct', store_env=True)
out = {}
for name in ('stored', 'packed', 'stored_again'):
    if name == 'packed':
        os.environ['QMPS_PACKED_ENV'] = '1'
    else:
        os.environ.pop('QMPS_PACKED_ENV', None)
    for k in range(2 * R):
        eng.set_window((k % R) * B)
        eng.launch_energy_only(B)
    eng.sync()
    eng.set_window(3 * B)
    eng.launch_energy_only(B)
    E = eng.results(B)[0].copy()
    n = 90
    ts = []
    for rep in range(5):
        eng.timer_begin()
        for k in range(n):
            eng.set_window((k % R) * B)
            eng.launch_energy_only(B)
        ts.append(eng.timer_end() / n * 1e3)
    out[name] = {'us_per_launch': ts, 'E': E}
# the warm-started full step (QMPS_FLAG_WARM_RESIDENT, nothing stored): every evaluation finds its converged environment, passes the
# acceptance step and skips the matrix build and the elimination
warm = {}
for name in ('stored', 'packed', 'stored_again'):
    if name == 'packed':
        os.environ['QMPS_PACKED_ENV'] = '1'
    else:
        os.environ.pop('QMPS_PACKED_ENV', None)

    def warm_step(k):
        eng.set_window((k % R) * B)
        eng.launch(B, max_iter=10000, tol=1e-13, solver='direct', store_env=False, accumulate_cost=True, warm_start=True)
        eng.cost_launch(B)
    for k in range(3 * R):
        warm_step(k)
    eng.sync()
    eng.set_window(3 * B)
    eng.launch(B, max_iter=10000, tol=1e-13, solver='direct', store_env=False, warm_start=True)
    Ew, itw, stw = eng.results(B)
    ts = []
    for rep in range(5):
        eng.timer_begin()
        for k in range(90):
            warm_step(k)
        ts.append(eng.timer_end() / 90 * 1e3)
    warm[name] = {'us_per_step': ts, 'E': Ew.copy(), 'iters_mean': float(itw.mean()), 'bad': int((stw != 0).sum())}
d = float(np.abs(out['stored']['E'] - out['packed']['E']).max())
print(json.dumps({'max_abs_dE_packed_vs_stored': d, 'us_stored': out['stored']['us_per_launch'], 'us_packed': out['packed']['us_per_launch'],
                  'us_stored_again': out['stored_again']['us_per_launch'],
                  'bytes_per_eval': {'stored': 776, 'packed': 648},
                  'warm_step': {'max_abs_dE_packed_vs_stored': float(np.abs(warm['stored']['E'] - warm['packed']['E']).max()),
                                'us_stored': warm['stored']['us_per_step'], 'us_packed': warm['packed']['us_per_step'], 'us_stored_again': warm['stored_again']['us_per_step'],
                                'iters_mean': [warm[k]['iters_mean'] for k in ('stored', 'packed')], 'bad': [warm[k]['bad'] for k in ('stored', 'packed')]}}))
