"""Experiment: the T trajectories of a D = 16 time evolution as K independent lock-step groups, one context + one host thread each
(ctypes releases the GIL during qmps_evolve_bfgs): do the groups' host gaps and straggler iterations overlap on the device?
usage: python profiles/experiments/scratch/evolve_groups.py T K [steps]"""
import sys, time, threading
import numpy as np
from scipy.linalg import expm
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from qmps_amd.new_time_evolve import LockstepEvolver
from qmps_amd.represent import ShallowCNOTStateTensor
import bench

T, K = int(sys.argv[1]), int(sys.argv[2])
KC = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # groups inside the library (qmps_set_evolve_groups)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
D = int(os.environ.get("EXP_D", "16")); P = 8 if D == 16 else 6
WW = expm(-1j * 0.05 * bench.tfim_h(1.0))
X = np.random.default_rng(20241022).standard_normal((T, P))
bounds = [(k * T // K, (k + 1) * T // K) for k in range(K)]
evs = [LockstepEvolver(D, b - a, P, ShallowCNOTStateTensor, tol=1e-12, maxiter=200, carry_hessian=True, speculative=True) for a, b in bounds]
out = [None] * K
tg = [0.0] * K
for e in evs:
    e.fg.eng.set_evolve_groups(KC)

def run(k, n):
    a, b = bounds[k]
    tt = time.perf_counter()
    out[k] = evs[k].steps(X[a:b], WW, n, counters=False)
    tg[k] = (time.perf_counter() - tt) * 1e3

def all_groups(n):
    th = [threading.Thread(target=run, args=(k, n)) for k in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    for k, (a, b) in enumerate(bounds):
        X[a:b] = out[k]['x']

all_groups(3)
t0 = time.perf_counter()
all_groups(steps)
for e in evs: e.fg.eng.sync()
dt = time.perf_counter() - t0
print('python thread times ms', ['%.3f' % t for t in tg], 'total %.3f' % (dt * 1e3))
print('T %d python threads %d library groups %d: %.4g trajectory steps/s, %.3f ms per time step, nit %s' % (T, K, KC, T * steps / dt, dt / steps * 1e3, [list(map(int, o['nit'])) for o in out][:2]))
