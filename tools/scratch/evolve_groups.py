"""config 4 with the trajectories of one GPU split into G independent lock-step groups, one host thread each (ctypes releases the
GIL inside the C calls): the host algebra of one group overlaps the kernels of the others, and the latency-bound solve chains of
several groups share the chip"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import expm
import bench
from qmps_amd.new_time_evolve import LockstepEvolver
from qmps_amd.represent import ShallowCNOTStateTensor
D, P = 16, 8
WW = expm(-0.05j * bench.tfim_h(1.0))
for T_total, G in ((1024, 1), (1024, 2), (1024, 4), (2048, 4), (2048, 8), (4096, 8)):
    T = T_total // G
    evs = [LockstepEvolver(D, T, P, ShallowCNOTStateTensor, tol=1e-12, maxiter=30, first_rungs=2, carry_hessian=True, speculative=True) for _ in range(G)]
    Xs = [np.random.default_rng(7 + g).standard_normal((T, P)) for g in range(G)]
    steps, warm = 8, 3
    def run(g, n):
        for _ in range(n):
            Xs[g] = evs[g].step(Xs[g], WW)['x']
    for n in (warm, steps):
        th = [threading.Thread(target=run, args=(g, n)) for g in range(G)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
    print(f'T={T_total} groups={G}: {T_total * steps / dt:.0f} trajectory steps/s, {dt / steps * 1e3:.2f} ms per time step', flush=True)
    for e in evs: e.close()
