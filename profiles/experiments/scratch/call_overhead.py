"""wall time of one qmps_overlap_gradient / qmps_overlap_eval_ansatz round trip against the HIP-event time of its kernels"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import expm
import bench
from qmps_amd import EnergyEngine
D, T, P = 16, 256, 8
WW = expm(-0.05j * bench.tfim_h(1.0))
rng = np.random.default_rng(1)
ref = rng.standard_normal((T, P))
eng = EnergyEngine(D, T * (2 * P + 1))
eng.overlap_set_refs_params(0, ref, WW)
X = ref + 0.01 * rng.standard_normal((T, P))
eng.set_kernel_timing_period(1)
eng.overlap_gradient(0, X, warm=False)
ts, ks = [], []
for k in range(30):
    Xk = X + 1e-3 * rng.standard_normal(X.shape)
    t = time.perf_counter(); eng.overlap_gradient(0, Xk, warm=True); ts.append(time.perf_counter() - t)
    ks.append(eng.kernel_time(1)[0])
print('gradient: wall %.1f us, kernels (events) %.1f us' % (np.median(ts) * 1e6, np.median(ks) * 1e3))
eng.overlap_set_group(2)
cand = np.repeat(X, 2, axis=0) + 1e-3 * rng.standard_normal((2 * T, P))
eng.overlap_eval_params(0, cand, want_r=True)
ts, ks = [], []
for k in range(30):
    t = time.perf_counter(); eng.overlap_eval_params(0, cand + 1e-4 * rng.standard_normal(cand.shape), want_r=True, warm=True); ts.append(time.perf_counter() - t)
    ks.append(eng.kernel_time(1)[0])
print('ladder stage 1: wall %.1f us, overlap kernel (events) %.1f us' % (np.median(ts) * 1e6, np.median(ks) * 1e3))
t = time.perf_counter()
for k in range(200): eng.sync()
print('empty sync: %.1f us' % ((time.perf_counter() - t) / 200 * 1e6))
