"""steps alternated over two contexts (two HIP streams): does the next kernel's load burst overlap the previous kernel's tail?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from qmps_amd import EnergyEngine
B, R = 65536, 5
engs = [EnergyEngine(4, R * B) for _ in range(2)]
for i, e in enumerate(engs):
    e.set_tensors(np.concatenate([bench.haar_tensors(10 * i + k, 4, B) for k in range(R)])); e.set_hamiltonian(bench.tfim_h()); e.set_kernel_timing_period(0)
cnt = [0]
def step(n_eng):
    e = engs[cnt[0] % n_eng]; w = (cnt[0] // n_eng) % R; cnt[0] += 1
    e.set_window(w * B); e.launch(B, solver='direct', store_env=False, accumulate_cost=True); e.cost_launch(B)
for n_eng in (1, 2, 1, 2):
    for _ in range(10): engs[0].probe_fp64_tflops()
    for _ in range(600): step(n_eng)
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for _ in range(900): step(n_eng)
    for e in engs: e.sync()
    dt = (time.perf_counter() - t0) / 900
    print(f'{n_eng} stream(s): {dt*1e6:.2f} us per step, {B/dt:.4e} evals/s', flush=True)
