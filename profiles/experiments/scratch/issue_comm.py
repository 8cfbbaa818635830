"""host issue time vs completion time per step of the accumulate-cost path, without and with a communicator of one rank"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from qmps_amd import EnergyEngine
B = 65536; R = 9
eng = EnergyEngine(4, R * B)
A = np.concatenate([bench.haar_tensors(k, 4, B) for k in range(R)]); eng.set_tensors(A); eng.set_hamiltonian(bench.tfim_h())
eng.set_kernel_timing_period(0)
cnt = [0]
def step():
    eng.set_window((cnt[0] % R) * B); cnt[0] += 1
    eng.launch(B, solver='direct', store_env=False, accumulate_cost=True)
    eng.cost_launch(B)
def run(label, n=600):
    for _ in range(10): eng.probe_fp64_tflops()
    for _ in range(600): step()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    print(f'{label}: issue {(t1 - t0) / n * 1e6:.1f} us/step, until done {(t2 - t0) / n * 1e6:.1f} us/step', flush=True)
run('no comm')
eng.comm_init(EnergyEngine.comm_unique_id(), 0, 1)
run('comm(1)')
run('comm(1) again')
