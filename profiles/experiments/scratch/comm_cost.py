import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from qmps_amd import EnergyEngine
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
B = 65536
eng = EnergyEngine(4, B)
A = bench.haar_tensors(1, 4, B); eng.set_tensors(A); eng.set_hamiltonian(bench.tfim_h())
def run(label, n=50, launch=True, cost=True):
    for _ in range(3):
        if launch: eng.launch()
        if cost: eng.cost_launch()
    eng.sync(); t = time.perf_counter()
    for _ in range(n):
        if launch: eng.launch()
        if cost: eng.cost_launch()
    eng.sync(); dt = (time.perf_counter() - t) / n * 1e6
    print(f'{label}: {dt:.1f} us per step')
run('no comm: launch+cost'); run('no comm: launch only', cost=False); run('no comm: cost only', launch=False)
eng.comm_init(EnergyEngine.comm_unique_id(), 0, 1)
run('comm(1): launch+cost'); run('comm(1): cost only', launch=False)
