import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
rng = np.random.default_rng(1); B = 65536
A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
eng = EnergyEngine(4, B); eng.set_tensors(A); eng.set_hamiltonian(O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
eng.set_kernel_timing_period(16)
for _ in range(600): eng.launch(B); eng.cost_launch(B)
eng.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300): eng.launch(B); eng.cost_launch(B)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print('issue %.1f us/step, until done %.1f us/step' % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
# tiny batch: pure host + launch overhead
eng2 = EnergyEngine(4, 64); eng2.set_tensors(A[:64]); eng2.set_hamiltonian(O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
eng2.set_kernel_timing_period(0)
for _ in range(200): eng2.launch(64); eng2.cost_launch(64)
eng2.sync()
t0 = time.perf_counter()
for _ in range(1000): eng2.launch(64); eng2.cost_launch(64)
t1 = time.perf_counter(); eng2.sync(); t2 = time.perf_counter()
print('B=64: issue %.1f us/step, until done %.1f us/step' % ((t1 - t0) / 1000 * 1e6, (t2 - t0) / 1000 * 1e6))
