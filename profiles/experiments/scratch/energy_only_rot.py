"""energy-only contraction chain over 9 rotating resident batches (tensors + environments = 432 MiB > Infinity Cache)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from qmps_amd import EnergyEngine
B, R = 65536, 9
eng = EnergyEngine(4, R * B)
A = np.concatenate([bench.haar_tensors(k, 4, B) for k in range(R)]); eng.set_tensors(A); eng.set_hamiltonian(bench.tfim_h())
for k in range(R):
    eng.set_window(k * B); eng.launch(B, store_env=True)
eng.set_window(0); E0, _, _ = eng.results(B)
for _ in range(10): eng.probe_fp64_tflops()
for k in range(4 * R):
    eng.set_window((k % R) * B); eng.launch_energy_only(B)
eng.sync(); eng.timer_begin()
n = 180
for k in range(n):
    eng.set_window((k % R) * B); eng.launch_energy_only(B)
us = eng.timer_end() / n * 1e3
eng.set_window(0); E1, _, _ = eng.results(B)
print(f'energy-only: {us:.2f} us per launch, {B / us * 1e-3:.3f} G evals/s, {B * 776 / us * 1e-6:.2f} TB/s on 776 B ({B * 776 / us * 1e-6 / 8 * 100:.1f} % of 8 TB/s), {B * 520 / us * 1e-6 / 8 * 100:.1f} % on 520 B; max |dE| vs solve {np.abs(E1 - E0).max():.2e}')
