import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qmps_amd import EnergyEngine
from oracle import qmps_oracle as O
rng = np.random.default_rng(1)
for B in (65536, 32768, 16384, 131072):
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    eng = EnergyEngine(4, B)
    eng.set_tensors(A); eng.set_hamiltonian(O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))
    eng.launch(B); eng.sync()
    for _ in range(3): eng.launch_energy_only()
    eng.sync(); eng.timer_begin()
    for _ in range(50): eng.launch_energy_only()
    ms = eng.timer_end()
    print(B, 'energy-only pass: %.2f us per launch, %.2f ns per evaluation' % (ms / 50 * 1e3, ms / 50 * 1e6 / B))
    eng.close()
