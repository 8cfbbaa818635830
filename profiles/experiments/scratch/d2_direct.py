import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine, _lib as L
rng = np.random.default_rng(3)
B = 65536
A = O.unitary_to_tensor(O.haar_unitaries(rng, 4, B))
h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
eng = EnergyEngine(2, B); eng.set_hamiltonian(h); eng.set_tensors(A)
res = {}
for solver in ('squaring', 'direct'):
    eng.launch(B, solver=solver); E, it, st = eng.results(B); r = eng.environments(B)
    for _ in range(10): eng.probe_fp64_tflops()
    for _ in range(100): eng.launch(B, solver=solver, accumulate_cost=True); eng.cost_launch(B)
    eng.sync(); eng.timer_begin()
    for _ in range(200): eng.launch(B, solver=solver, accumulate_cost=True); eng.cost_launch(B)
    ms = eng.timer_end() / 200
    res[solver] = (E, it, st, r)
    print(solver, f'{ms*1e3:.2f} us per step', 'status', np.bincount(st, minlength=3), 'mean it', it.mean(), 'max it', it.max())
E0, it0, st0, r0 = res['squaring']; E1, it1, st1, r1 = res['direct']
both = (st0 == 0) & (st1 == 0)
print('max |dE| both ok', np.abs(E0 - E1)[both].max(), 'max |dr|', np.abs(r0 - r1)[both].max(), 'status differs', int((st0 != st1).sum()))
ref = [O.energy_closed_form(A[b], h[0]) for b in range(0, 2000, 7)]
print('vs dense eig', np.abs(E1[0:2000:7, 0] - np.array(ref)).max())
# rotosolve through the fused D = 2 kernel with both solvers
P0 = rng.standard_normal((1365, 2))
for solver in ('squaring', 'direct'):
    eng.set_solver(solver)
    eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, P0, 2)
    t = time.perf_counter(); hist, pf = eng.rotosolve(L.ANSATZ_SHALLOW_CNOT, P0, 200); dt = time.perf_counter() - t
    print('rotosolve', solver, f'{dt / 400 * 1e6:.2f} us per update', float(np.nanmean(hist[-1])))
    res['p' + solver] = pf
print('params differ', np.abs(res['psquaring'] - res['pdirect']).max())
