"""Round 5: randomised stress of the device rotosolve drivers (qmps_rotosolve, qmps_double_rotosolve; qmps/rotosolve.py:154-181, qmps/tools.py:422-457) with
starting angles on the grid {0, +-pi/4, +-pi/2, pi} (product states, rank-deficient environments, flat sinusoids) mixed with random ones.  Trajectories
are not compared (at such points atan2(0, 0) may move a parameter along a flat direction); what must hold whatever the path:
  * the energy the driver reports for its final parameters is the oracle's energy AT those parameters (where the environment is unique and positive);
    (NOT checked: monotone sweeps - an angle shared by several gates, rz / rx on every qubit of a layer, makes the energy a multi-frequency function of
    it, the one-sinusoid update is a heuristic there, in the reference too, and sweeps do raise the energy);
  * no NaN, parameters finite.
Usage: python profiles/experiments/r05/stress_rotosolve.py [n_cases] [seed]"""
import sys, json, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine
import evolve_replay as ER

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
GRID = np.array([0.0, np.pi / 4, -np.pi / 4, np.pi / 2, -np.pi / 2, np.pi])
H = {'tfim': O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0}), 'xxz': O.hamiltonian_matrix({'XX': 1.0, 'YY': 1.0, 'ZZ': 0.5})}
engines = {}


def oracle_energy(kind, D, p, h):
    """Energy at p where the environment is unique, positive and VERIFIED: dense eigenvector polished by power steps, kept only if it satisfies the
    fixed-point equation to 1e-12 (numpy's eig alone is not reliable on defective transfer matrices; a slowly converging polish is not either)."""
    A = ER.tensor(kind, D, p)
    w, v = np.linalg.eig(O.transfer_matrix(A))
    order = np.argsort(-np.abs(w))
    if abs(w[order[1]]) > (1 - 1e-6) * abs(w[order[0]]):
        return None
    r = v[:, order[0]].reshape(D, D)
    r = r / np.trace(r)
    res = 1.0
    for _ in range(2000):
        rn = O.apply_transfer(A, r)
        rn = rn / np.trace(rn)
        res = np.abs(rn - r).max()
        r = rn
        if res < 1e-15:
            break
    if not res < 1e-12:
        return None
    r = (r + r.conj().T) / 2
    if np.linalg.eigvalsh(r).min() < 1e-9:
        return None
    return O.energy_closed_form(A, h, r)


tot = {'runs': 0, 'restarts': 0, 'final_checked': 0, 'max_dE_final': 0.0, 'nan_energies': 0}
bad, t0 = [], time.time()
for case in range(n_cases):
    D = int(rng.choice([2, 4, 8]))
    kind = int(rng.choice([0, 1, 3]))
    depth = int(rng.integers(1, 4))
    P = (3 if kind == 3 else 2) * depth
    R = 32
    X0 = rng.standard_normal((R, P))
    X0 = np.where(rng.random((R, P)) < (0.9 if rng.random() < 0.5 else 0.4), GRID[rng.integers(0, len(GRID), size=(R, P))], X0)
    hname = str(rng.choice(['tfim', 'xxz']))
    if D not in engines:
        engines[D] = EnergyEngine(D, 4096)
    eng = engines[D]
    eng.set_hamiltonian(H[hname])
    sweeps = 3
    for double in (False, True):
        try:
            es, p = (eng.double_rotosolve if double else eng.rotosolve)(kind, X0, sweeps)
        except Exception as e:
            bad.append({'case': case, 'D': D, 'kind': kind, 'double': double, 'error': str(e)[:200]})
            continue
        tot['runs'] += 1
        tot['restarts'] += R
        if not np.all(np.isfinite(p)):
            bad.append({'case': case, 'D': D, 'kind': kind, 'double': double, 'what': 'non-finite parameters'})
        tot['nan_energies'] += int(np.isnan(es).sum())
        for r in range(R):
            if np.isnan(es[-1, r]):
                continue
            e_or = oracle_energy(kind, D, p[r], H[hname])
            if e_or is None:
                continue
            tot['final_checked'] += 1
            d = abs(e_or - es[-1, r])
            tot['max_dE_final'] = max(tot['max_dE_final'], float(d))
            if not d < 1e-8:
                bad.append({'case': case, 'D': D, 'kind': kind, 'double': double, 'what': 'reported final energy is not the energy of the final parameters', 'restart': r,
                            'reported': float(es[-1, r]), 'oracle': float(e_or), 'x0': X0[r].tolist(), 'x': p[r].tolist()})
print(json.dumps({'cases': n_cases, 'seed': seed, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:8]}))
