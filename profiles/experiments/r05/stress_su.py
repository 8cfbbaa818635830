"""Round 5: randomised stress of the SU(2D) / U4 parameterisations on the device (a-8, a-9: qmps_energy_batch_su, qmps_su_unitaries, qmps_cell2_energy_batch_su)
against the host's matrix exponential (qmps_amd.ground_state.SU: the documented generator convention) + the oracle's energy: random parameters of every
scale (1e-8 ... 30: tiny exponents, and ones that need many squarings), zeros (the identity: a product state), single generators.
Usage: python profiles/experiments/r05/stress_su.py [n_batches] [seed]"""
import sys, json, time
import numpy as np
sys.path.insert(0, '.')
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine
from qmps_amd.ground_state import SU, U4

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
H = O.hamiltonian_matrix({'ZZ': -1.0, 'X': 1.0})
tot = {'unitaries': 0, 'max_dU': 0.0, 'max_unitarity': 0.0, 'energies': 0, 'checked': 0, 'max_dE': 0.0, 'cell': 0, 'cell_max_dE': 0.0}
bad, t0 = [], time.time()
engines = {}


def verified_env(A):
    D = A.shape[1]
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
    return r if np.linalg.eigvalsh(r).min() > 1e-9 else None


for batch in range(n_batches):
    D = int(rng.choice([2, 4, 8]))
    N = 2 * D
    npar = N * N - 1
    B = 24 if D == 8 else 64
    scale = 10.0 ** rng.uniform(-8, 1.5, size=(B, 1))
    Pm = scale * rng.standard_normal((B, npar))
    Pm[0] = 0.0
    Pm[1] = 0.0
    Pm[1, int(rng.integers(npar))] = float(rng.uniform(-20, 20))
    if D not in engines:
        engines[D] = EnergyEngine(D, 1024)
    eng = engines[D]
    U_host = np.stack([SU(p, N) for p in Pm])
    U_dev = eng.su_unitaries(Pm, N)
    if U_dev is not None:
        tot['unitaries'] += B
        tot['max_dU'] = max(tot['max_dU'], float(np.abs(U_dev - U_host).max()))
        tot['max_unitarity'] = max(tot['max_unitarity'], float(np.abs(np.einsum('bij,bkj->bik', U_dev, U_dev.conj()) - np.eye(N)).max()))
        if np.abs(U_dev - U_host).max() > 1e-10:
            k = int(np.argmax(np.abs(U_dev - U_host).reshape(B, -1).max(1)))
            bad.append({'what': 'SU unitary differs from the host exponential', 'D': D, 'k': k, 'scale': float(scale[k, 0]), 'd': float(np.abs(U_dev[k] - U_host[k]).max())})
    E, it, st = eng.energies_from_su(Pm, H)
    for k in range(B):
        tot['energies'] += 1
        A = O.unitary_to_tensor(U_host[k])
        r = verified_env(A)
        if r is None or st[k] != 0:
            continue
        tot['checked'] += 1
        d = abs(E[k, 0] - O.energy_closed_form(A, H, r))
        tot['max_dE'] = max(tot['max_dE'], float(d))
        if d > 1e-9:
            bad.append({'what': 'SU energy', 'D': D, 'k': k, 'dE': float(d), 'scale': float(scale[k, 0])})
    if D == 2:
        P30 = 10.0 ** rng.uniform(-6, 1.2, size=(B, 1)) * rng.standard_normal((B, 30))
        Ec, _, stc = eng.cell2_energies_su(P30, H)
        for k in range(B):
            if stc[k] != 0:
                continue
            A1, A2 = O.unitary_to_tensor(U4(P30[k, :15])), O.unitary_to_tensor(U4(P30[k, 15:]))
            r12, r21 = verified_env(O.merge(A1, A2)), verified_env(O.merge(A2, A1))
            if r12 is None or r21 is None:
                continue
            tot['cell'] += 1
            d = abs(Ec[k, 0] - O.two_site_cell_energy_closed(A1, A2, H, r12, r21))
            tot['cell_max_dE'] = max(tot['cell_max_dE'], float(d))
            if d > 1e-9:
                bad.append({'what': 'cell2 SU energy', 'k': k, 'dE': float(d)})
print(json.dumps({'batches': n_batches, 'seed': seed, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:8]}))
