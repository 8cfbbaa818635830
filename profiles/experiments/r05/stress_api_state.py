"""Round 5: a fuzzer of the stateful energy API (qmps_set_states / _set_hamiltonian / _set_window / _energy_launch with every flag combination /
_energy_only_launch / _cost_launch / _get_cost / _get_energies) - random sequences of calls on ONE context, every result compared with a STATELESS
evaluation of the same tensors on a second context.  The context carries a dozen flags (resident environments, windows, the in-kernel cost
accumulator, pending partial sums, ...): a stale one shows up as energies of another window, a cost that is not the sum of its energies, an environment
that belongs to other tensors.  Contract respected by the fuzzer: energy-only launches only on windows whose environments were stored after the last
upload; a WARM launch may find any environment resident (a stale or foreign one must be REJECTED by the acceptance step, not believed).
Usage: python profiles/experiments/r05/stress_api_state.py [n_sequences] [seed] [D]"""
import sys, json, time
import numpy as np
sys.path.insert(0, '.')
import bench
from qmps_amd import EnergyEngine

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
D = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(seed)
B, R = 300 + 16 * int(rng.integers(0, 5)) + int(rng.integers(0, 16)), 3
eng = EnergyEngine(D, R * B)
ref = EnergyEngine(D, B)
solvers = ['direct', 'squaring', 'plain'] if D == 4 else ['squaring', 'plain']
tot = {'sequences': 0, 'ops': 0, 'checks': 0, 'max_dE': 0.0, 'max_dcost': 0.0}
bad, t0 = [], time.time()


def h_draw():
    nt = int(rng.integers(1, 3))
    hs = []
    for _ in range(nt):
        G = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
        hs.append(bench.tfim_h(float(rng.uniform(0.2, 2.0))) if rng.random() < 0.5 else (G + G.conj().T) / 2)
    return np.stack(hs)


for seq in range(n_seq):
    A_all = np.concatenate([bench.haar_tensors(int(rng.integers(1 << 30)), D, B) for _ in range(R)])
    h = h_draw()
    eng.set_tensors(A_all)
    eng.set_hamiltonian(h)
    env_valid = [False] * R
    env_solver = ['direct'] * R
    win = 0
    eng.set_window(0)
    log = []
    for op in range(24):
        tot['ops'] += 1
        r = rng.random()
        if r < 0.12:
            A_all = np.concatenate([bench.haar_tensors(int(rng.integers(1 << 30)), D, B) for _ in range(R)])
            eng.set_tensors(A_all)
            env_valid = [False] * R
            eng.set_window(win * B)
            log.append('set_tensors')
            continue
        if r < 0.22:
            h = h_draw()
            eng.set_hamiltonian(h)
            log.append(f'set_hamiltonian({len(h)})')
            continue
        if r < 0.40:
            win = int(rng.integers(R))
            eng.set_window(win * B)
            log.append(f'set_window({win})')
            continue
        Aw = A_all[win * B:(win + 1) * B]
        if r < 0.55 and env_valid[win]:
            eng.launch_energy_only(B)
            what = 'energy_only'
            with_cost = False
            ref.set_solver(env_solver[win])
        else:
            solver = str(rng.choice(solvers))
            direct = solver == 'direct' and D == 4
            store = bool(rng.integers(2)) or not direct
            warm = bool(rng.integers(2)) and any(env_valid)
            with_cost = bool(rng.integers(2)) and direct
            eng.launch(B, max_iter=10000, tol=1e-13, solver=solver, store_env=store, accumulate_cost=with_cost, warm_start=warm)
            if store:
                env_valid[win] = True
                env_solver[win] = solver
            elif not warm:
                env_valid = [False] * R      # (the library's rule: a launch that neither stores nor reads environments leaves NONE resident)
            what = f'launch({solver}, store={store}, warm={warm}, acc={with_cost})'
            ref.set_solver(solver)
        E_ref, _, st_ref = ref.energies(Aw, h)      # stateless, the same solver
        log.append(what)
        do_cost = with_cost or rng.random() < 0.5
        try:
            if do_cost:
                eng.cost_launch(B)
            E, it, st = eng.results(B)
            cost_now = eng.get_cost() if do_cost else None
        except Exception as e:
            bad.append({'seq': seq, 'op': op, 'what': what, 'error': str(e)[:160], 'history': log[-10:]})
            # a fresh context: the failed one may be in any state
            eng = EnergyEngine(D, R * B)
            eng.set_tensors(A_all); eng.set_hamiltonian(h); eng.set_window(win * B)
            env_valid = [False] * R
            continue
        tot['checks'] += 1
        ok = (st == 0) & (st_ref == 0)
        d = float(np.abs(E[ok] - E_ref[ok]).max()) if ok.any() else 0.0
        tot['max_dE'] = max(tot['max_dE'], d)
        # (a status that differs where the energies agree and one side says 'not converged' is the squaring solver's granularity - it stops at the last
        # power of two below max_iter, a warm start gets there, a cold one may not: D = 2, |lambda_2| ~ 0.995, 1 evaluation in ~2 000 - not a state bug)
        mism = (st == 0) != (st_ref == 0)
        if d > 1e-9 or (mism.any() and not np.all((st[mism] == 1) | (st_ref[mism] == 1))):
            bad.append({'seq': seq, 'op': op, 'what': what, 'max_dE': d, 'status_mismatch': int(((st == 0) != (st_ref == 0)).sum()), 'history': log[-8:]})
        if do_cost:
            cost = cost_now
            dc = float(np.abs(cost - E.sum(0)).max())
            tot['max_dcost'] = max(tot['max_dcost'], dc)
            if dc > 1e-8 * max(1.0, float(np.abs(E).sum())):
                bad.append({'seq': seq, 'op': op, 'what': what + ' + cost', 'cost': cost.tolist(), 'sum_E': E.sum(0).tolist(), 'history': log[-8:]})
    tot['sequences'] += 1
print(json.dumps({'seed': seed, 'D': D, 'B': B, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:8]}))
