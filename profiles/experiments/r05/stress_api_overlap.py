"""Round 5: a fuzzer of the stateful OVERLAP API (qmps_overlap_set_refs_ansatz / _set_group / _set_active / _eval_ansatz with and without resident warm
starts / _gradient with and without warm starts / energy launches in between on the same context) - random call sequences on ONE context, every result
compared with a stateless evaluation on a second context.  Looks for stale one-shot state: a mask that survives its launch, a group size that leaks into
the next call, warm starts from another batch's fixed points, references overwritten by an energy launch.
Usage: python profiles/experiments/r05/stress_api_overlap.py [n_sequences] [seed] [D]"""
import sys, json, time
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, '.')
import bench
from qmps_amd import EnergyEngine
from qmps_amd.ground_state import Hamiltonian

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(seed)
Hm = Hamiltonian({'ZZ': -1.0, 'X': 1.0}).to_matrix()
kind = 0
P = 2 * max(1, int(np.log2(D)))
eng = EnergyEngine(D, 4096)
ref = EnergyEngine(D, 4096)
tot = {'sequences': 0, 'ops': 0, 'checks': 0, 'max_df': 0.0, 'max_dg': 0.0}
bad, t0 = [], time.time()
TOL = 1e-12
for seq in range(n_seq):
    T = int(rng.integers(2, 24))
    refp = rng.standard_normal((T, P))
    WW = expm(-1j * float(rng.choice([0.02, 0.05, 0.1])) * Hm)
    eng.overlap_set_refs_params(kind, refp, WW)
    prev_f = {}
    log = ['set_refs']
    for op in range(16):
        tot['ops'] += 1
        r = rng.random()
        try:
            if r < 0.10:
                T = int(rng.integers(2, 24))
                refp = rng.standard_normal((T, P))
                WW = expm(-1j * float(rng.choice([0.02, 0.05, 0.1])) * Hm)
                eng.overlap_set_refs_params(kind, refp, WW)
                prev_f = {}
                log.append('set_refs')
                continue
            if r < 0.20:
                # an energy launch on the same context in between (the drivers share contexts with the optimisers' energy calls)
                Bn = int(rng.integers(8, 200))
                A = bench.haar_tensors(int(rng.integers(1 << 30)), D, Bn)
                E, _, st = eng.energies(A, bench.tfim_h(1.0))
                E2, _, st2 = ref.energies(A, bench.tfim_h(1.0))
                tot['checks'] += 1
                if np.abs(E[(st == 0) & (st2 == 0)] - E2[(st == 0) & (st2 == 0)]).max() > 1e-10:
                    bad.append({'seq': seq, 'op': op, 'what': 'energies in between', 'history': log[-6:]})
                eng.overlap_set_refs_params(kind, refp, WW)      # (documented: an energy call overwrites the resident tensors; references are set again)
                prev_f = {}
                log.append('energies + set_refs')
                continue
            near = refp + 10.0 ** rng.uniform(-5, -1) * rng.standard_normal((T, P))
            if r < 0.60:
                G = int(rng.integers(1, 6))
                cand = np.repeat(near, G, axis=0) + 1e-3 * rng.standard_normal((T * G, P))
                mask = rng.random(T) < 0.7 if rng.random() < 0.5 else None
                warm = bool(rng.integers(2)) and D >= 8 and prev_f.get('stored_shape') == (T, G)      # (a warm start needs the slots of a launch that kept its fixed points)
                want_r = bool(rng.integers(2))
                eng.overlap_set_group(G)
                if mask is not None:
                    eng.overlap_set_active(mask)
                f, st = eng.overlap_eval_params(kind, cand, tol=TOL, warm=warm, want_r=want_r)
                eng.overlap_set_group(0)
                what = f'eval(G={G}, mask={mask is not None}, warm={warm})'
                # stateless: every candidate against its trajectory's reference
                ref.overlap_set_refs_params(kind, refp, WW)
                ref.overlap_set_group(G)
                fr, str_ = ref.overlap_eval_params(kind, cand, tol=TOL)
                ref.overlap_set_group(0)
                act = np.repeat(mask, G) if mask is not None else np.ones(T * G, bool)
                ok = act & (st == 0) & (str_ == 0)
                tot['checks'] += 1
                d = float(np.abs(f[ok] - fr[ok]).max()) if ok.any() else 0.0
                tot['max_df'] = max(tot['max_df'], d)
                if d > 1e-9 or not np.array_equal((st == 0)[act], (str_ == 0)[act]):
                    bad.append({'seq': seq, 'op': op, 'what': what, 'max_df': d, 'status_mismatch': int(((st == 0) != (str_ == 0))[act].sum()), 'T': T, 'history': log[-6:]})
                prev_f = {'stored_shape': (T, G)} if (want_r or warm) else {}
                log.append(what)
            elif D >= 4:
                mask = rng.random(T) < 0.7 if rng.random() < 0.4 else None
                warm = bool(rng.integers(2)) and prev_f.get('grad_T') == T
                two = bool(rng.integers(2))
                if mask is not None:
                    eng.overlap_set_active(mask)
                f, g, st = eng.overlap_gradient(kind, near, tol=TOL, warm=warm, two_sided_f=two)
                what = f'gradient(mask={mask is not None}, warm={warm}, two_sided_f={two})'
                ref.overlap_set_refs_params(kind, refp, WW)
                fr, gr, str_ = ref.overlap_gradient(kind, near, tol=TOL, two_sided_f=two)
                act = mask if mask is not None else np.ones(T, bool)
                ok = act & (st == 0) & (str_ == 0)
                tot['checks'] += 1
                d = float(np.abs(f[ok] - fr[ok]).max()) if ok.any() else 0.0
                dg = float(np.abs(g[ok] - gr[ok]).max()) if ok.any() else 0.0
                tot['max_df'] = max(tot['max_df'], d)
                tot['max_dg'] = max(tot['max_dg'], dg)
                if d > 1e-9 or dg > 1e-6 or not np.array_equal((st == 0)[act], (str_ == 0)[act]):
                    bad.append({'seq': seq, 'op': op, 'what': what, 'max_df': d, 'max_dg': dg, 'status_mismatch': int(((st == 0) != (str_ == 0))[act].sum()), 'T': T, 'history': log[-6:]})
                prev_f = {'grad_T': T}
                log.append(what)
        except Exception as e:
            bad.append({'seq': seq, 'op': op, 'error': str(e)[:200], 'history': log[-6:]})
            eng = EnergyEngine(D, 4096)
            eng.overlap_set_refs_params(kind, refp, WW)
            prev_f = {}
    tot['sequences'] += 1
print(json.dumps({'seed': seed, 'D': D, **tot, 'anomalies': len(bad), 'seconds': time.time() - t0, 'bad': bad[:8]}))
