#!/usr/bin/env python3
"""bench.py - two-site energy evaluations per second (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
the environment solve + two-site energy over B evaluations (at D = 4 ONE fused kernel: direct fixed-point
solve, acceptance power step, energies), the device-side sum of the batch's energies and - for N > 1 - the
path's single exchange step, ONE RCCL all-reduce of the summed cost over xGMI PER STEP.
Workload = BASELINE.json configs[2]: TFIM g=1, D=4, B=65536 Haar-random state unitaries, tol 1e-13.

  --rotate R   R distinct resident batches are cycled, one per step, so that no step re-reads what an earlier
               one left in the 256 MiB Infinity Cache (default: the smallest R with R x batch bytes > 256 MiB).
  --scaling    weak: --batch evaluations per GPU (default);  strong: --batch is the GLOBAL batch, rank r
               evaluates the contiguous shard qmps_amd.dist.shard_bounds(B, r, N)  (B/G per GPU).

The product path uses no PyTorch: kernels, streams, events and the RCCL communicator live in libqmps_hip.so
(ctypes).  For N > 1 torch.distributed (gloo, CPU) is launcher plumbing only: rendezvous, the barriers around
the timed region, the broadcast of the RCCL unique id and the max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line (DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); already exported on the
# GPU boxes, set here too in case a launcher drops the environment
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    # N ranks share one host: keep each rank's BLAS/OpenMP pools small while it draws its synthetic inputs
    for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ.setdefault(_v, '4')

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# inputs, byte / flop counts, CPU baselines, launcher plumbing (benchlib/common.py) and the three other workloads; THIS file is the
# command line, the headline energy workload with its timed region (main), and the other_configs leg of the default run
from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.common import FP64_PEAK_TFLOPS, HBM_PEAK_GBPS, MALL_MIB, ROOT, _one_blas_thread, _ref_chunk  # noqa: E402,F401
from benchlib.overlap import main_overlap, nearby_tensors  # noqa: E402,F401
from benchlib.evolve import main_evolve, evolve_cpu_baseline  # noqa: E402,F401
from benchlib.rotosolve import main_rotosolve, rotosolve_shard_plan  # noqa: E402,F401


def other_configs(args, budget_s=60.0):
    """BASELINE.json configs[1], [3], [4] as the optimiser workloads they name, on the driver's line: each entry is the JSON line
    `bench.py --workload ...` prints for that configuration (value, ms_per_step, roofline, cpu_baseline), run in this process on
    contexts of their own after the headline's timed region, with bounded step counts; an entry that would start after the
    budget is skipped and says so."""
    import copy
    t0 = time.perf_counter()
    plan = [
        ('config1_rotosolve_D2_b4096', dict(workload='rotosolve', D=2, batch=4096, steps=160, warmup=8, hamiltonian=None, double_frequency=False, shard=False)),
        # the same configuration with a landscape: the D = 2 universal gate (15 angles), double-frequency rule of Optimizer('Rotosolve')
        ('config1_family_rotosolve_D2_shallowfull_double', dict(workload='rotosolve', D=2, batch=4096, steps=24, warmup=2, hamiltonian=None, double_frequency=True, shard=False, ansatz='shallow-full', no_cpu_baseline=True)),
        ('config3_rotosolve_D8_xxz_256x3', dict(workload='rotosolve', D=8, batch=768, steps=160, warmup=8, hamiltonian=None, double_frequency=False, shard=False)),
        ('config4_evolve_D16_depth4_T256', dict(workload='evolve', D=16, batch=256, steps=10, warmup=3, tol=1e-12, carry_hessian=None)),
        # the same time evolution with more trajectories than the configuration names (the lock-step groups of qmps_evolve_bfgs)
        ('config4_evolve_D16_depth4_T2048', dict(workload='evolve', D=16, batch=2048, steps=10, warmup=3, tol=1e-12, carry_hessian=None, no_cpu_baseline=True, no_extras=True)),
        # config 4's loop at the bond dimension and gate of the REFERENCE's own time evolution (qmps/new_time_evolve.py:186-187: D = 2, ShallowFullStateTensor,
        # 15 angles): the whole optimiser on the device, a workgroup per trajectory (qmps_evolve_bfgs_device)
        ('config4_family_evolve_D2_shallowfull_T256', dict(workload='evolve', D=2, batch=256, steps=10, warmup=3, tol=1e-12, carry_hessian=None, ansatz='shallow-full', no_cpu_baseline=True, no_extras=True)),
    ]
    res = {}
    for name, over in plan:
        if time.perf_counter() - t0 > budget_s:
            res[name] = {'skipped': f'the {budget_s:.0f} s budget of other_configs was spent'}
            continue
        o = copy.copy(args)
        for k, v in over.items():
            setattr(o, k, v)
        o.collect = []
        o.max_iter = 10000
        t1 = time.perf_counter()
        try:
            {'rotosolve': main_rotosolve, 'evolve': main_evolve}[o.workload](o)
            d = o.collect[0]
        except (Exception, SystemExit) as e:        # an extra must never take the headline line down with it
            res[name] = {'error': f'{type(e).__name__}: {e}'}
            continue
        r = d.get('roofline') or {}
        res[name] = {'metric': d['metric'], 'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'warmup': d['warmup'],
                     'roofline': {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'hbm_frac', 'kernel', 'kernel_ms', 'kernel_share_of_wall') if k in r},
                     'cpu_baseline': d.get('cpu_baseline'), 'workload': d['config'].get('workload'),
                     'config': {k: v for k, v in d['config'].items() if k in ('baseline_config', 'hamiltonian', 'D', 'restarts', 'n_params', 'shifts', 'us_per_parameter_update', 'mean_energy_first_sweep',
                                                                                'mean_energy_last_sweep', 'best_energy', 'exact_ground_state_energy', 'D2_optimum', 'D2_manifold_optimum', 'ansatz', 'depth', 'not_converged_or_not_pd',
                                                                                'trajectories_per_gpu', 'driver', 'lockstep_groups', 'bfgs_iterations_per_step', 'carry_hessian', 'not_converged',
                                                                                'mean_final_objective', 'kernel_share_of_wall', 'device_busy', 'adaptive_gradient', 'solver_rounds_mean_gradient_batches',
                                                                                'solver_rounds_max_gradient_batches')},
                     'wall_s': time.perf_counter() - t1}
        if 'repeats' in d:
            res[name]['repeats'] = d['repeats']
        if 'identity_start' in d:       # config 4 BOTH ways: the reference's own restart of every minimisation beside the carried Hessians
            res[name]['identity_start'] = d['identity_start']
    res['what'] = ('BASELINE.json configs[1], [3], [4] run as `--workload rotosolve|evolve` in this process (their own synthetic inputs, contexts and CPU-baseline samples); '
                   'config 4 carries the inverse Hessians between time steps (`identity_start`: scipy\'s / the reference\'s restart from the identity)')
    res['wall_s'] = time.perf_counter() - t0
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--workload', choices=['energy', 'overlap', 'rotosolve', 'evolve'], default='energy',
                    help="'energy' = the headline (two-site energy evaluations, BASELINE.json configs[2]); 'overlap' = the time-evolution "
                         "overlap objective (configs[4]; use with --D 16 --batch 768); 'rotosolve' = sweeps of the device-resident optimiser loop "
                         '(--steps = sweeps, --batch = evaluations per parameter update)')
    ap.add_argument('--shard', action='store_true',
                    help='rotosolve workload at N > 1: --batch / shifts restarts IN ALL, split into contiguous blocks over the ranks '
                         '(qmps_amd.dist.shard_bounds), the sweep costs summed and the best energy taken by RCCL all-reduces (strong scaling; '
                         'BASELINE.json configs[3]); default: every rank runs its own restarts (replicas, weak scaling)')
    ap.add_argument('--hamiltonian', choices=['tfim', 'xxz'], default=None,
                    help='two-site Hamiltonian (default: the one BASELINE.json names for the bond dimension: xxz at D = 8, tfim otherwise)')
    ap.add_argument('--dt', type=float, default=0.05, help='evolve workload: time step (W = exp(-i dt h))')
    ap.add_argument('--gradient', choices=['auto', 'two-sided', 'fd'], default='auto',
                    help="evolve workload: 'two-sided' (auto at D >= 4) = one right + one left eigen-solve per iterate, the central-difference "
                         "neighbours by eta' = <y, T'(r)>/<y, r>; 'fd' = every neighbour eigen-solved (what scipy's BFGS does with the reference objective)")
    ap.add_argument('--python-driver', action='store_true',
                    help='evolve workload: the lock-step BFGS loop in numpy (tools.batched_bfgs), one ctypes call per batch, instead of the one-call native driver (qmps_evolve_bfgs)')
    ap.add_argument('--ansatz', choices=['shallow-cnot', 'shallow-full'], default='shallow-cnot',
                    help="evolve workload at D = 2: 'shallow-full' = ShallowFullStateTensor(2, .) with 15 angles (qmps/new_time_evolve.py:186-187); default: "
                         'ShallowCNOTStateTensor with 8 angles (scripts/loschmidt.py:203-207)')
    ap.add_argument('--host-driver', action='store_true',
                    help='evolve workload at D = 2: the host loop of qmps_evolve_bfgs (lock-step, a round trip per BFGS iteration) instead of the device-resident optimiser')
    ap.add_argument('--no-speculative', action='store_true',
                    help='evolve workload: always evaluate the backtracking ladder before the gradient (default: objective and gradient at the full '
                         'quasi-Newton step first, the ladder only when some trajectory rejects that step)')
    ap.add_argument('--carry-hessian', dest='carry_hessian', action='store_true', default=None,
                    help='evolve workload: carry the inverse Hessians from time step to time step (the default at D >= 8; at D = 2, 4 the identity start measured faster)')
    ap.add_argument('--no-carry-hessian', dest='carry_hessian', action='store_false',
                    help='evolve workload: start the BFGS of every time step from the identity (what scipy - the reference - does) instead of '
                         'the inverse Hessians the previous step ended with; the default run reports this variant as the extra `identity_start`')
    ap.add_argument('--bfgs-iters', type=int, default=30, help='evolve workload: cap on BFGS iterations per time step')
    ap.add_argument('--depth', type=int, default=None, help='rotosolve workload: layers of the ShallowCNOT ansatz (default log2 D)')
    ap.add_argument('--double-frequency', action='store_true', help='rotosolve workload: six shifts per parameter (qmps/tools.py:422-457)')
    # defaults: the chip needs tens of ms of sustained load before its clocks settle
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=450)
    ap.add_argument('--D', type=int, default=4)
    ap.add_argument('--batch', type=int, default=65536, help='evaluations per GPU per step (weak) or in all (strong)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--rotate', type=int, default=0,
                    help='distinct resident batches cycled step by step (0 = smallest count whose tensors exceed the 256 MiB '
                         'Infinity Cache; 1 = re-evaluate one resident batch)')
    ap.add_argument('--max-iter', type=int, default=10000)
    ap.add_argument('--tol', type=float, default=1e-13)
    ap.add_argument('--seed', type=int, default=20241022)
    ap.add_argument('--solver', choices=['direct', 'squaring', 'plain'], default='direct',
                    help="'direct' = exact fixed-point solve accepted by one power step, fused with the energy (library default at "
                         "D = 4; other bond dimensions run 'squaring'); 'squaring' = power iteration 2^m steps at a time; "
                         "'plain' = plain power iteration")
    ap.add_argument('--store-env', action='store_true',
                    help="'direct' only: also write the environments r[B][D][D] to HBM in every step (default: energies, "
                         'iteration counts and status only - SURVEY 8(d)\'s 32 D^2 + 8 bytes per evaluation)')
    ap.add_argument('--handoff', type=int, default=None, help='plain power steps before the squaring tail (default: library default)')
    ap.add_argument('--settle-ms', type=float, default=60.0,
                    help='milliseconds of sustained FP64 probe-kernel load before the warm-up steps, so that the power '
                         'management has raised the clocks whatever --warmup is (0 disables; reported in config)')
    ap.add_argument('--exchange-every', type=int, default=1,
                    help='N > 1: the summed costs of this many steps travel in one RCCL all-reduce (1 = an exchange per step, the '
                         'headline; a grouped figure is printed as an extra)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-repeats', action='store_true', help='energy workload: time the --steps block once (default: 16 blocks, median / min / max reported beside `value`)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='energy workload, N = 1: skip the `other_configs` extra (BASELINE.json configs[1], [3], [4] as their own optimiser workloads, bounded to ~60 s)')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the informational legs that run after the timed region (PCIe-inclusive, ansatz-parameter-inclusive, '
                         'contraction-only, grouped exchange): under rocprofv3 the per-kernel averages then cover the timed workload only')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)

    if args.workload == 'overlap':
        if args.max_iter == 10000:
            args.max_iter = 60 if args.D in (2, 4) else 100000      # D = 2, 4: squarings; D = 8, 16: power steps
        return main_overlap(args)
    if args.workload == 'evolve':
        if args.steps == 2000 and args.warmup == 450:
            args.steps, args.warmup = 8, 2
        if args.batch == 65536:
            args.batch = 256
        if args.D == 4 and '--D' not in sys.argv:
            args.D = 16
        if args.tol == 1e-13:
            args.tol = 1e-12
        return main_evolve(args)
    if args.workload == 'rotosolve':
        if args.steps == 2000 and args.warmup == 450:
            args.steps, args.warmup = 160, 8
        return main_rotosolve(args)
    world, rank, local_rank = world_of(args)

    D = args.D
    _, B, global_batch = shard_plan(args.scaling, args.batch, rank, world)
    if B < 1:
        sys.exit(f'bench.py: rank {rank} owns no evaluations (global batch {args.batch} over {world} ranks)')
    tensor_bytes = 32 * D * D
    R = args.rotate if args.rotate > 0 else max(1, -(-(MALL_MIB * 2 ** 20 + 1) // (B * tensor_bytes)))
    R = min(R, 64)

    # synthetic inputs: every rank draws its own R resident batches (seed + 1000 k + rank)
    A_all = np.concatenate([haar_tensors(args.seed + 1000 * k + rank, D, B) for k in range(R)])
    A = A_all[:B]
    h, h_name = hamiltonian_of(args)

    # CPU baselines first: nothing has touched the GPU yet, so the process pool may fork
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(D, A, h, args.max_iter, args.tol)

    dist = None
    force_dist = os.environ.get('QMPS_BENCH_FORCE_DIST') == '1'   # exercise the N > 1 code path at world_size 1
    rccl_env = {}
    if world > 1 or force_dist:
        # The exchange is 128 bytes per step and must not take compute-unit slots from the energy kernel (two of its waves
        # fill a SIMD's registers to 480 of 512: any resident RCCL wave displaces one): one channel,
        # and not the many-channel MSCCL small-message algorithms.  Defaults only - an exported value wins.
        for k, v in (('NCCL_MAX_NCHANNELS', '1'), ('RCCL_MSCCL_ENABLE', '0'), ('RCCL_MSCCLPP_ENABLE', '0')):
            os.environ.setdefault(k, v)
            rccl_env[k] = os.environ[k]
    if world > 1 or force_dist:
        import torch.distributed as dist  # launcher plumbing only (gloo, CPU)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.path.isdir('/sys/class/net/lo'):
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')     # the container hostname may not resolve
        dist.init_process_group('gloo', rank=rank, world_size=world)

    from qmps_amd import EnergyEngine, _lib
    eng = EnergyEngine(D, R * B, device=local_rank)
    info = _lib.device_info(local_rank)
    eng.set_tensors(A_all)
    eng.set_hamiltonian(h)
    if args.handoff is not None:
        eng.set_solver(args.solver, handoff=args.handoff)

    collective = 'none (N=1)'
    rccl_ok = False
    if dist is not None:
        rccl_ok, err = init_rccl(eng, dist, rank, world)
        if rccl_ok:
            ex = max(1, min(16, args.exchange_every))
            eng.set_exchange_period(ex)
            collective = (f'RCCL communicator of {eng.comm_count()} ranks (ncclCommCount; {", ".join(k + "=" + v for k, v in rccl_env.items())}); ' +
                          ('one all-reduce(sum, f64[16]) per step' if ex == 1 else
                           f'one all-reduce(sum, f64[{ex} x 16]) per {ex} steps: every step\'s summed cost is reduced once, {ex} of them per message'))
        else:
            # reported, never silent: the data path is unchanged (no collective in it); only the summed cost
            # travels over the launcher's gloo group, once, after the timed region
            collective = err + '; summed cost reduced over gloo after the timed region'
            print(f'bench.py[rank {rank}]: {collective}', file=sys.stderr, flush=True)

    # HIP events around the dominant kernel on some launches of the timed region, not on every one: a pair of events
    # costs several us of command-processor fencing per step
    timing_period = max(1, min(args.steps // 4, 16))
    eng.set_kernel_timing_period(timing_period)

    direct = args.solver == 'direct' and D == 4
    store_env = args.store_env or not direct
    accumulate = not (D == 4 and args.solver == 'squaring')      # every path but the two-kernel D = 4 squaring solver
    count = [0]

    def step():
        eng.set_window((count[0] % R) * B)
        count[0] += 1
        eng.launch(B, max_iter=args.max_iter, tol=args.tol, solver=args.solver, store_env=store_env, accumulate_cost=accumulate)
        eng.cost_launch(B)

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()
        eng.sync()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        eng.timer_begin()
        for _ in range(n):
            step()
        ev = eng.timer_end()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ev

    # clock settle: sustained load from the library's FP64 probe kernel (not steps of the workload), see DESIGN.md section 5
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        eng.probe_fp64_tflops()
    for _ in range(args.warmup):
        step()
    if dist is not None and rccl_ok:
        eng.exchange_stats(reset=True)
    elapsed, ev_ms = timed(args.steps)
    exch = eng.exchange_stats() if dist is not None and rccl_ok else None
    # the same block of --steps steps again, 15 times (every rank: the blocks carry the barriers and, at N > 1, the exchanges of the
    # first): `value` stays the FIRST block's (the contract's K timed steps); median / min / max of all 16 are reported beside it
    block_s = [elapsed] + [timed(args.steps)[0] for _ in range(0 if args.no_repeats else 15)]

    cost = eng.get_cost()
    if dist is not None and not rccl_ok:
        import torch
        t = torch.tensor([float(cost[0])], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        cost = np.array([t.item()])
    kernel_ms, kernel_name = eng.kernel_time(args.steps)
    # iteration counts / status of every resident batch (each window holds the results of its last step)
    its, sts = [], []
    for k in range(min(R, count[0])):
        eng.set_window(k * B)
        _, it_k, st_k = eng.results(B)
        its.append(it_k)
        sts.append(st_k)
    iters, status = np.concatenate(its), np.concatenate(sts)
    eng.set_window(0)

    extras = {}
    if rank == 0 and not args.no_extras:
        # PCIe-inclusive rate (never `value`): host tensors in, energies out through the one-shot entry point
        eng.set_solver(args.solver, handoff=args.handoff if args.handoff is not None else eng.handoff)
        eng.energies(A, h, max_iter=args.max_iter, tol=args.tol)
        t1 = time.perf_counter()
        for _ in range(3):
            eng.energies(A, h, max_iter=args.max_iter, tol=args.tol)
        extras['pcie_inclusive_evals_per_s'] = 3 * B / (time.perf_counter() - t1)
        # the reference's own call pattern (params -> energy, SparseFullEnergyOptimizer with the default ShallowCNOT ansatz,
        # ground_state.py:150-168): only 8 P bytes per evaluation cross PCIe, the circuit is simulated on the device
        if D in (2, 4, 8, 16):
            depth = {2: 1, 4: 2, 8: 3, 16: 4}[D]
            prm = np.random.default_rng(args.seed + 7).standard_normal((B, 2 * depth))

            def ansatz_eval():
                eng.set_ansatz_params(_lib.ANSATZ_SHALLOW_CNOT, prm)
                eng.launch(B, max_iter=args.max_iter, tol=args.tol, solver=args.solver, store_env=store_env)
                return eng.results(B)
            ansatz_eval()
            t1 = time.perf_counter()
            for _ in range(3):
                _, it_a, st_a = ansatz_eval()
            extras['ansatz_params_inclusive'] = {
                'evals_per_s': 3 * B / (time.perf_counter() - t1), 'mean_power_iterations': float(it_a.mean()),
                'fallback_fraction': float((it_a > 1).mean()) if direct else None, 'not_converged_or_not_pd': int((st_a != 0).sum()),
                'what': 'ShallowCNOT parameters in host memory -> energies in host memory (device-side circuit, environment, energy)'}
        # the contraction chain alone (north star: A - Abar - h - A - Abar with the resident environment): energy-only launches
        # over the rotating windows; bytes: SURVEY 8(d)'s 32 D^2 + 8 (headline accounting) and the 48 D^2 + 8 the launch
        # really reads (tensor + environment)
        eng.set_tensors(A_all)
        for k in range(R):
            eng.set_window(k * B)
            eng.launch(B, max_iter=args.max_iter, tol=args.tol, solver=args.solver, store_env=True)
        for k in range(2 * R):
            eng.set_window((k % R) * B)
            eng.launch_energy_only(B)
        eng.sync()
        n_co = max(30, 3 * R)
        eng.timer_begin()
        for k in range(n_co):
            eng.set_window((k % R) * B)
            eng.launch_energy_only(B)
        us = eng.timer_end() / n_co * 1e3
        eng.set_window(0)
        extras['contraction_only'] = {
            'us_per_launch': us, 'evals_per_s': B / (us * 1e-6),
            'hbm_gbps_520B': B * bytes_per_eval(D) / (us * 1e-6) * 1e-9, 'hbm_frac_520B': B * bytes_per_eval(D) / (us * 1e-6) * 1e-9 / HBM_PEAK_GBPS,
            'hbm_gbps_tensor_plus_env': B * (48 * D * D + 8) / (us * 1e-6) * 1e-9,
            'hbm_frac_tensor_plus_env': B * (48 * D * D + 8) / (us * 1e-6) * 1e-9 / HBM_PEAK_GBPS,
            'working_set_mib': R * B * 48 * D * D / 2 ** 20,
            'what': f'qmps_energy_only_launch (no environment solve) cycled over the {R} resident batches; reads tensor + environment '
                    f'({48 * D * D + 8} B per evaluation); the {bytes_per_eval(D)} B figure is SURVEY 8(d)\'s accounting'}
        if direct:
            # warm start ("reachable only for small K (warm-started environments)"): every evaluation finds its
            # converged environment resident, passes the acceptance test and skips the matrix build and the elimination
            eng.set_tensors(A_all)
            for k in range(R):
                eng.set_window(k * B)
                eng.launch(B, max_iter=args.max_iter, tol=args.tol, solver='direct', store_env=True)
            # (this leg runs on rank 0 ALONE: with a communicator `cost_launch` is a collective call - a rank that issued it by
            # itself would queue unmatched all-reduces - so the summed cost is only taken without one)
            with_cost = not (dist is not None and rccl_ok)

            def warm_step(k):
                eng.set_window((k % R) * B)
                eng.launch(B, max_iter=args.max_iter, tol=args.tol, solver='direct', store_env=False, accumulate_cost=with_cost, warm_start=True)
                if with_cost:
                    eng.cost_launch(B)
            for k in range(3 * R):
                warm_step(k)
            eng.sync()
            n_w = max(90, 10 * R)
            eng.timer_begin()
            for k in range(n_w):
                warm_step(k)
            us_w = eng.timer_end() / n_w * 1e3
            _, it_w, st_w = eng.results(B)
            eng.set_window(0)
            extras['warm_start'] = {
                'us_per_step': us_w, 'evals_per_s': B / (us_w * 1e-6), 'accepted_fraction': float((it_w == 1).mean()),
                'not_converged_or_not_pd': int((st_w != 0).sum()),
                'hbm_gbps_520B': B * bytes_per_eval(D) / (us_w * 1e-6) * 1e-9, 'hbm_frac_520B': B * bytes_per_eval(D) / (us_w * 1e-6) * 1e-9 / HBM_PEAK_GBPS,
                'hbm_gbps_tensor_plus_env': B * (48 * D * D + 8) / (us_w * 1e-6) * 1e-9,
                'hbm_frac_tensor_plus_env': B * (48 * D * D + 8) / (us_w * 1e-6) * 1e-9 / HBM_PEAK_GBPS,
                'flops_per_eval': 1920 + 5040, 'working_set_mib': R * B * 48 * D * D / 2 ** 20,
                'what': 'QMPS_FLAG_WARM_RESIDENT: the resident (converged) environment of every evaluation is accepted by one power step; '
                        f'no matrix build, no elimination; reads tensor + environment ({48 * D * D + 8} B per evaluation), cycled over the {R} resident batches'}
    if rank == 0 and not args.no_extras and D <= 8:
        # BASELINE.json configs[2] says "power-iteration environment solve": the same step with the iterative solvers on the
        # record (QMPS_ENV_POWER = the reference's krylov / PowerCircuit, QMPS_ENV_POWER_SQUARING = the same 2^m steps at a time);
        # same fixed point, same tolerance as the direct solve of the headline.  No cost exchange in this rank-0-only leg.
        eng.set_tensors(A_all)
        legs = {}
        for name in ('plain', 'squaring'):
            if name == args.solver:
                continue
            n_leg = max(3, min(20, args.steps)) if name == 'plain' else max(5, min(60, args.steps))
            def leg_step(k, name=name):
                eng.set_window((k % R) * B)
                eng.launch(B, max_iter=args.max_iter, tol=args.tol, solver=name, store_env=True)
            for k in range(2):
                leg_step(k)
            eng.sync()
            eng.timer_begin()
            for k in range(n_leg):
                leg_step(k)
            ms = eng.timer_end() / n_leg
            _, it_l, st_l = eng.results(B)
            legs[name] = {'evals_per_s': B / (ms * 1e-3), 'ms_per_step': ms, 'steps_timed': n_leg, 'mean_power_iterations': float(it_l.mean()),
                          'max_power_iterations': int(it_l.max()), 'not_converged_or_not_pd': int((st_l != 0).sum())}
            if name == 'plain':
                # the power-iteration solve priced two ways: SURVEY 8(d)'s operator-form count (what a reader of the north star expects) and the
                # flops the D = 4 kernel executes (the map as a real 16 x 16 matrix: a third of them); a launch ends with its slowest evaluation
                # (max_power_iterations steps of ~0.2 us for a wave alone on its SIMD: the floor of the leg whatever the batch)
                fl_survey = float(flops_per_eval(D, it_l.astype(np.float64)).sum())
                fl_exec, fl_note, _ = executed_flops(D, 'plain', it_l, eng, args.max_iter)
                legs[name].update({'fp64_tflops_survey_formula': fl_survey / (ms * 1e-3) * 1e-12, 'fp64_frac_survey_formula': fl_survey / (ms * 1e-3) * 1e-12 / FP64_PEAK_TFLOPS,
                                   'fp64_tflops_executed': fl_exec / (ms * 1e-3) * 1e-12, 'fp64_frac_executed': fl_exec / (ms * 1e-3) * 1e-12 / FP64_PEAK_TFLOPS,
                                   'executed_flops_note': fl_note})
        eng.set_window(0)
        legs['what'] = ('the same resident batches, environment by power iteration to the same tolerance: "plain" = normalised power '
                        'iteration (krylov / PowerCircuit of the reference), "squaring" = 2^m power steps at a time; energies stored, no cost sum')
        extras['power_iteration'] = legs
    if dist is not None and rccl_ok and not args.no_extras and args.exchange_every == 1:
        # the grouped exchange (16 steps' costs per all-reduce) as an extra, every rank takes part
        eng.set_tensors(A_all)
        eng.set_exchange_period(16)
        for _ in range(32):
            step()
        n16 = max(32, args.steps // 4)
        el16, _ = timed(n16)
        eng.set_exchange_period(1)
        extras['grouped_exchange_16'] = {'evals_per_s': global_batch * n16 / el16,
                                         'what': 'same steps, the summed costs of 16 consecutive steps per all-reduce'}

    if rank == 0 and world == 1 and dist is None and not args.no_extras and not args.no_other_configs and (D, args.batch) == (4, 65536):
        extras['other_configs'] = other_configs(args)
    tot = np.array([float(iters.sum()), float((status != 0).sum()), float((iters > 1).sum()), float(len(iters))])
    if dist is not None:
        import torch
        t = torch.tensor(tot, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tot = t.numpy()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = global_batch * args.steps / elapsed
        step_ms_events = ev_ms / args.steps
        # roofline of the dominant kernel: executed-algorithm FLOPs of ONE launch (mean over the resident batches)
        flops_all, flop_note, handoff = executed_flops(D, args.solver, iters, eng, args.max_iter)
        flops = flops_all / max(1, len(iters) // B)
        kernel_ms_pair = kernel_ms
        single_kernel_step = direct and world == 1 and dist is None
        if single_kernel_step:
            # the step IS the dominant kernel (the cost is accumulated inside it): its average duration is the event-bracketed
            # timed region / steps (a pair of events around single launches adds ~2.5 us of command-processor fencing, reported
            # beside it as kernel_ms_event_pairs)
            kernel_ms = step_ms_events
        tflops = flops / (kernel_ms * 1e-3) * 1e-12
        traffic = committed_traffic(D, B, args.solver, store_env, R)
        hbm_gbps = B * bytes_per_eval(D) / (kernel_ms * 1e-3) * 1e-9
        step_gbps = B * bytes_per_eval(D) / (step_ms_events * 1e-3) * 1e-9
        out = {
            'metric': 'two-site energy evals/sec at D=4, batch=65536' if (D, args.batch) == (4, 65536)
                      else f'two-site energy evals/sec at D={D}, batch={args.batch}',
            'value': value, 'unit': 'two-site energy evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{h_name} two-site energy, D={D}, '
                                   + (f'batch={B} per GPU' if args.scaling == 'weak' else f'global batch={global_batch} split B/G per GPU')
                                   + f', Haar-random state unitaries, in-kernel environment solve (tol {args.tol:g}, cap {args.max_iter}, '
                                     f'solver {args.solver}{"" if store_env else ", environments not stored"}), {R} resident batches cycled',
                       'baseline_config': {2: 'BASELINE.json configs[1]', 4: 'BASELINE.json configs[2]', 8: 'BASELINE.json configs[3]', 16: 'BASELINE.json configs[4] (energy kernel)'}[D],
                       'hamiltonian': h_name, 'D': D, 'batch_per_gpu': B, 'global_batch': global_batch, 'tol': args.tol, 'max_iter': args.max_iter, 'seed': args.seed,
                       'resident_batches': R, 'working_set_mib': R * B * tensor_bytes / 2 ** 20,
                       'clock_settle_ms': args.settle_ms,
                       'mean_power_iterations': tot[0] / tot[3],
                       'fallback_fraction': (tot[2] / tot[3]) if direct else None,
                       'max_power_iterations_rank0': int(iters.max()), 'not_converged_or_not_pd': int(tot[1]),
                       'collective': collective, 'rccl_ranks_seen': getattr(init_rccl, 'ranks_seen', None) if dist is not None else None,
                       'exchange_pipeline_rank0': None if exch is None else {
                           'slot_guard_checks': exch[0], 'host_waited_for_an_exchange': exch[1], 'host_wait_ms': exch[2],
                           'what': 'timed region, rank 0: the host issues a step in ~10 us and is throttled at the ring (it may run 6 steps ahead): '
                                   'how often and for how long it waited for the exchange that last used the slot.  Whether the all-reduce or the '
                                   'energy kernel sets the pace shows in ms_per_step against roofline.kernel_ms'},
                       'device': info['name'], 'arch': info['arch']},
            'roofline': {'bound': 'fp64_matrix' if D == 16 or (D == 4 and args.solver == 'squaring') else 'fp64_valu',
                         'achieved': tflops, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': tflops / FP64_PEAK_TFLOPS, 'hbm_frac': hbm_gbps / HBM_PEAK_GBPS, 'traffic': traffic,
                         'kernel': kernel_name, 'kernel_ms': kernel_ms, 'step_ms_events': step_ms_events,
                         'kernel_ms_event_pairs': kernel_ms_pair, 'kernel_timed_every': timing_period,
                         'kernel_ms_from': ('HIP events bracketing the timed region on the context stream / steps: the step is this ONE kernel'
                                            if single_kernel_step else 'HIP event pairs around the kernel on every kernel_timed_every-th launch'),
                         'note': '`bound` names the roofline that binds: fp64_valu = the FP64 vector pipe (the fused D = 4 kernel and the D = 2, 8 '
                                 'kernels issue no MFMA), fp64_matrix = v_mfma_f64 (D = 16, the D = 4 squaring solver); both peaks are 78.6 TFLOP/s '
                                 'spec (measured on this part: v_fma_f64 70.9, v_mfma_f64_16x16x4 47.7 TFLOP/s, profiles/archive/r01_probe.json); `hbm_frac` = '
                                 'the same launch against the 8 TB/s HBM roofline on the algorithmic bytes.  `traffic` = HBM bytes per launch from the '
                                 'PMC passes COMMITTED under profiles/ (rocprofv3 --pmc cannot run inside this process: `traffic.source` names the run).  '
                                 'FLOPs = ' + flop_note,
                         'hbm': {'achieved': hbm_gbps, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                 'frac': hbm_gbps / HBM_PEAK_GBPS, 'bytes_per_eval': bytes_per_eval(D),
                                 'whole_step_gbps': step_gbps, 'whole_step_frac': step_gbps / HBM_PEAK_GBPS,
                                 'working_set_mib': R * B * tensor_bytes / 2 ** 20,
                                 'note': f'algorithmic bytes (SURVEY 8(d): 32 D^2 + 8 = {bytes_per_eval(D)} B per evaluation) over the '
                                         f'dominant kernel / the whole step; {R} resident batches cycled, tensors '
                                         f'{"exceed" if R * B * tensor_bytes > MALL_MIB * 2 ** 20 else "fit inside"} the {MALL_MIB} MiB Infinity Cache'}},
            'summed_cost': float(cost[0]),
            'repeats': {'blocks': len(block_s), 'steps_per_block': args.steps,
                        'value_median': global_batch * args.steps / float(np.median(block_s)), 'value_min': global_batch * args.steps / max(block_s),
                        'value_max': global_batch * args.steps / min(block_s), 'ms_per_step_median': float(np.median(block_s)) / args.steps * 1e3,
                        'ms_per_step_min': min(block_s) / args.steps * 1e3, 'ms_per_step_max': max(block_s) / args.steps * 1e3,
                        'what': 'the timed block of --steps steps repeated back to back (block 0 is `value`), each bracketed by barrier + synchronise, max over ranks'},
        }
        out.update(extras)
        # the median of the repeated blocks beside `value` (the driver's --steps 20 makes `value` a 0.6 ms sample), and at N > 1 the verdict on
        # what paces a step
        out['value_median'] = out['repeats']['value_median']
        out.update(exchange_report(world, ms_per_step, kernel_ms, value, None if exch is None else float(exch[2]), args.steps,
                                   (extras.get('grouped_exchange_16') or {}).get('evals_per_s')))
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu
        emit(args, out)

    if dist is not None and rccl_ok:
        eng.comm_destroy()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
