"""benchlib.rotosolve - `bench.py --workload rotosolve [--shard]`: the device-resident rotosolve loop (configs 1, 3; a-10 / f-2)."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403  (inputs, byte / flop counts, CPU baselines, launcher plumbing, emit)
from .common import FP64_PEAK_TFLOPS, HBM_PEAK_GBPS, MALL_MIB, ROOT, _one_blas_thread  # noqa: F401


def rotosolve_shard_plan(R_global, rank, world, shard):
    """(first restart, restarts on this rank, restarts in all).  --shard: the R_global restarts are split into contiguous
    blocks (qmps_amd.dist.shard_bounds; BASELINE.json configs[3]: "256 random restarts x 3 angle samples sharded over 8 MI355X");
    otherwise every rank runs its own R_global restarts (replicas)."""
    from qmps_amd.dist import shard_bounds
    if shard:
        lo, hi = shard_bounds(R_global, rank, world)
        return lo, hi - lo, R_global
    return rank * R_global, R_global, world * R_global


def main_rotosolve(args):
    """--workload rotosolve: the caller that produces the batch (qmps/rotosolve.py:154-181,
    qmps/tools.py:422-457).  R restarts of the optimisers' default ansatz (ShallowCNOTStateTensor, depth log2(D)) in
    lock-step; one step = one SWEEP of the device-resident rotosolve (every parameter once: shifted batches of 3 R
    evaluations - ansatz, environment, energy - and the closed-form updates), --batch = 3 R evaluations per parameter
    update.  `value` counts the energy evaluations the optimiser consumed per second.  Replicas only at N > 1."""
    world, rank, local_rank = world_of(args)
    D = args.D
    nsh = 6 if args.double_frequency else 3
    R_global = max(1, args.batch // nsh)
    first, R, R_all = rotosolve_shard_plan(R_global, rank, world, args.shard)
    if R < 1:
        sys.exit(f'bench.py: rank {rank} owns no restarts ({R_global} over {world} ranks)')
    depth = getattr(args, 'depth', None) or {2: 1, 4: 2, 8: 3, 16: 4}[D]
    P = 2 * depth
    # --ansatz shallow-full (D = 2 only): ShallowFullStateTensor(2, v), 15 angles - a universal two-qubit gate, so the D = 2 optimum
    # -1.269909412573 (/root/reference/scripts/noisy_optimization.py:93) is reachable.  BASELINE.json configs[1] as written
    # (ShallowCNOT, depth 1) is a FLAT landscape for TFIM: E(beta, gamma) = 0 identically, in the reference itself
    # (tests/test_refshim_cpu.py::test_config1_landscape_is_flat_in_the_reference_itself) - it times the machinery, not an optimisation.
    full = getattr(args, 'ansatz', 'shallow-cnot') == 'shallow-full'
    if full:
        if D != 2:
            sys.exit('bench.py: --ansatz shallow-full is the D = 2 gate of the reference (represent.py:383-404)')
        P = 15
    dist = None
    force_dist = os.environ.get('QMPS_BENCH_FORCE_DIST') == '1'
    if world > 1 or (force_dist and args.shard):
        for k, v in (('NCCL_MAX_NCHANNELS', '1'), ('RCCL_MSCCL_ENABLE', '0'), ('RCCL_MSCCLPP_ENABLE', '0')):
            os.environ.setdefault(k, v)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.path.isdir('/sys/class/net/lo'):
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    h, h_name = hamiltonian_of(args)
    # --shard: ONE global set of restarts (same seed on every rank), this rank's contiguous block of it
    p0 = (np.random.default_rng(args.seed).standard_normal((R_all, P))[first:first + R] if args.shard
          else np.random.default_rng(args.seed + rank).standard_normal((R, P)))
    shifts = np.array([0.0, np.pi, np.pi / 2, -np.pi / 2, np.pi / 4, -np.pi / 4]) if nsh == 6 else np.array([0.0, np.pi / 2, -np.pi / 2])
    shifted = np.repeat(p0, nsh, axis=0)
    shifted[:, 0] += np.tile(shifts, R)          # the batch of the first parameter update: evaluation nsh r + k = restart r, shift k
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        # the oracle on one host core over a bounded sample of the same shifted batch: circuit model -> tensor (numpy),
        # plain power iteration + closed-form energy (C)
        from oracle import c_oracle as C
        from oracle import qmps_oracle as O
        C.build()
        n = min(len(shifted), 3000 if D <= 4 else (600 if D == 8 else 150))
        t = time.perf_counter()
        A_cpu = np.stack([O.unitary_to_tensor(O.shallow_full_unitary(q) if full else O.shallow_cnot_unitary(D, q)) for q in shifted[:n]])
        C.energy_batch(A_cpu, h, max_iter=args.max_iter, tol=args.tol, threads=1)
        cpu = {'value': n / (time.perf_counter() - t), 'unit': 'two-site energy evals/s', 'cores': 1, 'kind': 'port',
               'sample': f'first {n} evaluations of the first parameter update\'s shifted batch: parameters -> unitary by the oracle\'s gate-by-gate '
                         'circuit model (numpy) -> tensor -> plain power iteration + closed-form energy (oracle/qmps_oracle.c), 1 thread'}
    from qmps_amd import EnergyEngine, _lib
    kind = _lib.ANSATZ_SHALLOW_FULL if full else _lib.ANSATZ_SHALLOW_CNOT
    eng = EnergyEngine(D, nsh * R, device=local_rank)
    info = _lib.device_info(local_rank)
    eng.set_hamiltonian(h)
    collective, reducer = 'none: independent restarts (replicas only)', None
    if args.shard and dist is not None:
        from qmps_amd.dist import RcclReducer
        ok, err = init_rccl(eng, dist, rank, world)
        if ok:
            reducer = RcclReducer(eng)
            collective = (f'RCCL communicator of {eng.comm_count()} ranks (ncclCommCount): after the sweeps of a run, the summed cost of every '
                          'sweep over all ranks\' restarts (ncclAllReduce sum, <= 16 doubles per message) and the best final energy (ncclAllReduce min)')
        else:
            # reported, never silent; the restarts themselves need no collective, the reduction then travels over the launcher's gloo group
            import torch

            class _Gloo:
                def allreduce_sum(self, v):
                    t = torch.tensor(np.asarray(v, dtype=np.float64)); dist.all_reduce(t, op=dist.ReduceOp.SUM); return t.numpy().copy()

                def allreduce_min(self, v):
                    t = torch.tensor(np.asarray(v, dtype=np.float64)); dist.all_reduce(t, op=dist.ReduceOp.MIN); return t.numpy().copy()
            reducer = _Gloo()
            collective = err + '; sweep costs reduced over gloo'
            print(f'bench.py[rank {rank}]: {collective}', file=sys.stderr, flush=True)
    run = eng.double_rotosolve if args.double_frequency else eng.rotosolve
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        eng.probe_fp64_tflops()
    sweeps_w = max(1, min(args.warmup, 64))
    sweeps = max(1, min(args.steps, 256))
    run(kind, p0, sweeps_w, max_iter=args.max_iter, tol=args.tol)
    eng.sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    hist, pfin = run(kind, p0, sweeps, max_iter=args.max_iter, tol=args.tol)
    reduced = None
    if reducer is not None:
        # the path's exchange step for sharded restarts, inside the timed region
        from qmps_amd.dist import reduce_sweep_costs
        reduced = reduce_sweep_costs(hist, reducer)
    eng.sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # roofline of the dominant kernel of a parameter update: the environment + energy kernel over one shifted batch, timed by
    # HIP events on stand-alone launches of that very batch (inside the run the sweep is a replayed hipGraph: no events there)
    roof = None
    if rank == 0:
        eng.set_ansatz_params(kind, shifted)
        eng.set_kernel_timing_period(1)
        for _ in range(12):
            eng.launch(nsh * R, max_iter=args.max_iter, tol=args.tol, solver='direct', store_env=(D != 4))
        kms, kname = eng.kernel_time(8)
        _, it_r, st_r = eng.results(nsh * R)
        fl, fl_note, _ = executed_flops(D, 'direct', it_r, eng, args.max_iter)
        if D == 4:
            fl += 1700.0 * len(it_r)                 # the fused ansatz prologue (DESIGN.md kernel table)
        tf = fl / (kms * 1e-3) * 1e-12
        byts = nsh * R * (8 * P + 16) if D == 4 else nsh * R * bytes_per_eval(D)
        roof = {'bound': 'fp64_matrix' if D == 16 else 'fp64_valu', 'achieved': tf, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / FP64_PEAK_TFLOPS,
                'hbm_frac': byts / (kms * 1e-3) * 1e-9 / HBM_PEAK_GBPS, 'traffic': None, 'kernel': kname, 'kernel_ms': kms,
                'kernel_ms_from': 'HIP event pairs around 8 stand-alone launches of the first parameter update\'s shifted batch (same kernel, same shape as inside the captured sweep)',
                'mean_power_iterations': float(it_r.mean()), 'not_converged_or_not_pd': int((st_r != 0).sum()),
                'note': 'small batches are latency-bound: the fraction says how far below the FP64 roofline a parameter update sits.  FLOPs = ' + fl_note}
    if rank == 0:
        evals_all = sweeps * P * nsh * R_all + R_all  # shifted batches (a sweep's record comes from the next sweep's shift-0 rows) + the final evaluation
        out = {'metric': f'rotosolve energy evals/sec at D={D}, {R_all} restarts x {nsh} shifts', 'value': evals_all / elapsed,
               'unit': 'two-site energy evals/s', 'n_gpus': world, 'steps': sweeps, 'warmup': sweeps_w,
               'ms_per_step': elapsed / sweeps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if args.shard else 'weak', 'vs_baseline': None,
               'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': f'device-resident {"double-frequency " if nsh == 6 else ""}rotosolve, {h_name}, D={D}, {"ShallowFull" if full else f"ShallowCNOT depth {depth}"} '
                                      f'({P} parameters), {R} restarts x {nsh} shifts = {nsh * R} evaluations per parameter update, one step = one sweep; '
                                      'the whole run is ONE C call (fixed costs - allocation, graph capture, copies - included)',
                          'baseline_config': {2: 'BASELINE.json configs[1]', 4: 'BASELINE.json configs[2] (as an optimiser loop)', 8: 'BASELINE.json configs[3]', 16: 'BASELINE.json configs[4] (energy objective)'}[D],
                          'hamiltonian': h_name, 'D': D, 'restarts': R, 'shifts': nsh,
                          'n_params': P, 'us_per_parameter_update': elapsed / (sweeps * P) * 1e6,
                          'best_energy': float(np.nanmin(hist[-1])), 'mean_energy_first_sweep': float(np.nanmean(hist[0])),
                          'mean_energy_last_sweep': float(np.nanmean(hist[-1])), 'exact_ground_state_energy': (-4 / np.pi) if h_name.startswith('TFIM') else None,
                          'D2_optimum': -1.269909412573 if (D == 2 and h_name.startswith('TFIM')) else None, 'D2_manifold_optimum': -1.2725424859 if (D == 2 and h_name.startswith('TFIM')) else None, 'ansatz': 'ShallowFullStateTensor' if full else 'ShallowCNOTStateTensor', 'depth': None if full else depth,
                          'restarts_global': R_all, 'restarts_this_rank': R, 'sharded': bool(args.shard),
                          'summed_cost_last_sweep_all_ranks': None if reduced is None else float(reduced[0][-1]),
                          'restarts_counted_all_ranks': None if reduced is None else reduced[1],
                          'best_energy_all_ranks': None if reduced is None else reduced[2],
                          'collective': collective, 'rccl_ranks_seen': getattr(init_rccl, 'ranks_seen', None) if (args.shard and dist is not None) else None,
                          'device': info['name'], 'arch': info['arch']},
               'roofline': roof, 'cpu_baseline': cpu}
        emit(args, out)
    if reducer is not None and hasattr(reducer, 'engine'):
        eng.comm_destroy()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
