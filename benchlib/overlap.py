"""benchlib.overlap - `bench.py --workload overlap`: the time-evolution overlap objective alone (f-3)."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403  (inputs, byte / flop counts, CPU baselines, launcher plumbing, emit)
from .common import FP64_PEAK_TFLOPS, HBM_PEAK_GBPS, MALL_MIB, ROOT, _one_blas_thread  # noqa: F401


def nearby_tensors(seed, D, B, eps_max):
    """Candidates of a time-evolution step: U exp(i eps H) for one Haar reference unitary U, H random Hermitian,
    eps ~ U(0, eps_max).  Returns (A_ref (2,D,D), candidates (B,2,D,D))."""
    rng = np.random.default_rng(seed)
    n = 2 * D
    Z = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    U, _ = np.linalg.qr(Z)
    G = rng.standard_normal((B, n, n)) + 1j * rng.standard_normal((B, n, n))
    w, V = np.linalg.eigh((G + G.conj().transpose(0, 2, 1)) / 2)
    eps = rng.uniform(0.0, eps_max, B)
    Us = U[None] @ (V * np.exp(1j * eps[:, None] * w)[:, None, :]) @ V.conj().transpose(0, 2, 1)
    to_tensor = lambda Q: Q[..., :D].reshape(Q.shape[:-2] + (D, 2, D)).swapaxes(-3, -2)
    return np.ascontiguousarray(to_tensor(U)), np.ascontiguousarray(to_tensor(Us))


def main_overlap(args):
    """--workload overlap: BASELINE.json configs[4] (TFIM quench time evolution, D = 16 on the matrix cores): one step =
    the overlap objective eta_b (dominant eigenvalue of the mixed two-site transfer map, qmps/new_time_evolve.py:193-221)
    of B resident candidates against the current state.  Independent trajectories: replicas only, no collective."""
    world, rank, local_rank = world_of(args)
    D, B = args.D, args.batch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from scipy.linalg import expm
    A, cands = nearby_tensors(args.seed + rank, D, B, 0.1)
    WW = expm(-1j * 0.05 * tfim_h(1.0))
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import qmps_oracle as O
        n = min(B, 24 if D >= 16 else 200)
        t = time.perf_counter()
        for k in range(n):
            O.overlap_eta(A, cands[k], WW)
        cpu = {'value': n / (time.perf_counter() - t), 'unit': 'overlap evals/s', 'cores': 1, 'kind': 'port',
               'sample': f'first {n} candidates; numpy dense eig of the {D * D} x {D * D} mixed transfer matrix (what xmps '
                         'Map.right_fixed_point computes for the reference)'}
    from qmps_amd import EnergyEngine, _lib
    eng = EnergyEngine(D, B, device=local_rank)
    info = _lib.device_info(local_rank)
    eng.set_tensors(cands)
    eng.overlap_set(A, WW)
    eng.set_kernel_timing_period(max(1, min(args.steps // 4, 16)))
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        eng.probe_fp64_tflops()
    for _ in range(args.warmup):
        eng.overlap_launch(B, max_rounds=args.max_iter, tol=args.tol)
    eng.sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    eng.timer_begin()
    for _ in range(args.steps):
        eng.overlap_launch(B, max_rounds=args.max_iter, tol=args.tol)
    ev_ms = eng.timer_end()
    eng.sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eta, rounds, st = eng.overlap_results(B)
    kernel_ms, kernel_name = eng.kernel_time(args.steps)
    if rank == 0:
        # executed algorithm: per power step 8 complex D^3 products (Y_s = x Bm_s^+, x' += C_s Y_s, s < 4) = 64 D^3 flop,
        # set-up 8 products (merge(A,A), merge(B,B)) + the WW combination
        if D in (2, 4):   # squarings of the complex D^2 x D^2 matrix: 8 (D^2)^3 flop each
            flops = float((rounds.astype(np.float64) * 8 * (D * D) ** 3 + 64 * D ** 3 + 32 * D ** 4).sum())
        else:
            flops = float((rounds.astype(np.float64) * 64 * D ** 3 + 64 * D ** 3 + 128 * D * D).sum())
        tflops = flops / (kernel_ms * 1e-3) * 1e-12
        byts = B * (32 * D * D + 16)
        out = {'metric': f'time-evolution overlap evals/sec at D={D}, batch={B}', 'value': world * B * args.steps / elapsed,
               'unit': 'overlap evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': f'TFIM quench time-evolution overlap objective, D={D}, batch={B} candidates per GPU near one Haar '
                                      f'reference state (U exp(i eps H), eps < 0.1), W = exp(-0.05 i h_TFIM), tol {args.tol:g}, cap {args.max_iter} '
                                      + ('squarings' if D in (2, 4) else 'power steps'),
                          'baseline_config': 'BASELINE.json configs[4]', 'D': D, 'batch_per_gpu': B, 'seed': args.seed,
                          'mean_power_steps': float(rounds.mean()), 'max_power_steps': int(rounds.max()), 'not_converged': int((st != 0).sum()),
                          'mean_abs_eta': float(np.abs(eta).mean()), 'collective': 'none: independent trajectories (replicas only)',
                          'device': info['name'], 'arch': info['arch']},
               'roofline': {'bound': 'fp64_matrix' if D in (4, 16) else 'fp64_valu', 'achieved': tflops, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tflops / FP64_PEAK_TFLOPS,
                            'hbm_frac': byts / (kernel_ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS,
                            'traffic': committed_traffic(D, B, 'overlap', 0, 1), 'kernel': kernel_name, 'kernel_ms': kernel_ms, 'step_ms_events': ev_ms / args.steps,
                            'note': 'executed FLOPs = sum_b [steps_b 64 D^3 + 64 D^3] (complex D^3 products = 8 D^3 flop), steps read back per '
                                    'item; D = 16: v_mfma_f64_16x16x4 (measured 47.7 TFLOP/s issue rate on this part, profiles/archive/r01_probe.json)',
                            'hbm': {'achieved': byts / (kernel_ms * 1e-3) * 1e-9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                    'frac': byts / (kernel_ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS, 'bytes_per_eval': 32 * D * D + 16}},
               'cpu_baseline': cpu}
        emit(args, out)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
