"""benchlib.evolve - `bench.py --workload evolve`: BASELINE config 4 as a workload, trajectory time steps per second (N-2)."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403  (inputs, byte / flop counts, CPU baselines, launcher plumbing, emit)
from .common import FP64_PEAK_TFLOPS, HBM_PEAK_GBPS, MALL_MIB, ROOT, _one_blas_thread  # noqa: F401


def _dominant_kernel(eng, fallback):
    """name the library gives the dominant kernel of its last timed launch (c->dominant), or `fallback` when nothing was timed"""
    try:
        return eng.kernel_time(1)[1] or fallback
    except Exception:
        return fallback


def evolve_cpu_baseline(D, P, WW, seed, maxiter, budget_T=2, full=False):
    """The same lock-step BFGS time step with the ORACLE as evaluator, the way the reference obtains eta (xmps Map ->
    scipy.sparse.linalg.eigs, ARPACK in operator form: oracle.overlap_eta_arpack) and the oracle's own circuit model for
    parameters -> tensor, one host core, on a bounded sample: budget_T trajectories, one time step."""
    from oracle import qmps_oracle as O
    from qmps_amd.tools import batched_bfgs
    _one_blas_thread()
    X = np.random.default_rng(seed).standard_normal((budget_T, P))
    unitary = (lambda D_, x: O.shallow_full_unitary(x)) if full else O.shallow_cnot_unitary
    A = [O.unitary_to_tensor(unitary(D, x)) for x in X]
    n = [0]

    def fb(G):
        def f(C):
            n[0] += len(C)
            return np.array([-np.sqrt(abs(O.overlap_eta_arpack(A[b // G], O.unitary_to_tensor(unitary(D, C[b])), WW)[0]))
                             for b in range(len(C))])
        return f
    t = time.perf_counter()
    res = batched_bfgs(fb(2 * P + 1), fb(8), X, maxiter=maxiter)
    dt = time.perf_counter() - t
    out = {'value': budget_T / dt, 'unit': 'trajectory time steps/s', 'cores': 1, 'kind': 'port',
           'sample': f'{budget_T} trajectories x 1 time step of the same lock-step BFGS (maxiter {maxiter}), {n[0]} objective evaluations, each '
                     f'ARPACK (scipy eigs, operator form: what xmps Map.right_fixed_point runs for the reference) on the {D * D}-dimensional map '
                     '+ the oracle\'s gate-by-gate circuit for parameters -> tensor; numpy, 1 thread',
           'objective_evals_per_s': n[0] / dt, 'iterations': int(res['nit']), 'mean_final_objective': float(res['fun'].mean())}
    # the reference's minimiser itself on trajectory 0 of the same sample (scripts/loschmidt.py:371: minimize(obj, params, (A_, WW)) -
    # scipy BFGS, forward differences, Wolfe search): how far the lock-step minimum is from scipy's on the same objective
    from scipy.optimize import minimize
    n[0] = 0
    t = time.perf_counter()
    f0 = fb(1)
    sp = minimize(lambda p: float(f0(p[None])[0]), X[0].copy(), method='BFGS', options={'maxiter': maxiter})
    out.update({'scipy_bfgs_final_objective': float(sp.fun), 'lockstep_final_objective_same_trajectory': float(res['fun'][0]),
                'scipy_bfgs_s': time.perf_counter() - t, 'scipy_bfgs_nfev': int(n[0]), 'scipy_bfgs_nit': int(sp.nit),
                'scipy_bfgs_what': 'scipy.optimize.minimize(method="BFGS") - the reference\'s per-step call - on trajectory 0 of this sample, same start, same oracle objective'})
    return out


def main_evolve(args):
    """--workload evolve: BASELINE.json configs[4] as it is worded - TFIM quench TIME EVOLUTION at D = 16, depth 4, independent
    trajectories per GPU.  One step = one TIME STEP of all T trajectories (qmps/new_time_evolve.py:276-292,
    scripts/loschmidt.py:367-375): reference tensors A_t = tensor(params_t) built on the device, then the minimiser the reference
    runs per step (scipy BFGS with finite-difference gradients) in lock-step over the trajectories: per iteration one device
    batch of T (2P + 1) central-difference candidates and one of T x 8 backtracking candidates - parameters -> tensor ->
    dominant eigenvalue of the mixed transfer map -> -sqrt|eta| - warm-started from the fixed points resident in the candidates'
    slots.  `value` = trajectory time steps per second.  Independent trajectories: replicas only at N > 1, no collective."""
    world, rank, local_rank = world_of(args)
    D, T = args.D, args.batch
    depth = {2: 4, 4: 2, 8: 3, 16: 4}[D]          # D = 2: scripts/loschmidt.py evolves ShallowCNOTStateTensor(2, .) with 8 angles
    P = 2 * depth
    full = D == 2 and args.ansatz == 'shallow-full'      # qmps/new_time_evolve.py:186-187: ShallowFullStateTensor(2, .), 15 angles
    if full:
        P = 15
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.path.isdir('/sys/class/net/lo'):
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from scipy.linalg import expm
    WW = expm(-1j * args.dt * tfim_h(1.0))
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = evolve_cpu_baseline(D, P, WW, args.seed, args.bfgs_iters, full=full)
    if args.carry_hessian is None:
        # measured (profiles/archive/r03h_evolve_*.json): D = 16 3.2 against 9.5 iterations per time step, D = 8 6.8 against 13 - but
        # D = 4 19.6 against 12 and D = 2 no gain: the shallow ansaetze of D = 2, 4 have flat directions a carried Hessian mis-scales
        args.carry_hessian = D >= 8
    from qmps_amd import _lib
    from qmps_amd.new_time_evolve import LockstepEvolver
    from qmps_amd.represent import ShallowCNOTStateTensor, ShallowFullStateTensor
    ev = LockstepEvolver(D, T, P, ShallowFullStateTensor if full else ShallowCNOTStateTensor, tol=args.tol, maxiter=args.bfgs_iters, device=local_rank,
                         gradient=args.gradient, first_rungs=2 if (args.gradient != 'fd' and (args.python_driver or args.no_speculative)) else None, carry_hessian=args.carry_hessian,
                         speculative=args.gradient != 'fd' and not args.no_speculative, native=not args.python_driver, device_driver=not args.host_driver)
    native = ev.native            # the whole timed region is ONE C call (qmps_evolve_bfgs); else: the numpy loop, one ctypes call per batch
    info = _lib.device_info(local_rank)
    X = np.random.default_rng(args.seed + rank).standard_normal((T, P))
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        ev.fg.eng.probe_fp64_tflops()
    if native and args.warmup > 0:
        X = ev.steps(X, WW, args.warmup)['x']
    else:
        for _ in range(args.warmup):
            X = ev.step(X, WW)['x']
    ev.fg.eng.overlap_stats(reset=True)
    ev.fl.eng.overlap_stats(reset=True)
    ev.fg.kernel_ms, ev.fl.kernel_ms = [], []
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    nit, nfev, f_last = [], 0, None
    if native:
        # the timed region runs WITHOUT instrumentation (a pair of HIP event records around a gradient batch costs the stream ~12 us,
        # 6 % of a time step at 256 trajectories); the kernel times come from an instrumented pass over the next time steps (below)
        res = ev.steps(X, WW, args.steps, counters=False)
        X, nit, f_last = res['x'], [int(n) for n in res['nit']], res['fun'][-1]
    else:
        for _ in range(args.steps):
            res = ev.step(X, WW)
            X = res['x']
            nit.append(res['nit'])
            nfev += res['nfev']
            f_last = res['fun']
    ev.fg.eng.sync()
    ev.fl.eng.sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    elapsed_instr = elapsed
    block_s = [elapsed]
    if native and not args.no_repeats:
        # the timed block repeated (the evolution goes on: the next `steps` time steps of the same trajectories), barrier + sync around each
        Xr = X
        for _ in range(4):
            if dist is not None:
                dist.barrier()
            tb = time.perf_counter()
            Xr = ev.steps(Xr, WW, args.steps, counters=False)['x']
            ev.fg.eng.sync()
            if dist is not None:
                dist.barrier()
            eb = time.perf_counter() - tb
            if dist is not None:
                import torch
                t = torch.tensor([eb], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                eb = float(t.item())
            block_s.append(eb)
        X = Xr
    if native:
        # instrumented pass: the NEXT args.steps time steps of the same trajectories with HIP events around every gradient batch
        # and the batch counters; the roofline figures, the solver statistics and the kernel share of wall time are this pass's
        ev.fg.eng.overlap_stats(reset=True)
        ev.fg.kernel_ms = []
        t2 = time.perf_counter()
        res2 = ev.steps(X, WW, args.steps)
        ev.fg.eng.sync()
        elapsed_instr = time.perf_counter() - t2
        nfev = res2['nfev']
    sg = ev.fg.eng.overlap_stats()
    if getattr(ev, 'device', False):
        # device-resident optimiser: one launch, its own counters (squarings summed by the kernel).  D = 2: every candidate is eigen-solved;
        # D = 4: only the iterates are (their 2P neighbours go through the two-sided quotient) - the set-up flops are counted for the
        # solved candidates only (scipy's nfev / (2P + 1); the backtracking points of rejected steps are not in that count: a lower bound)
        solved = res2['nfev'] if D == 2 else res2['nfev'] // (2 * P + 1)
        sg = {'evaluations': solved, 'rounds_sum': res2['squarings'], 'rounds_max': 0, 'not_converged': res2['failed_evaluations']}
    # (native driver: one context, its statistics pool the - rare - ladder batches with the gradient batches)
    sl = ev.fl.eng.overlap_stats() if ev.fl is not ev.fg else {k: 0 for k in sg}
    kms_timed = (list(ev.fg.kernel_ms), list(ev.fl.kernel_ms) if ev.fl is not ev.fg else [])
    device_busy = None
    if native:
        # D = 8, 16 (device-resident algebra): the device time of an UN-instrumented pass - one event pair per time step, first kernel to
        # last - over the wall time of the same pass: how much of a time step the device works
        if D in (8, 16) and not getattr(ev, 'device', False) and os.environ.get('QMPS_EVOLVE_HOST_ALGEBRA') is None:
            t3 = time.perf_counter()
            res3 = ev.steps(res2['x'], WW, args.steps, time_steps=True)
            ev.fg.eng.sync()
            e3 = time.perf_counter() - t3
            device_busy = {'device_ms_per_step': res3['device_ms'] / args.steps, 'wall_ms_per_step': e3 / args.steps * 1e3, 'share': res3['device_ms'] * 1e-3 / e3,
                           'what': 'one HIP event pair per time step (QMPS_BFGS_TIME_STEPS) around everything the step enqueues - evaluations, step kernels, '
                                   'idle launches at a chain\'s tail - in a pass without any other instrumentation; the host gap between two time steps is outside '
                                   '(lock-step groups, T >= 512: summed over the groups\' streams, which overlap - the share then exceeds 1)'}
    identity_leg = None
    if args.carry_hessian and not args.no_extras:
        # the same time steps the way scipy (the reference) starts them: inverse Hessian = identity at the top of every step
        ev.carry_hessian = False
        n_leg = max(2, min(4, args.steps))
        Xl = X.copy()
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
        nit_l, f_l = [], None
        if native:      # one C call, un-instrumented, like the timed region
            res = ev.steps(Xl, WW, n_leg, counters=False)
            ev.fg.eng.sync()
            Xl, f_l, nit_l = res['x'], res['fun'][-1], [int(n) for n in np.atleast_1d(res['nit'])]
        else:
            for _ in range(n_leg):
                res = ev.step(Xl, WW)
                Xl, f_l = res['x'], res['fun']
                nit_l.append(res['nit'])
        el = time.perf_counter() - t1
        ev.carry_hessian = True
        identity_leg = {'value': T * n_leg / el, 'unit': 'trajectory time steps/s (this rank)', 'steps': n_leg, 'ms_per_step': el / n_leg * 1e3,
                        'bfgs_iterations_per_step': float(np.mean(nit_l)), 'mean_final_objective': float(np.nanmean(f_l)),
                        'what': 'BFGS restarted from the identity at every time step (scipy / the reference); same tolerance, same ladder'}
    ev.fg.kernel_ms = kms_timed[0]
    ladder_ms = kms_timed[1]
    if rank == 0:
        squaring = D in (2, 4)
        per_round = 8 * (D * D) ** 3 if squaring else 64 * D ** 3           # a squaring of the complex D^2 x D^2 matrix / a power step (8 complex D^3 products)
        setup = 64 * D ** 3 + (32 * D ** 4 if squaring else 128 * D * D)
        kms = np.array(ev.fg.kernel_ms)
        two_sided = ev.two_sided
        # two-sided gradient: besides the two solves per iterate, 2P neighbours x (merge(B', B'): 4 complex D^3 products + the contraction
        # with G) and per iterate G_s = y^+ C_s r (12 products + the set-up of C_s)
        n_iter_evals = sg['evaluations'] // 2 if two_sided else 0
        flops_g = sg['rounds_sum'] * per_round + sg['evaluations'] * setup + n_iter_evals * (2 * P * (32 * D ** 3 + 32 * D * D) + 16 * 8 * D ** 3)
        tflops = flops_g / max(kms.sum() * 1e-3, 1e-12) * 1e-12
        byts = sg['evaluations'] * (32 * D * D + 16 + (32 * D * D if not squaring else 0))
        kernel_total_ms = float(kms.sum() + np.sum(ladder_ms))
        out = {'metric': f'time-evolution trajectory steps/sec at D={D}, depth={depth}, {T} trajectories per GPU',
               'value': world * T * args.steps / elapsed, 'unit': 'trajectory time steps/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': f'TFIM g=1 quench time evolution, D={D}, ' + (f'ShallowFull (15 parameters)' if full else f'ShallowCNOT depth {depth} ({P} parameters)') + f', {T} independent trajectories per GPU '
                                      f'from random parameters, W = exp(-{args.dt:g} i h), one step = one time step of every trajectory: lock-step BFGS '
                                      f'(<= {args.bfgs_iters} iterations, gtol 1e-5, ' + ('inverse Hessians carried from time step to time step, '
                                                                                          if args.carry_hessian else 'identity start at every time step, ') +
                                      'central differences h = 1e-6 ' +
                                      ('from one right + one left eigen-solve per iterate (neighbours to second order in h), ' +
                                       ('full step evaluated with its gradient first, ladder only on rejection' if ev.speculative else 'ladder in two stages (2 + 6 rungs)') if two_sided
                                       else 'with every neighbour eigen-solved, 8-point backtracking ladder') + '), objective '
                                      f'-sqrt|eta| with eta to {args.tol:g} (residual of the power method / rank-one test of the squaring)',
                          'baseline_config': 'BASELINE.json configs[4]', 'D': D, 'trajectories_per_gpu': T, 'n_params': P, 'seed': args.seed,
                          'bfgs_iterations_per_step': float(np.mean(nit)), 'carry_hessian': bool(args.carry_hessian),
                          'lockstep_groups': (ev.fg.eng.evolve_groups(T) if (native and not getattr(ev, 'device', False)) else 1),
                          'driver': (('qmps_evolve_bfgs_device: the optimiser on the device, ' + ('a workgroup of one to three waves per trajectory (a quad of lanes per candidate)' if D == 2 else 'a workgroup of eight waves per trajectory (wave 0 eigen-solves the point, the others probe its neighbours)') + ', the whole timed region is ONE LAUNCH') if getattr(ev, 'device', False) else
                                     ('qmps_evolve_bfgs: the whole timed region is one C call; optimiser algebra in kernels on device-resident state, the host enqueues chains of iterations '
                                      '(QMPS_EVOLVE_HOST_ALGEBRA: the round-4 host loop)' if (D in (8, 16) and os.environ.get('QMPS_EVOLVE_HOST_ALGEBRA') is None) else
                                      'qmps_evolve_bfgs: the whole timed region is one C call (host loop between the batches)')) if native else 'numpy loop (tools.batched_bfgs), one ctypes call per batch',
                          'adaptive_gradient': bool(getattr(ev, 'adaptive_gradient', False)),
                          'adaptive_gradient_rule': 'eigen-solves of a trajectory\'s gradient stop at residual clamp(1e-3 max|g|, 1e-8, 1e-6) (QMPS_BFGS_ADAPTIVE_GRADIENT); objective by the two-sided quotient, error <= 1e-12' if getattr(ev, 'adaptive_gradient', False) else None,
                          'objective_evals_per_step': nfev / args.steps,
                          'objective_evals_per_s': world * nfev / elapsed,
                          'mean_final_objective': float(np.nanmean(f_last)), 'worst_final_objective': float(np.nanmax(f_last)),
                          'solver_rounds_mean_gradient_batches': sg['rounds_sum'] / max(1, sg['evaluations']), 'solver_rounds_max_gradient_batches': sg['rounds_max'],
                          'solver_rounds_mean_ladder_batches': sl['rounds_sum'] / max(1, sl['evaluations']), 'solver_rounds_max_ladder_batches': sl['rounds_max'],
                          'not_converged': sg['not_converged'] + sl['not_converged'],
                          'kernel_share_of_wall': kernel_total_ms * 1e-3 / elapsed_instr,
                          'kernel_share_of_wall_what': 'gradient-evaluation kernels (HIP event pairs) over the wall time of the INSTRUMENTED pass, which synchronises after every evaluation to read its events; see device_busy for the un-instrumented run',
                          'device_busy': device_busy if native else None,
                          'instrumented_pass_ms_per_step': elapsed_instr / args.steps * 1e3,
                          'collective': 'none: independent trajectories (replicas only)', 'device': info['name'], 'arch': info['arch']},
               'roofline': {'bound': 'fp64_matrix' if D == 16 else ('fp64_matrix' if D == 4 else 'fp64_valu'), 'achieved': tflops, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': tflops / FP64_PEAK_TFLOPS, 'traffic': None,
                            'kernel': _dominant_kernel(ev.fg.eng, f'evolve_bfgs_d{D}_kernel') if getattr(ev, 'device', False) else ev.fg.eng.kernel_time(1)[1], 'kernel_ms': float(kms.mean()), 'launches': int(len(kms)),
                            'kernel_ms_from': ('HIP events around EVERY gradient evaluation of an instrumented pass over the time steps that follow the timed region (same trajectories, same number of steps; the timed region itself runs without event records)' if native else 'HIP events around EVERY gradient evaluation of the timed region') + ' (sum of durations / launches)' +
                                              (': right solve + left solve + neighbour tensors + G + probes' if two_sided else ': the overlap kernel of the T (2P+1) candidates'),
                            'note': (f'dominant work = the gradient evaluation ({2 * T} eigen-solves + {2 * P * T} neighbour probes per launch); ' if two_sided else
                                     f'dominant kernel = the overlap kernel of the gradient batches (T (2P+1) = {T * (2 * P + 1)} candidates per launch); ') +
                                    f'executed FLOPs = rounds x {per_round} + evaluations x {setup} (+ probes) with rounds summed by the kernels themselves (qmps_overlap_stats) over the same launches',
                            'groups_note': 'lock-step groups run on their own streams and overlap: kernel_ms sums their launches, so `achieved` (FLOPs / summed kernel time) is a per-stream rate, a lower bound of the device rate, and kernel_share_of_wall can exceed 1',
                            'hbm': {'achieved': byts / max(kms.sum() * 1e-3, 1e-12) * 1e-9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                    'frac': byts / max(kms.sum() * 1e-3, 1e-12) * 1e-9 / HBM_PEAK_GBPS,
                                    'note': 'candidate tensor in + fixed point in and out (warm start) + eta / objective / status out per evaluation; the reference tensor is shared by a group'}},
               'cpu_baseline': cpu}
        if identity_leg is not None:
            out['identity_start'] = identity_leg
        if len(block_s) > 1:
            vals = [world * T * args.steps / b for b in block_s]
            out['repeats'] = {'blocks': len(block_s), 'steps_per_block': args.steps, 'value_median': float(np.median(vals)), 'value_min': float(min(vals)),
                              'value_max': float(max(vals)), 'what': 'the timed block of --steps time steps repeated back to back on the evolving trajectories (block 0 is `value`)'}
        emit(args, out)
    ev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
