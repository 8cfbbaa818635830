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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == FP64 matrix peak (= 157.3 TF FP32 vector / 2, MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec
MALL_MIB = 256            # Infinity Cache


def flops_per_eval(D, K):
    """real fp64 FLOPs of one evaluation with K power steps."""
    return K * (32 * D ** 3 + 4 * D ** 2) + 64 * D ** 3 + 128 * D ** 2


def bytes_per_eval(D):
    """algorithmic HBM bytes per evaluation (A in, E out)."""
    return 32 * D * D + 8


def haar_tensors(seed, D, B):
    """Haar-random 2D x 2D unitaries qr(randn + i randn) (qmps/ansatze.py:30) -> A[b,s,i,j] = U[b,2i+s,j]."""
    rng = np.random.default_rng(seed)
    out = np.empty((B, 2, D, D), dtype=np.complex128)
    step = 8192
    for lo in range(0, B, step):
        n = min(step, B - lo)
        Z = rng.standard_normal((n, 2 * D, 2 * D)) + 1j * rng.standard_normal((n, 2 * D, 2 * D))
        Q, _ = np.linalg.qr(Z)
        out[lo:lo + n] = Q[:, :, :D].reshape(n, D, 2, D).transpose(0, 2, 1, 3)
    return out


def tfim_h(g=1.0):
    X = np.array([[0, 1], [1, 0]], dtype=complex)
    Z = np.array([[1, 0], [0, -1]], dtype=complex)
    I = np.eye(2, dtype=complex)
    return -np.kron(Z, Z) + 0.5 * g * (np.kron(I, X) + np.kron(X, I))


def xxz_h(delta=0.5):
    """Heisenberg XXZ two-site term XX + YY + delta ZZ (BASELINE.json configs[3]; Hamiltonian({'XX': 1, 'YY': 1, 'ZZ': delta}).to_matrix(),
    qmps/ground_state.py:73-88)."""
    X = np.array([[0, 1], [1, 0]], dtype=complex)
    Y = np.array([[0, -1j], [1j, 0]], dtype=complex)
    Z = np.array([[1, 0], [0, -1]], dtype=complex)
    return np.kron(X, X) + np.kron(Y, Y) + delta * np.kron(Z, Z)


def hamiltonian_of(args):
    """(h, description): --hamiltonian tfim|xxz; default = the one BASELINE.json names for the bond dimension (XXZ at D = 8)."""
    name = args.hamiltonian or ('xxz' if args.D == 8 else 'tfim')
    if name == 'xxz':
        return xxz_h(0.5), 'Heisenberg XXZ (XX + YY + 0.5 ZZ)'
    return tfim_h(1.0), 'TFIM g=1'


def committed_traffic(D, B, solver, store_env, rotate):
    """HBM bytes per step from the PMC passes committed under profiles/ (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE,
    collected in separate rocprofv3 --pmc runs of this very command, tools/prof.sh + tools/collect_profiles.py);
    None when no profile of this configuration is committed."""
    try:
        table = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    except Exception:
        return None
    for name in ((solver, 'squaring') if D == 16 else (solver,)):       # (D = 16 has no direct solve: 'direct' runs the iterative kernel)
        hit = table.get(f'D={D}|B={B}|solver={name}|store_env={int(store_env)}|rotate={rotate}')
        if hit is not None:
            return hit
    return None


# ---- CPU baselines (run BEFORE the GPU is initialised: the process-parallel leg forks) ------------------------
def _one_blas_thread():
    """numpy/scipy in a worker must not start its own thread pool: N workers x N BLAS threads would fight for N cores."""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass


def _ref_chunk(args):
    from oracle import qmps_oracle as O
    U, h = args
    t = time.perf_counter()
    for k in range(len(U)):
        O.reference_structured_energy(U[k], h)
    return time.perf_counter() - t


def effective_cpus():
    """(usable CPUs, explanation): os.cpu_count() reports the host's hardware threads; the container may be limited to
    fewer by its affinity mask or its cgroup CPU quota (cpu.max / cfs_quota_us)."""
    n = os.cpu_count() or 1
    why = [f'os.cpu_count() = {n}']
    try:
        a = len(os.sched_getaffinity(0))
        why.append(f'affinity mask = {a}')
        n = min(n, a)
    except Exception:
        pass
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: (t.split()[0], t.split()[1])),
                        ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', lambda t: (t.strip(), open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ('max', '-1'):
                q = float(quota) / float(period)
                why.append(f'cgroup quota = {q:.1f} CPUs ({path})')
                n = max(1, min(n, int(q + 0.5)))
            else:
                why.append(f'cgroup quota = unlimited ({path})')
            break
        except Exception:
            continue
    return n, ', '.join(why)


def cpu_baseline(D, A, h, max_iter, tol, budget_s=8.0):
    """The oracle ("port") timed on this box's host cores on a bounded sample of the same workload, plus:
    all host threads (OpenMP), and the reference-STRUCTURED numpy path (dense eig -> Cholesky -> null-space
    completion -> Kronecker state vector -> dense psi^+ (1 x h x 1) psi) on 1 core and
    process-parallel over all host cores."""
    import multiprocessing as mp
    from oracle import c_oracle as C
    from oracle import qmps_oracle as O
    C.build()
    n0 = min(len(A), 2048)
    t = time.perf_counter()
    C.energy_batch(A[:n0], h, max_iter=max_iter, tol=tol, threads=1)
    rate = n0 / (time.perf_counter() - t)
    reps = max(1, int(round(rate * budget_s / len(A))))
    t = time.perf_counter()
    for _ in range(reps):
        C.energy_batch(A, h, max_iter=max_iter, tol=tol, threads=1)
    v1 = reps * len(A) / (time.perf_counter() - t)
    cores, cores_why = effective_cpus()
    nthr = min(cores, C.max_threads())
    C.energy_batch(A[:4096], h, max_iter=max_iter, tol=tol, threads=nthr)      # thread pool start-up outside the timing
    t = time.perf_counter()
    C.energy_batch(A, h, max_iter=max_iter, tol=tol, threads=nthr)
    vall = len(A) / (time.perf_counter() - t)
    out_all = {'value': vall, 'threads': nthr, 'usable_cpus': cores, 'usable_cpus_from': cores_why,
               'speedup_over_1_thread': vall / v1, 'sample': f'all {len(A)} evaluations, OpenMP, dynamic schedule'}
    if vall < 0.5 * nthr * v1:
        out_all['note'] = (f'speed-up {vall / v1:.1f}x on {nthr} threads: below half of linear - the usable CPUs may be hardware threads '
                           'sharing cores, or throttled by the container (see usable_cpus_from)')
    # reference-structured numpy: one core, then a pool of forked workers over all cores
    nref = min(len(A), 600)
    U = np.stack([O.tensor_to_unitary(A[k]) for k in range(nref)])     # complete each tensor to a unitary (the reference's input)
    _one_blas_thread()                                                 # "1 core" means one thread
    t = time.perf_counter()
    for k in range(nref):
        O.reference_structured_energy(U[k], h)
    vref = nref / (time.perf_counter() - t)
    per_worker = max(50, int(vref * 4))                               # ~4 s of work per worker
    nproc = cores
    Up = U[np.arange(per_worker) % nref]
    ctx = mp.get_context('fork')
    try:
        with ctx.Pool(nproc, initializer=_one_blas_thread) as pool:
            pool.map(_ref_chunk, [(Up[:5], h)] * nproc)              # workers up and warm
            t = time.perf_counter()
            pool.map(_ref_chunk, [(Up, h)] * nproc, chunksize=1)
            vpar = nproc * per_worker / (time.perf_counter() - t)
        par = {'value': vpar, 'processes': nproc, 'speedup_over_1_core': vpar / vref,
               'sample': f'{nproc} forked workers x {per_worker} evaluations each (cycled over the first {nref} of the workload)'}
    except Exception as e:                                             # reported, never silent
        par = {'value': None, 'error': repr(e)}
    return {'value': v1, 'unit': 'two-site energy evals/s', 'cores': 1, 'kind': 'port',
            'sample': f'{reps} pass(es) over the {len(A)} evaluations of the first resident batch of the GPU workload, same seed, '
                      f'C oracle (oracle/qmps_oracle.c: plain power iteration + closed-form energy), 1 thread',
            'all_cores': out_all,
            'reference_structured_numpy': {'value': vref, 'cores': 1,
                                           'sample': f'first {nref} evaluations; dense eig + Cholesky + null-space completion + '
                                                     'Kronecker state vector (the reference\'s per-evaluation structure, numpy/scipy)',
                                           'all_cores': par}}


def squaring_schedule_ops(steps, skip, period, max_steps, e0_start):
    """(squarings, mat-vecs) env_square_d4_kernel executes for an item that reports `steps` power steps after the
    hand-off: replay of the kernel's schedule (include/qmps_hip.h QMPS_SKIP_ROUNDS_D4 / QMPS_MATVEC_PERIOD_D4)."""
    m = 0
    while m < skip and (2 << m) <= max_steps:
        m += 1
    nsq, nmv, count = m, 0, 0
    it = (1 << m) if (e0_start and m > 0) else 0
    while it < steps:
        it += 1 << m
        nmv += 1
        count += 1
        if it < steps and count == period and m < 29 and it + (2 << m) <= max_steps:
            nsq += 1
            m += 1
            count = 0
    return nsq, nmv


def executed_flops(D, solver, iters, eng, max_iter):
    """FLOPs of the algorithm the dominant kernel actually executed, from the iteration counts read back per item."""
    n2 = (D * D) ** 3
    direct = solver == 'direct' and D == 4
    hybrid = (solver == 'squaring' and D <= 4) or (solver == 'direct' and D == 2)
    handoff = eng.handoff if hybrid else 0
    if direct:
        # energy_direct_d4_kernel per evaluation (FMA = 2 flop), qmps_direct_core.h:
        #   real 16 x 16 transfer matrix: 16 rows x (4 x 4 + 12 x 8) FMA                        = 1792 FMA
        #   Gauss-Jordan on 16 x (16 + 1): 16 rows x sum_k (16 - k) FMA + 16 x 16 multipliers    = 2176 FMA + 256 mul
        #   acceptance power step from the tensor: 4 rows x (2 x 4 x 14 + 4 x 8 x 4) FMA         =  960 FMA
        #   two-site density matrix: B = A A 4 x 256, Y = B r 4 x 224, rho 4 x 128 FMA           = 2432 FMA
        #   LDL^H test ~60 FMA, energy 28 FMA per term
        # an evaluation that fell back (iters > 1) rebuilds R (3584 flop) and adds 2 x 16^3 + 2 x 16^2 flop per round
        per = 2.0 * (1792 + 2176 + 960 + 2432 + 60 + 28) + 256
        rounds = np.where(iters > 1, np.log2(np.maximum(iters - 1, 1)), 0.0)
        flops = float((per + rounds * (2.0 * n2 + 2.0 * (D * D) ** 2) + (iters > 1) * 3584.0).sum())
        note = ('executed algorithm of the fused kernel: real 16 x 16 transfer matrix (3584 flop) + Gauss-Jordan (4608) + '
                'acceptance power step (1920) + density matrix / LDL^H / energy (5040) = 15152 flop per evaluation; squaring '
                'rounds of fallen-back evaluations added from the iteration count read back per item')
    elif hybrid and D == 4:
        skip, period = eng.squaring_schedule
        skip = skip if handoff == 0 else 0
        sq_flops = np.zeros(len(iters))
        for k in np.unique(iters):
            if k <= handoff:
                continue
            nsq, nmv = squaring_schedule_ops(int(k) - handoff, skip, period, max_iter - handoff, handoff == 0)
            sq_flops[iters == k] = 32.0 * D ** 4 + nsq * 2.0 * n2 + nmv * 2.0 * (D * D) ** 2
        k_plain = np.minimum(iters, handoff).astype(np.float64)
        plain_flops = k_plain * (32 * D ** 3 + 4 * D ** 2)
        epilogue_flops = 64 * D ** 3 + 128 * D ** 2
        flops = float(sq_flops.sum()) if handoff == 0 else float((plain_flops + sq_flops + epilogue_flops).sum())
        note = ('executed algorithm: per item 32 D^4 (real transfer matrix) + n_sq 2 (D^2)^3 (squarings on the matrix cores) + '
                'n_mv 2 (D^2)^2 (mat-vecs with T^(2^m)); n_sq, n_mv replayed from the iteration count read back per item')
    elif hybrid:
        k_plain = np.minimum(iters, handoff).astype(np.float64)
        m_sq = np.where(iters > handoff, np.log2(np.maximum(iters - handoff, 1)), 0.0)
        sq_flops = np.where(iters > handoff, m_sq * 2.0 * n2 + 32.0 * D ** 4, 0.0)
        flops = float((k_plain * (32 * D ** 3 + 4 * D ** 2) + sq_flops + 64 * D ** 3 + 128 * D ** 2).sum())
        note = ('executed algorithm: m = log2(K) squarings of the real D^2 x D^2 transfer matrix per item (2 (D^2)^3 flop each) + '
                'its construction; K read back per item')
    elif solver == 'direct' and D == 8:
        # env_direct_d8: real 64 x 64 system - build 64 rows x 480 FMA = 30 720 FMA, Gauss-Jordan 64 pivots x 64 rows x ~34 FMA
        # = 139 264 FMA (DESIGN.md kernel table) - then the block kernel's acceptance step(s) and the energy epilogue
        flops = float((2.0 * (30720 + 139264) + flops_per_eval(D, iters.astype(np.float64))).sum())
        note = ('executed algorithm at D = 8: direct 64 x 64 real solve (2 x (30 720 + 139 264) flop) + SURVEY 8(d) K_b(32D^3+4D^2)+64D^3+128D^2 '
                'for the acceptance step(s) and the energies, K_b read back per item')
    else:
        flops = float(flops_per_eval(D, iters.astype(np.float64)).sum())
        note = 'SURVEY 8(d): sum_b [K_b(32D^3+4D^2)+64D^3+128D^2], K_b read back per item'
    return flops, note, handoff


def exchange_report(world, ms_per_step, kernel_ms, value, host_wait_ms, steps, grouped_16_evals_per_s):
    """N > 1: is a step paced by the per-step all-reduce or by the energy kernel?  Top-level fields of the line:
      host_wait_ms             rank 0, timed region: how long the host stood at the 8-slot ring waiting for the exchange that last used a slot
      grouped_exchange_16_evals_per_s   the same steps with ONE all-reduce per 16 steps (None if that extra did not run; its dict stays under `grouped_exchange_16`)
      exchange_bound           True when the exchange sets the pace: the host waited for more than a tenth of the timed region, or the step takes
                               more than 1.5 x its kernel AND grouping the exchange gains more than 15 %
    A step is ~30 us at the headline shape: with two communicators alternating, an all-reduce must complete within two steps to stay
    hidden."""
    if world <= 1:
        return {'host_wait_ms': None, 'grouped_exchange_16_evals_per_s': None, 'exchange_bound': None}
    waited = host_wait_ms is not None and host_wait_ms > 0.1 * ms_per_step * steps
    slow = kernel_ms is not None and kernel_ms > 0 and ms_per_step > 1.5 * kernel_ms
    gain = grouped_16_evals_per_s is not None and value > 0 and grouped_16_evals_per_s > 1.15 * value
    return {'host_wait_ms': host_wait_ms, 'grouped_exchange_16_evals_per_s': grouped_16_evals_per_s, 'exchange_bound': bool(waited or (slow and gain)),
            'exchange_bound_rule': 'host_wait_ms > 10 % of the timed region, or (ms_per_step > 1.5 x kernel_ms and grouped_exchange_16_evals_per_s > 1.15 x value)'}


def emit(args, out):
    """rank 0's ONE JSON line - or, when this workload runs as an `other_configs` entry of the default run, its dict"""
    sink = getattr(args, 'collect', None)
    if sink is not None:
        sink.append(out)
    else:
        print(json.dumps(out), flush=True)


def world_of(args):
    """(world, rank, local_rank) from the launcher's environment; every workload refuses a launch whose WORLD_SIZE is not --gpus
    (a `--gpus 8` line that silently ran one rank would report n_gpus 1 as if it were the 8-GPU number)"""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('QMPS_BENCH_ONE_DEVICE') == '1':      # functional test of the N > 1 branch on a one-GPU box
        local_rank = 0
    if world != args.gpus:
        sys.exit(f'bench.py: WORLD_SIZE={world} but --gpus {args.gpus}')
    return world, rank, local_rank


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks HERE - a child `python -m torch.distributed.run`
    created before this process has touched the GPU (it never does) - relay the child's stdout (rank 0's JSON line) and exit with
    its return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f'bench.py: --gpus {args.gpus} without a launcher: starting {args.gpus} ranks through torch.distributed.run (port {port})', file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=dict(os.environ), stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    sys.exit(proc.wait())


def shard_plan(scaling, batch, rank, world):
    """(first evaluation, evaluations on this rank, global batch).  weak: `batch` per GPU; strong: `batch` is the global
    batch and rank r owns the contiguous block qmps_amd.dist.shard_bounds(batch, r, world) (B/G per GPU)."""
    from qmps_amd.dist import shard_bounds
    if scaling == 'strong':
        lo, hi = shard_bounds(batch, rank, world)
        return lo, hi - lo, batch
    return rank * batch, batch, world * batch


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


def init_rccl(eng, dist, rank, world):
    """RCCL communicator for this rank's engine: rank 0 creates the unique id, the launcher's gloo group broadcasts it, every rank
    joins; all ranks then agree (gloo) on whether it worked.  Returns (ok, error text)."""
    import torch
    from qmps_amd import EnergyEngine, _lib
    err = ''
    try:
        ids = [EnergyEngine.comm_unique_id() if rank == 0 else None]
    except _lib.QmpsError as e:          # keep the ranks in step: everyone must reach the broadcast
        ids, err = [None], str(e)
    dist.broadcast_object_list(ids, src=0)
    if ids[0] is not None:
        try:
            eng.comm_init(ids[0], rank, world)
            if eng.comm_count() != world:
                err = f'communicator has {eng.comm_count()} ranks, expected {world}'
        except _lib.QmpsError as e:
            err = str(e)
    else:
        err = err or 'rank 0 could not create an RCCL unique id'
    flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.SUM)
    # what EVERY rank's communicator says about its size (ncclCommCount; 0 = that rank has none), gathered over gloo: `rccl_ranks_seen`
    seen = torch.zeros(world, dtype=torch.int64)
    try:
        seen[rank] = eng.comm_count() if not err or 'expected' in err else 0
    except _lib.QmpsError:
        pass
    dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    init_rccl.ranks_seen = [int(v) for v in seen]
    if flag.item() != 0.0:
        try:
            eng.comm_destroy()
        except _lib.QmpsError:
            pass
        return False, f'RCCL communicator unavailable on {int(flag.item())} rank(s) ({err or "see other ranks"})'
    return True, ''


def rotosolve_shard_plan(R_global, rank, world, shard):
    """(first restart, restarts on this rank, restarts in all).  --shard: the R_global restarts are split into contiguous
    blocks (qmps_amd.dist.shard_bounds; BASELINE.json configs[3]: "256 random restarts x 3 angle samples sharded over 8 MI355X");
    otherwise every rank runs its own R_global restarts (replicas)."""
    from qmps_amd.dist import shard_bounds
    if shard:
        lo, hi = shard_bounds(R_global, rank, world)
        return lo, hi - lo, R_global
    return rank * R_global, R_global, world * R_global


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
                          'D2_optimum': -1.269909412573 if (D == 2 and h_name.startswith('TFIM')) else None, 'ansatz': 'ShallowFullStateTensor' if full else 'ShallowCNOTStateTensor', 'depth': None if full else depth,
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
                                                                                'mean_energy_last_sweep', 'best_energy', 'exact_ground_state_energy', 'D2_optimum', 'ansatz', 'depth', 'not_converged_or_not_pd',
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
