"""benchlib.common - what the four workloads of bench.py share: the synthetic inputs, the algorithmic byte / flop counts of SURVEY 8(d), the
CPU baselines, the launcher plumbing (world, self-launch, shard plans, the RCCL communicator) and the JSON sink.  Split out of bench.py in
round 5 (the file had grown to 1 400 lines); bench.py re-exports every name, the contract and the command lines did not change."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BENCH_PY = os.path.join(ROOT, 'bench.py')

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
    elif solver == 'plain' and D == 4 and not os.environ.get('QMPS_POWER_LANE'):
        # env_power_d4_kernel (round 6): the map as a real 16 x 16 matrix, one row per lane: 16 rows x (4 x 4 + 12 x 8) FMA to build it, then per
        # power step and lane 16 FMA (mat-vec) + 4 (trace) + ~9 flop (normalisation, distance) - K + 1 steps (the convergence test runs one step
        # behind) - and the energy-only kernel: density matrix 2432 FMA, LDL^H ~60, energy 28 per term
        k = iters.astype(np.float64) + 1.0
        flops = float((2.0 * 1792 + k * 16.0 * (2.0 * 16 + 2.0 * 4 + 9.0) + 2.0 * (2432 + 60 + 28)).sum())
        note = ('executed algorithm of env_power_d4_kernel + energy_only_d4_kernel: real 16 x 16 transfer matrix (3584 flop) + (K_b + 1) power steps in '
                'real coordinates (784 flop each: 512 of the mat-vec, the rest trace normalisation and the Frobenius distance) + density matrix / '
                'LDL^H / energy (5040); K_b read back per item.  SURVEY 8(d) prices the same step in operator form (A r A^+) at 2112 flop')
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
           '--master-port', str(port), BENCH_PY] + sys.argv[1:]
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
