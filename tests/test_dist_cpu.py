"""CPU, world_size 2 (gloo): the N > 1 control flow - contiguous shards of the batch axis, no
data-path collective, one all-reduce of the summed cost - with the oracle as the rank-local
evaluator (the HIP engine plays that role on the GPU box)."""
import os
import socket

import numpy as np
import pytest

from qmps_amd.dist import shard_bounds


def test_shard_bounds_partition():
    for B in (0, 1, 7, 64, 65536, 768):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, out):
    import torch.distributed as dist
    from oracle import c_oracle as C
    from oracle import qmps_oracle as O
    from qmps_amd.dist import ShardedCost
    from tests.gloo_reducer import GlooReducer
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)                       # same global batch on every rank
        A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
        h = np.stack([O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}), O.hamiltonian_matrix({'ZZ': -1, 'X': 1})])
        cost = ShardedCost(rank, world, GlooReducer(), lambda a: C.energy_batch(a, h)['E'])
        total = cost(A)
        out[rank] = total
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _plan_worker(rank, world, port, scaling, batch, out):
    """bench.py's shard plan on every rank + one gloo all-reduce of the local summed cost: what `--scaling strong`
    (global batch split B/G) and `--scaling weak` (batch per GPU) evaluate, with the oracle as the local evaluator."""
    import torch
    import torch.distributed as dist
    from oracle import c_oracle as C
    from oracle import qmps_oracle as O
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, n, global_batch = _bench_module().shard_plan(scaling, batch, rank, world)
        rng = np.random.default_rng(7)
        A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, global_batch))        # the same global batch on every rank
        h = O.hamiltonian_matrix({'ZZ': -1, 'X': 1})
        local = C.energy_batch(A[lo:lo + n], h)['E'].sum(0) if n else np.zeros(1)
        t = torch.tensor(local)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out[rank] = (lo, n, global_batch, t.numpy().copy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('scaling,batch', [('strong', 37), ('strong', 64), ('weak', 20)])
def test_bench_shard_plan_world2(scaling, batch):
    import torch.multiprocessing as mp
    from oracle import c_oracle as C
    from oracle import qmps_oracle as O
    C.build()
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_plan_worker, args=(world, port, scaling, batch, out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    global_batch = batch if scaling == 'strong' else world * batch
    assert all(r[2] == global_batch for r in res)
    assert res[0][0] == 0 and res[0][0] + res[0][1] == res[1][0] and res[1][0] + res[1][1] == global_batch   # contiguous, complete
    if scaling == 'strong':
        assert abs(res[0][1] - res[1][1]) <= 1                    # B/G per GPU
    else:
        assert res[0][1] == res[1][1] == batch
    rng = np.random.default_rng(7)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, global_batch))
    expect = C.energy_batch(A, O.hamiltonian_matrix({'ZZ': -1, 'X': 1}))['E'].sum(0)
    assert np.allclose(res[0][3], res[1][3], rtol=0, atol=0) and np.allclose(res[0][3], expect, rtol=0, atol=1e-11)


@pytest.mark.parametrize('B', [101, 2])
def test_sharded_cost_world2(B):
    import torch.multiprocessing as mp
    from oracle import c_oracle as C
    from oracle import qmps_oracle as O
    C.build()
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, B, out), nprocs=world, join=True)
        res = [np.asarray(out[r]) for r in range(world)]
    rng = np.random.default_rng(42)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, B))
    h = np.stack([O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5}), O.hamiltonian_matrix({'ZZ': -1, 'X': 1})])
    expect = C.energy_batch(A, h)['E'].sum(0)
    assert np.allclose(res[0], res[1], rtol=0, atol=0)        # every rank holds the same reduced cost
    assert np.allclose(res[0], expect, rtol=0, atol=1e-11)


def _roto_shard_worker(rank, world, port, R_global, out):
    """bench.py --workload rotosolve --shard on every rank, with the oracle as the local evaluator: this rank's contiguous
    block of the ONE global set of restarts, sweep energies of the block, then the path's exchange step
    (qmps_amd.dist.reduce_sweep_costs: all-reduce sums of the per-sweep costs + all-reduce min of the best energy)."""
    import torch.distributed as dist
    from oracle import c_oracle as C
    from oracle import qmps_oracle as O
    from qmps_amd.dist import reduce_sweep_costs
    from tests.gloo_reducer import GlooReducer
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        first, R, R_all = _bench_module().rotosolve_shard_plan(R_global, rank, world, True)
        hist = _sweep_energies(C, O, R_all)[:, first:first + R]
        out[rank] = (first, R, R_all) + reduce_sweep_costs(hist, GlooReducer())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _sweep_energies(C, O, R_all, sweeps=19):
    """(sweeps, R_all) energies of R_all restarts along a deterministic parameter path (a stand-in for the sweeps of the
    optimiser: what matters here is which rank owns which column); restart 3 is given an invalid (NaN) entry."""
    P0 = np.random.default_rng(99).standard_normal((R_all, 4))
    h = O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})
    rows = []
    for k in range(sweeps):
        A = np.stack([O.unitary_to_tensor(O.shallow_cnot_unitary(4, p + 0.05 * k)) for p in P0])
        rows.append(C.energy_batch(A, h)['E'][:, 0])
    hist = np.array(rows)
    if R_all > 3:
        hist[5, 3] = np.nan
    return hist


@pytest.mark.parametrize('R_global', [11, 4])
def test_sharded_rotosolve_exchange_world2(R_global):
    import torch.multiprocessing as mp
    from oracle import c_oracle as C
    from oracle import qmps_oracle as O
    C.build()
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_roto_shard_worker, args=(world, port, R_global, out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert res[0][0] == 0 and res[0][0] + res[0][1] == res[1][0] and res[1][0] + res[1][1] == R_global      # contiguous, complete
    assert abs(res[0][1] - res[1][1]) <= 1 and res[0][2] == res[1][2] == R_global
    hist = _sweep_energies(C, O, R_global)
    ok = np.isfinite(hist).all(axis=0)
    for r in res:                                           # every rank holds the same reduced numbers
        assert np.allclose(r[3], hist[:, ok].sum(axis=1), rtol=0, atol=1e-11) and len(r[3]) == 19   # 19 sweeps + the count: two messages
        assert r[4] == int(ok.sum()) and abs(r[5] - hist[-1, ok].min()) < 1e-14
    # replicas (no --shard): every rank its own restarts
    plan = _bench_module().rotosolve_shard_plan
    assert plan(R_global, 1, 2, False) == (R_global, R_global, 2 * R_global)


def test_bench_self_launch_and_world_checks():
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself (a child
    torch.distributed.run created before anything touches a GPU) and exits with the children's return code - here, without a
    device, every rank stops at QMPS_ERR_NO_DEVICE (the product has no CPU fallback); every workload refuses a launch whose
    WORLD_SIZE differs from --gpus instead of reporting n_gpus = 1."""
    import subprocess
    import sys
    from qmps_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('a GPU is visible here: the ranks would run')
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--batch', '64'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert 'starting 2 ranks through torch.distributed.run' in out.stderr
    assert out.stderr.count('no usable gfx950 device') + out.stderr.count('no CPU fallback') >= 2, out.stderr[-3000:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]          # no JSON line from a run that computed nothing
    for wl in ('energy', 'overlap', 'evolve', 'rotosolve'):
        out = subprocess.run([sys.executable, bench, '--gpus', '1', '--workload', wl, '--no-cpu-baseline'],
                             env=dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=120)
        assert out.returncode != 0 and 'WORLD_SIZE=2 but --gpus 1' in out.stderr, (wl, out.stderr[-500:])


def _exchange_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        bench = _bench_module()
        # every rank times "its" region; the line carries the MAX over ranks (bench.py's contract) and rank 0's ring statistics
        ms = torch.tensor([0.030 + 0.010 * rank], dtype=torch.float64)
        dist.all_reduce(ms, op=dist.ReduceOp.MAX)
        ms_per_step = float(ms.item())
        value = world * 65536 / (ms_per_step * 1e-3)
        rep_hidden = bench.exchange_report(world, ms_per_step, 0.028, value, 0.0, 2000, 1.05 * value)
        rep_waiting = bench.exchange_report(world, ms_per_step, 0.028, value, 0.2 * ms_per_step * 2000, 2000, 1.05 * value)
        rep_grouped = bench.exchange_report(world, 0.060, 0.028, value, 0.0, 2000, 1.6 * value)
        out[rank] = (ms_per_step, rep_hidden, rep_waiting, rep_grouped)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_exchange_report_fields_world2():
    """bench.py's N > 1 top-level fields (host_wait_ms, grouped_exchange_16_evals_per_s, exchange_bound): computed alike on every rank of a
    world-size-2 gloo group from the max-over-ranks step time; at N = 1 they are None."""
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_exchange_worker, args=(2, port, out), nprocs=2, join=True)
        res = dict(out)
    assert res[0][0] == res[1][0] == pytest.approx(0.040)
    for r in (0, 1):
        _, hidden, waiting, grouped = res[r]
        assert hidden['exchange_bound'] is False and hidden['host_wait_ms'] == 0.0 and hidden['grouped_exchange_16_evals_per_s'] > 0
        assert waiting['exchange_bound'] is True
        assert grouped['exchange_bound'] is True
    one = _bench_module().exchange_report(1, 0.03, 0.028, 2e9, None, 2000, None)
    assert one == {'host_wait_ms': None, 'grouped_exchange_16_evals_per_s': None, 'exchange_bound': None}
