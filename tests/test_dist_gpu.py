"""GPU (one device): the RCCL communicator of libqmps_hip at world size 1 - unique id, init, the
all-reduce of a host vector, and the asynchronous device-side cost + all-reduce used by bench.py.
(N > 1 runs only under the driver; the control flow is covered on CPU in test_dist_cpu.py.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import qmps_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_allreduce_and_cost():
    from qmps_amd import EnergyEngine
    rng = np.random.default_rng(0)
    A = O.unitary_to_tensor(O.haar_unitaries(rng, 8, 500))
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})])
    with EnergyEngine(4, 1024) as eng:
        uid = EnergyEngine.comm_unique_id()
        assert len(uid) == 128
        eng.comm_init(uid, 0, 1)
        v = eng.allreduce_sum([1.5, -2.0, 3.25])
        assert np.array_equal(v, [1.5, -2.0, 3.25])
        E, _, _ = eng.energies(A, h)
        eng.cost_launch()
        cost = eng.get_cost()
        assert np.allclose(cost, E.sum(0), rtol=0, atol=1e-9)
        assert np.allclose(eng.allreduce_cost(), cost, rtol=0, atol=0)
        eng.comm_destroy()


@pytest.mark.parametrize('with_comm', [True, False])
def test_grouped_exchange_keeps_every_step_cost(with_comm):
    """qmps_set_exchange_period: the summed costs of several steps share one all-reduce; get_cost always returns the
    cost of the LAST step whether the group is full (3 steps of period 3), partly filled (2 of 3) or the period has
    just been changed; more steps than ring slots x period wrap the ring."""
    from qmps_amd import EnergyEngine
    rng = np.random.default_rng(3)
    h = np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'ZZ': 0.5})])
    batches = [O.unitary_to_tensor(O.haar_unitaries(rng, 4, 96)) for _ in range(5)]
    with EnergyEngine(2, 128) as eng:
        if with_comm:
            eng.comm_init(EnergyEngine.comm_unique_id(), 0, 1)
        eng.set_hamiltonian(h)
        sums = []
        for A in batches:
            E, _, _ = eng.energies(A, h)
            sums.append(E.sum(0))
        eng.set_exchange_period(3)
        seen = []
        for k in range(29):                       # 29 steps: 9 full groups (> 4 ring slots) + a partial one
            eng.set_tensors(batches[k % 5])
            eng.launch()
            eng.cost_launch()
            if k in (2, 4, 13, 27, 28):
                seen.append((k, eng.get_cost()))
        for k, c in seen:
            assert np.allclose(c, sums[k % 5], rtol=0, atol=1e-10), k
        eng.set_exchange_period(1)
        eng.set_tensors(batches[1]); eng.launch(); eng.cost_launch()
        assert np.allclose(eng.get_cost(), sums[1], rtol=0, atol=1e-10)
        with pytest.raises(Exception):
            eng.set_exchange_period(17)
        if with_comm:
            eng.comm_destroy()


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_distributed_code_path_world1(scaling):
    """bench.py's N > 1 branch (gloo rendezvous, unique-id broadcast, RCCL init + ncclCommCount check, one all-reduce
    per step, the grouped-exchange extra), forced at world size 1 through the same launcher the driver uses."""
    env = dict(os.environ, QMPS_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3',
           '--warmup', '1', '--no-cpu-baseline', '--batch', '4096', '--rotate', '3', '--scaling', scaling]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['value'] > 1e6 and 'RCCL communicator of 1 ranks' in d['config']['collective']
    assert 'per step' in d['config']['collective'] and d['scaling'] == scaling and d['config']['resident_batches'] == 3
    assert d['grouped_exchange_16']['evals_per_s'] > 1e6
    assert d['exchange_bound'] is None and d['value_median'] > 1e6       # (world size 1: nothing to be bound by)


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_two_ranks_on_one_device(scaling):
    """Two ranks through the driver's launcher, both on device 0 (QMPS_BENCH_ONE_DEVICE): RCCL refuses two ranks on one
    GPU ('Duplicate GPU detected'), so this covers everything of bench.py's N = 2 branch EXCEPT the RCCL exchange itself -
    rendezvous, shard plan, every rank reaching the collective calls in step, the reported (never silent) fall-back of the
    summed cost to the launcher's gloo group, max-over-ranks timing, one JSON line from rank 0, clean exit of both ranks."""
    import json
    env = dict(os.environ, QMPS_BENCH_ONE_DEVICE='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29519', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5',
           '--warmup', '2', '--no-cpu-baseline', '--no-extras', '--batch', '4096', '--rotate', '2', '--scaling', scaling]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                     # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == scaling and d['value'] > 1e6
    per_gpu = 4096 if scaling == 'weak' else 2048
    assert d['config']['batch_per_gpu'] == per_gpu and d['config']['global_batch'] == 2 * per_gpu
    coll = d['config']['collective']
    if 'RCCL communicator of 2 ranks' not in coll:             # (a box with two GPUs would take the RCCL path)
        assert 'RCCL communicator unavailable on 2 rank(s)' in coll and 'gloo' in coll
    assert np.isfinite(d['summed_cost']) and d['config']['not_converged_or_not_pd'] == 0
    # round 5: at N > 1 the line says at its top level whether the exchange or the kernel paces a step, and carries the median of the blocks
    assert d['exchange_bound'] in (True, False) and 'exchange_bound_rule' in d and 'host_wait_ms' in d and d['value_median'] > 1e6


def test_bench_two_ranks_with_extras():
    """The same with the informational legs enabled: the rank-0-only extras must not issue collective calls (round 2's advisor
    finding: a cost exchange inside a rank-0-only leg would leave unmatched all-reduces behind)."""
    import json
    env = dict(os.environ, QMPS_BENCH_ONE_DEVICE='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29521', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5',
           '--warmup', '2', '--no-cpu-baseline', '--batch', '4096', '--rotate', '2']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and 'warm_start' in d and 'power_iteration' in d and 'contraction_only' in d


def test_bench_extras_with_a_communicator_world1():
    """Extras enabled WITH a live RCCL communicator (world size 1 through the launcher): the warm-start leg runs without a
    cost exchange, the run completes and reports it."""
    import json
    env = dict(os.environ, QMPS_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29523', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5',
           '--warmup', '2', '--no-cpu-baseline', '--batch', '4096', '--rotate', '2']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert 'RCCL communicator of 1 ranks' in d['config']['collective'] and d['warm_start']['accepted_fraction'] == 1.0


@pytest.mark.parametrize('nproc', [1, 2])
def test_bench_sharded_rotosolve(nproc):
    """--workload rotosolve --shard (BASELINE.json configs[3]: restarts sharded over the GPUs, RCCL all-reduce of the sweep costs):
    world size 1 with a real RCCL communicator; two ranks on one device (RCCL refuses a duplicate GPU: the reported gloo
    fall-back) - shard plan, exchange inside the timed region, one JSON line."""
    import json
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    env['QMPS_BENCH_FORCE_DIST' if nproc == 1 else 'QMPS_BENCH_ONE_DEVICE'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr',
           '127.0.0.1', '--master-port', str(29525 + nproc), os.path.join(ROOT, 'bench.py'), '--gpus', str(nproc), '--workload', 'rotosolve',
           '--D', '8', '--batch', '768', '--shard', '--steps', '4', '--warmup', '1', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == nproc and d['scaling'] == 'strong' and c['sharded'] and c['restarts_global'] == 256
    assert c['restarts_this_rank'] == 256 // nproc and c['restarts_counted_all_ranks'] == 256
    assert c['best_energy_all_ranks'] <= c['best_energy'] + 1e-12 and np.isfinite(c['summed_cost_last_sweep_all_ranks'])
    if nproc == 1:
        assert 'RCCL communicator of 1 ranks' in c['collective']
    elif 'RCCL communicator of 2 ranks' not in c['collective']:
        assert 'unavailable' in c['collective'] and 'gloo' in c['collective']


def test_bench_evolve_two_ranks_on_one_device():
    """--workload evolve through the driver's launcher with two ranks (BASELINE.json configs[4]: trajectories sharded over the
    GPUs, no data-path collective): each rank evolves its own trajectories with the one-call native driver, the launcher's
    gloo group carries the barriers and the max-over-ranks time, rank 0 prints the one JSON line with the whole-job rate."""
    import json
    env = dict(os.environ, QMPS_BENCH_ONE_DEVICE='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29531', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'evolve', '--D', '16',
           '--batch', '32', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-extras']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and c['trajectories_per_gpu'] == 32 and 'qmps_evolve_bfgs' in c['driver']
    assert d['value'] > 0 and c['not_converged'] == 0 and c['mean_final_objective'] < -0.999
    assert abs(d['value'] - 2 * 32 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']
