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


def test_bench_distributed_code_path_world1():
    """bench.py's N > 1 branch (gloo rendezvous, unique-id broadcast, RCCL init, all-reduce per step),
    forced at world size 1 through the same launcher the driver uses."""
    env = dict(os.environ, QMPS_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3',
           '--warmup', '1', '--no-cpu-baseline', '--batch', '4096']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['value'] > 1e6 and 'RCCL' in d['config']['collective']
