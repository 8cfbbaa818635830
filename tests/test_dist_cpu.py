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
    from qmps_amd.dist import GlooReducer, ShardedCost
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
