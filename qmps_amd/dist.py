"""Multi-GPU layout of the hot path: one process per GPU, the batch of independent evaluations
(restarts x rotosolve shifts x Hamiltonian terms) split into contiguous shards, and ONE exchange
step - an all-reduce of the summed cost.

The reference has no distributed code; its only parallelism is joblib over independent
trajectories (poincare_map/2body_scars.py:445,607) - the same "independent units" pattern.

Product path: `RcclReducer` (ncclAllReduce inside libqmps_hip.so over xGMI; the unique id is
exchanged by whatever launcher plumbing the host has - bench.py uses torch.distributed/gloo).
(The CPU test-suite plugs a torch.distributed/gloo reducer with the same `allreduce_sum` method into
`ShardedCost` - tests/gloo_reducer.py - so the N > 1 control flow is testable without GPUs; the product
package itself never imports torch.)
"""
import numpy as np


def shard_bounds(B, rank, world):
    """Contiguous block of the batch axis owned by `rank`: sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError('bad rank/world')
    base, extra = divmod(int(B), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class RcclReducer:
    """Sum over ranks through the engine's RCCL communicator (qmps_comm_init must have run)."""

    def __init__(self, engine):
        self.engine = engine

    def allreduce_sum(self, values):
        return self.engine.allreduce_sum(values)


class ShardedCost:
    """cost[t] = sum over the GLOBAL batch of E[b, t], each rank evaluating only its shard.

    `evaluate_local(states_shard) -> E[n_local, n_terms]` is the rank-local hot path
    (EnergyEngine.energies in production)."""

    def __init__(self, rank, world, reducer, evaluate_local):
        self.rank, self.world = rank, world
        self.reducer = reducer
        self.evaluate_local = evaluate_local

    def __call__(self, states):
        lo, hi = shard_bounds(len(states), self.rank, self.world)
        E = np.asarray(self.evaluate_local(states[lo:hi]), dtype=np.float64)
        local = E.reshape(hi - lo, -1).sum(0) if hi > lo else np.zeros(max(E.shape[-1] if E.ndim > 1 else 1, 1))
        return self.reducer.allreduce_sum(local)
