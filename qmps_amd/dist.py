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

    def allreduce_min(self, values):
        return self.engine.allreduce_min(values)


def reduce_sweep_costs(hist, reducer, chunk=16):
    """Exchange step of a sharded optimiser run (BASELINE.json configs[3]: restarts sharded over the GPUs, RCCL all-reduce):
    hist (n_sweeps, R_local) = the energies of this rank's restarts after every sweep.  Returns (summed cost per sweep over ALL
    ranks' restarts (n_sweeps,), restarts counted, best final energy over all ranks).  Rows of NaN-free values only count;
    sums travel `chunk` (<= 16) doubles per all-reduce (the library's small-vector all-reduce), the best cost as one ncclMin."""
    hist = np.asarray(hist, dtype=np.float64).reshape(len(hist), -1)
    ok = np.isfinite(hist).all(axis=0) if hist.shape[1] else np.zeros(0, bool)
    local = np.concatenate([hist[:, ok].sum(axis=1), [float(ok.sum())]])
    total = np.concatenate([reducer.allreduce_sum(local[k:k + chunk]) for k in range(0, len(local), chunk)])
    best = reducer.allreduce_min(np.array([hist[-1, ok].min() if ok.any() else np.inf]))[0]
    return total[:-1], int(round(total[-1])), float(best)


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
