"""CPU test plumbing: the reducer interface of qmps_amd.dist (allreduce_sum) over a torch.distributed gloo group.
Lives under tests/ - the product package never imports torch."""
import numpy as np


class GlooReducer:
    """Same reduction over a torch.distributed (gloo) process group - CPU test plumbing."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self.group = torch, dist, group

    def allreduce_sum(self, values):
        t = self._torch.tensor(np.asarray(values, dtype=np.float64))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.numpy().copy()

    def allreduce_min(self, values):
        t = self._torch.tensor(np.asarray(values, dtype=np.float64))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group)
        return t.numpy().copy()
