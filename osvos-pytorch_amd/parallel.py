"""Data-parallel gradient exchange: one process per GPU, torch.distributed ('nccl' = RCCL over
xGMI on ROCm; 'gloo' on CPU for the tests).  The reference has no distributed code at all
(single device, gradient *accumulation* only: train_online.py:140-149, train_parent.py:163-172);
this is the one exchange step the data-parallel version of those loops needs.

Each rank runs its own micro-batches and accumulates locally; once per optimizer step the flat
gradient buffer (15.27 M fp32 = 61 MB, frozen deconv weights carry no gradient) is summed across
ranks with ONE all-reduce -- a single large collective, sized for the per-link-bound xGMI ring
rather than many small buckets -- and every rank applies the identical SGD update.

With the reference's per-frame class-balance weights (osvos_layers.py:30-32) W ranks x (nAveGrad/W)
micro-batches reproduce the single-process gradient exactly (up to summation order) whenever W
divides nAveGrad: use average=False and keep `loss /= nAveGrad`.  average=True divides the sum by
the world size (weak scaling: every rank keeps nAveGrad local micro-batches)."""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradientAllReducer:
    def __init__(self, module, average=False, process_group=None, always=False):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.average = average
        self.group = process_group
        self.always = always          # run the collective even in a 1-rank group (exercises RCCL on one GPU)
        self._flat = None

    def all_reduce(self):
        """Sum (or average) the .grad of every parameter that has one across all ranks."""
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not self.always):
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        total = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != total or self._flat.device != grads[0].device:
            self._flat = torch.empty(total, device=grads[0].device, dtype=grads[0].dtype)
        views, off = [], 0
        for g in grads:
            views.append(self._flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(views, grads)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            self._flat.div_(dist.get_world_size(self.group))
        torch._foreach_copy_(grads, views)

    def broadcast_parameters(self, src=0):
        """Make every rank start from rank `src`'s weights."""
        if not dist.is_initialized():
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)


def shard_indices(n_items, rank, world_size):
    """Frames / sequences of a job handled by `rank`: r, r+W, r+2W, ...."""
    return list(range(rank, n_items, world_size))
