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
    """Gradients live in ONE flat buffer: after the first backward every ``p.grad`` is re-pointed to a view into it
    (``attach``), the network's backward accumulates into those views in place, the collective runs on the flat buffer
    itself and ``zero_grads`` is one memset -- no flatten / unflatten copies (2 x 61 MB per optimizer step before) and no
    per-tensor allocations.  If somebody replaces a ``.grad`` (``optimizer.zero_grad()`` with set_to_none) the next call
    copies it back in and re-attaches."""

    def __init__(self, module, average=False, process_group=None, always=False):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.average = average
        self.group = process_group
        self.always = always          # run the collective even in a 1-rank group (exercises RCCL on one GPU)
        self._flat = None
        self._views = {}              # id(param) -> view into _flat

    def attach(self):
        """Point the .grad of every parameter that has one at its slice of the flat buffer.  Returns the flat buffer
        (None when no parameter has a gradient yet)."""
        have = [p for p in self.params if p.grad is not None]
        if not have:
            return None
        total = sum(p.grad.numel() for p in have)
        layout_ok = (self._flat is not None and self._flat.numel() == total and self._flat.device == have[0].grad.device
                     and len(self._views) == len(have) and all(id(p) in self._views for p in have))
        if not layout_ok:
            self._flat = torch.empty(total, device=have[0].grad.device, dtype=have[0].grad.dtype)
            self._views, off = {}, 0
            for p in have:
                n = p.grad.numel()
                self._views[id(p)] = self._flat[off:off + n].view_as(p.grad)
                off += n
        stale = [p for p in have if p.grad.data_ptr() != self._views[id(p)].data_ptr()]
        if stale:
            torch._foreach_copy_([self._views[id(p)] for p in stale], [p.grad for p in stale])
            for p in stale:
                p.grad = self._views[id(p)]
        return self._flat

    def zero_grads(self):
        """Replacement for ``optimizer.zero_grad()``: keeps the views alive, one memset.  Falls back to setting the
        gradients to None before the first ``attach``."""
        if self._flat is None:
            for p in self.params:
                p.grad = None
            return
        self._flat.zero_()
        for p in self.params:
            v = self._views.get(id(p))
            if v is not None:
                p.grad = v

    def all_reduce(self):
        """Sum (or average) the .grad of every parameter that has one across all ranks."""
        flat = self.attach()
        if flat is None:
            return
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not self.always):
            return
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            flat.div_(dist.get_world_size(self.group))

    def broadcast_parameters(self, src=0):
        """Make every rank start from rank `src`'s weights."""
        if not dist.is_initialized():
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)


def shard_indices(n_items, rank, world_size):
    """Frames / sequences of a job handled by `rank`: r, r+W, r+2W, ...."""
    return list(range(rank, n_items, world_size))
