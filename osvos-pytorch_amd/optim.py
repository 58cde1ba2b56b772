"""Fused multi-tensor SGD: every parameter tensor of the network updated by ONE kernel launch.

Drop-in for the optimizer the reference builds (train_online.py:79-88, train_parent.py:87-103):
``optim.SGD([{'params': ..., 'weight_decay': wd, 'lr': lr}, ...], lr=lr, momentum=0.9)`` -- same constructor
arguments, same ``param_groups`` / ``state`` layout (``state[p]['momentum_buffer']``), so ``state_dict()`` is
interchangeable with ``torch.optim.SGD``'s (checkpoint a run with one, resume with the other).  Update rule
(torch.optim.SGD, dampening 0, no Nesterov):  d = g + wd*p;  buf = d (first step) | momentum*buf + d;  p -= lr*buf.

CUDA (ROCm) float32 parameters only; there is no CPU fallback.
"""
import ctypes as C

import torch

from ._lib import check, lib


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if dampening != 0.0 or nesterov:
            raise ValueError("FusedSGD implements the reference's configuration only: dampening=0, nesterov=False")
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("FusedSGD: negative lr / momentum / weight_decay")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay, nesterov=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        buckets = {}          # (momentum, first) -> list of (p, g, buf, lr, wd)
        for group in self.param_groups:
            lr, wd, mom = float(group["lr"]), float(group["weight_decay"]), float(group["momentum"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("FusedSGD needs contiguous float32 CUDA (ROCm) parameters; there is no CPU fallback")
                if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                    raise RuntimeError("FusedSGD: gradient of a parameter is not a contiguous float32 tensor on its device")
                st = self.state[p]
                first = "momentum_buffer" not in st or st["momentum_buffer"] is None
                if mom == 0.0:
                    # torch keeps no buffer without momentum; the kernel still needs somewhere to write d
                    buf = st.get("_scratch")
                    if buf is None:
                        buf = st["_scratch"] = torch.empty_like(p)
                    first = True
                elif first:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                    buf = st["momentum_buffer"]
                else:
                    buf = st["momentum_buffer"]
                buckets.setdefault((mom, first), []).append((p, g, buf, lr, wd))
        if not buckets:
            return loss
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        vp = C.c_void_p
        for (mom, first), items in buckets.items():
            n = len(items)
            P = (vp * n)(*[vp(t[0].data_ptr()) for t in items])
            G = (vp * n)(*[vp(t[1].data_ptr()) for t in items])
            B = (vp * n)(*[vp(t[2].data_ptr()) for t in items])
            cnt = (C.c_long * n)(*[t[0].numel() for t in items])
            lrs = (C.c_float * n)(*[t[3] for t in items])
            wds = (C.c_float * n)(*[t[4] for t in items])
            check(lib().osvos_sgd_step_multi(P, G, B, cnt, lrs, wds, n, mom, int(first), stream), "sgd_step_multi")
            # the kernel wrote through raw pointers: tell autograd (and the packed-weight cache keyed on _version)
            torch.autograd.graph.increment_version([t[0] for t in items])
        return loss

    def state_dict(self):
        sd = super().state_dict()
        for st in sd["state"].values():        # scratch of momentum-free groups is not state
            st.pop("_scratch", None)
        return sd
