"""bench.py's parity gate, plumbing only (CPU tier): a stand-in 'timed net' that IS the oracle must come out at zero error and within the bars,
and a perturbed one must fail them -- the real gate (HIP path vs oracle) runs on the GPU box inside bench.py."""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _OracleNet(object):
    """Quacks like the product OSVOS module for the calls parity_gate makes; runs oracle/torch_ref.py."""

    def __init__(self, wts, noise=0.0):
        from oracle import torch_ref
        self.tr, self.noise = torch_ref, noise
        self.p = torch_ref.as_leaf_params(wts)

    def state_dict(self):
        return OrderedDict((k, v.detach()) for k, v in self.p.items())

    def named_parameters(self):
        return list(self.p.items())

    def forward(self, x):
        outs = self.tr.forward(self.p, x)
        return [o + self.noise * torch.sin(17.0 * o) for o in outs] if self.noise else outs

    def join_backward(self):
        pass


class _Opt(object):
    def __init__(self, net):
        self.net = net

    def zero_grad(self):
        for _, v in self.net.named_parameters():
            v.grad = None


def _workload(mode, noise):
    import bench
    from oracle import synth
    n, h, w = (1, 33, 47) if mode == "online" else (2, 33, 47)
    wts = synth.calibrate_heads(synth.make_weights(1), synth.torch_forward_fn(), synth.make_frame(n, h, w, seed=3))
    wl = bench.Workload.__new__(bench.Workload)
    wl.net = _OracleNet(wts, noise)
    wl.opt = _Opt(wl.net)
    wl.mode, wl.precision, wl.h, wl.w, wl.batch, wl.n_ave, wl.epoch = mode, "fp32x3", h, w, n, 5, 0
    wl.x, wl.gt = torch.from_numpy(synth.make_frame(n, h, w, seed=3)), torch.from_numpy(synth.make_mask(n, h, w, seed=3))
    wl.running = torch.zeros(())
    wl.fused_loss, wl.item_sync, wl.reducer = False, 0, None
    from oracle.torch_ref import cbce_loss
    wl.cbce = cbce_loss
    return wl


def test_gate_is_exact_for_the_oracle_itself_and_trips_on_a_perturbed_net():
    import bench
    for mode in ("online", "parent"):
        r = bench.parity_gate(_workload(mode, 0.0))
        assert r["within_bars"] and r["max_dlogit_over_std"] == 0.0 and r["loss_rel"] == 0.0 and r["iou"] == 1.0, r
        assert r["grad_rel_l2_worst"]["value"] == 0.0 and set(r["grad_rel_l2"]) == {"stages.0.0.weight", "stages.2.1.weight", "fuse.weight"}
        assert len(r["loss"]) == (1 if mode == "online" else 5)
    bad = bench.parity_gate(_workload("online", 0.02))
    assert not bad["within_bars"] and bad["max_dlogit_over_std"] > 1e-3, bad


def test_gate_forward_only_mode():
    import bench
    wl = _workload("online", 0.0)
    wl.mode = "infer"
    r = bench.parity_gate(wl)
    assert r["within_bars"] and "loss_rel" not in r and set(r["bars"]) == {"max_dlogit_over_std", "iou"}
