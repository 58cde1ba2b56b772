#!/usr/bin/env python3
"""bf16 ERROR BUDGET of the OSVOS forward: which layers spend the parity budget of configs[2], and what would a mixed-precision
policy that clears SURVEY 8(d)'s flat bf16 bars (logits <= 0.1 std, loss <= 2e-3, IoU >= 1 - 1e-3 where the margins are real) cost?

CHECKER-SIDE analysis (CPU only, torch fp32; like tests/, it may use oracle/): the HIP bf16 path is EMULATED operand-exactly -- every 3x3
convolution takes RNE-bf16-rounded activations and weights, multiplies exactly (a product of two bf16 values is exact in fp32) and accumulates in
fp32, bias + ReLU in fp32, the stored activation is rounded to bf16 again (csrc/conv3x3_bf16.hip, bf16-store mode); pooling on bf16 values
(max commutes with rounding); side_prep output, head and loss fp32.  Only the fp32 summation ORDER differs from the kernels (1e-6-level).  The
emulator's all-bf16 row must therefore land on the GPU path's own gate numbers (bench.py configs[2] parity: 0.127 std / 5.7e-3 / 0.9884) --
that is its validation, printed first.

Per-layer operand policies:
  b   bf16 x bf16                              1 MFMA product per algorithmic product (what configs[2] runs)
  w2  bf16 activation x (w_hi + w_lo)          2 products: removes the weight-rounding half of the error
  x3  (a_hi + a_lo) x (w_hi + w_lo) - a_lo w_lo 3 products: ~2^-16 relative per term ("bf16x3")
  f   fp32 x fp32                              what the f32x3 kernels deliver (6 products)
A policy string names one letter per trunk stage 0..4 plus one for the four side_prep convolutions, e.g. "bbbbb/b", "bbbff/f".

usage: bf16_budget.py [--n 2] [--height 480 --width 854] [--fixture synthetic|trained] [--policies p1,p2,...]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import torch_ref  # noqa: E402


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def conv_policy(x, w, b, pol):
    """x: fp32 tensor holding the values the producer STORED (bf16-representable when the producer was a bf16-store layer)."""
    if pol == "f":
        return F.conv2d(x, w, b, padding=1)
    xh, wh = rb(x), rb(w)
    y = F.conv2d(xh, wh, b, padding=1)
    if pol == "b":
        return y
    wl = rb(w - wh)
    if pol == "w2":
        return y + F.conv2d(xh, wl, None, padding=1)
    if pol == "x3":
        xl = rb(x - xh)
        return y + F.conv2d(xh, wl, None, padding=1) + F.conv2d(xl, wh, None, padding=1)
    raise ValueError(pol)


def forward_policy(params, x, policy):
    """vgg_osvos.py:59-74 with per-stage operand policies; a 'b' / 'w2' stage stores its activations in bf16 (the bf16-store mode), an 'x3' /
    'f' stage in fp32."""
    trunk, side = policy.split("/")
    H, W = int(x.shape[-2]), int(x.shape[-1])
    names = torch_ref.trunk_conv_names()
    sides, side_out = [], []
    for si in range(5):
        pol = trunk[si]
        if si > 0:
            x = F.max_pool2d(x, kernel_size=2, stride=2, ceil_mode=True)
        for n in names[si]:
            x = F.relu(conv_policy(x, params[n + ".weight"], params[n + ".bias"], pol))
            if pol in ("b", "w2"):
                x = rb(x)
        if si > 0:
            i, s = si - 1, 2 ** si
            prep = conv_policy(x, params["side_prep.%d.weight" % i], params["side_prep.%d.bias" % i], side)
            up = F.conv_transpose2d(prep, params["upscale.%d.weight" % i], stride=s)
            sides.append(torch_ref.crop_to(up, H, W))
            score = F.conv2d(prep, params["score_dsn.%d.weight" % i], params["score_dsn.%d.bias" % i])
            side_out.append(torch_ref.crop_to(F.conv_transpose2d(score, params["upscale_.%d.weight" % i], stride=s), H, W))
    fused = F.conv2d(torch.cat(sides, dim=1), params["fuse.weight"], params["fuse.bias"])
    return side_out + [fused]


# forward GFLOP per 854x480 frame and stage: trunk stages 0..4, then the four side_prep convolutions together
STAGE_GFLOP = [1.417 + 30.223, 15.111 + 30.223, 15.147 + 2 * 30.293, 15.147 + 2 * 30.293, 3 * 7.644]
SIDE_GFLOP = 3.778 + 1.893 + 0.947 + 0.239
PRODUCTS = {"b": 1, "w2": 2, "x3": 3, "f": 6}


# COST MODEL: executed MFMA products of the FORWARD relative to all-bf16 (the backward stays bf16 in every policy: SURVEY's gradient bar, 0.25
# rel-L2, is met with room); step-level estimate = 1 + 0.30 (fwd - 1): forward convolutions are ~30 % of the configs[2] step
# (profiles/r05_step_timeline_bf16_b12.txt), ignoring that an x3 / f stage also moves fp32 activations (so the estimate is a LOWER bound).


def parse(policy):
    """'bbbbb/b' with single letters; 'w' = w2, 'x' = x3."""
    m = {"b": "b", "w": "w2", "x": "x3", "f": "f"}
    t, s = policy.split("/")
    return [m[c] for c in t], m[s]


def run_policy(params, x, gt, ref, policy):
    t, s = parse(policy)

    class P:
        def split(self, _):
            return t, s
    outs = forward_policy(params, x, P())
    res = {"policy": policy}
    dl, rms, lrel = [], [], []
    for i in range(5):
        d = (outs[i].double() - ref["outs"][i])
        sd = float(ref["outs"][i].std())
        dl.append(float(d.abs().max()) / sd)
        rms.append(float(d.pow(2).mean().sqrt()) / sd)
        l = float(torch_ref.cbce_loss(outs[i], gt, size_average=False))
        lrel.append(abs(l - ref["loss"][i]) / abs(ref["loss"][i]))
    g, r = outs[4] > 0, ref["outs"][4] > 0
    union = int((g | r).sum())
    res.update({"max_dlogit_over_std": [float("%.3g" % v) for v in dl], "rms_dlogit_over_std": [float("%.3g" % v) for v in rms],
                "loss_rel": [float("%.2g" % v) for v in lrel], "iou": round(float((g & r).sum()) / max(1, union), 6),
                "flipped": int((g != r).sum()), "pixels": int(g.numel())})
    ex = sum(gf * PRODUCTS[t[i]] for i, gf in enumerate(STAGE_GFLOP)) + SIDE_GFLOP * PRODUCTS[s]
    fwd = ex / (sum(STAGE_GFLOP) + SIDE_GFLOP)
    res["fwd_products_x"] = round(fwd, 3)
    res["step_cost_estimate_x"] = round(1.0 + 0.30 * (fwd - 1.0), 3)
    res["within_flat_bars"] = bool(max(dl) <= 0.1 and max(lrel) <= 2e-3)       # SURVEY 8(d) logits + loss; the IoU column is judged separately
    res["iou_within_1e-3"] = bool(res["iou"] >= 1 - 1e-3)
    return res


DEFAULT_POLICIES = ["bbbbb/b",
                    "fbbbb/b", "bfbbb/b", "bbfbb/b", "bbbfb/b", "bbbbf/b", "bbbbb/f",            # ONE stage exact at a time: who spends the budget
                    "bbbbx/x", "bbbxx/x", "bbxxx/x", "bxxxx/x", "xxxxx/x",                       # deepest-first 3-product policies
                    "xbbbb/b", "xxbbb/b", "xxxbb/b",                                             # shallow-first
                    "wwwww/w", "bbbww/w", "bbwww/w",                                             # weight split only (2 products)
                    "fffff/f"]


def synthetic_problem(n, h, w, seed=0):
    """bench.py's own problem (synth_problem), built on CPU through the oracle's modules: He-init weights, heads calibrated to N(-1, 3^2)."""
    from oracle import synth
    wts, x, m = synth.calibrated_problem(n, h, w, seed=seed)
    return {k: torch.from_numpy(v) for k, v in wts.items()}, torch.from_numpy(x), torch.from_numpy(m)


def trained_problems():
    """tests/trained_fixture.py: the trained-like net (needs the GPU: ~10 s of parent training on the exact fp32 kernels) and the four cases
    tests/test_gpu_trained_like.py checks.  The POLICIES are still emulated on the host CPU."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import trained_fixture as tf
    from oracle import synth
    wts, frames, _ = tf.train_like()
    h, w = tf.RECIPE["h"], tf.RECIPE["w"]
    cases = [("train0", frames[0][0], frames[0][1]), ("train3", frames[3][0], frames[3][1]),
             ("heldout",) + synth.trainable_frame(1, h, w, seed=tf.RECIPE["frame_seed"] + 99),
             ("heldout_240x427",) + synth.trainable_frame(1, 240, 427, seed=tf.RECIPE["frame_seed"] + 98)]
    params = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in wts.items()}
    return params, [(n, torch.from_numpy(x), torch.from_numpy(m)) for n, x, m in cases]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--fixture", default="synthetic")
    ap.add_argument("--case", default="heldout")
    ap.add_argument("--policies", default=",".join(DEFAULT_POLICIES))
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    if args.fixture == "synthetic":
        params, x, gt = synthetic_problem(args.n, args.height, args.width)
        problems = [("synthetic He-init net, calibrated heads, %dx%d N=%d (configs[2]'s kind of problem)" % (args.width, args.height, args.n), x, gt)]
    else:
        params, cases = trained_problems()
        problems = [("trained-like fixture (tests/trained_fixture.py), case %s %s" % (n, tuple(x.shape)), x, gt) for n, x, gt in cases]
    all_rows = {}
    for label, x, gt in problems:
        all_rows[label] = one_problem(params, x, gt, label, args.policies.split(","))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(all_rows, f, indent=1)


def one_problem(params, x, gt, label, policies):
    t0 = time.time()
    with torch.no_grad():
        p64 = {k: v.double() for k, v in params.items()}
        r = torch_ref.forward(p64, x.double())
        ref = {"outs": r, "loss": [float(torch_ref.cbce_loss(o, gt.double(), size_average=False)) for o in r]}
        rows = []
        print("# %s; truth = the oracle in float64 (%.0f s)" % (label, time.time() - t0), flush=True)
        print("# %-10s %-38s %-38s %-44s %-9s %-8s %-6s %-6s %s" % ("policy", "max|dlogit|/std per head", "rms/std per head", "loss rel per head", "IoU", "flipped",
                                                                 "fwd x", "step x", "flat bars"), flush=True)
        for pol in policies:
            res = run_policy(params, x, gt, ref, pol)
            rows.append(res)
            print("  %-10s %-38s %-38s %-44s %-9.6f %-8d %-6.2f %-6.3f %s" % (
                pol, " ".join("%.3f" % v for v in res["max_dlogit_over_std"]), " ".join("%.4f" % v for v in res["rms_dlogit_over_std"]),
                " ".join("%.1e" % v for v in res["loss_rel"]), res["iou"], res["flipped"], res["fwd_products_x"], res["step_cost_estimate_x"],
                ("OK" if res["within_flat_bars"] else "-") + (" +IoU" if res["iou_within_1e-3"] else "")), flush=True)
    return rows


if __name__ == "__main__":
    main()
