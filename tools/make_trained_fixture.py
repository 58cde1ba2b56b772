#!/usr/bin/env python3
"""Runs the trained-like fixture's recipe (tests/trained_fixture.py) on the GPU, prints its loss curve and margin statistics and -- with
--write -- pins them in tests/golden/trained_like.json; --oracle-steps N also replays the first N optimizer steps on the float64 torch-CPU
oracle and records how far the fp32 GPU trajectory is from it.  --lr / --steps / --n-ave override the recipe (recipe search)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import trained_fixture as tf  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lr", type=float, default=0.0)
ap.add_argument("--steps", type=int, default=0)
ap.add_argument("--n-ave", type=int, default=0)
ap.add_argument("--warm-steps", type=int, default=-1)
ap.add_argument("--warm-factor", type=float, default=0.0)
ap.add_argument("--cool-steps", type=int, default=-1)
ap.add_argument("--cool-factor", type=float, default=0.0)
ap.add_argument("--oracle-steps", type=int, default=0)
ap.add_argument("--write", action="store_true")
args = ap.parse_args()
recipe = dict(tf.RECIPE)
if args.lr:
    recipe["lr"] = args.lr
if args.steps:
    recipe["steps"] = args.steps
if args.n_ave:
    recipe["n_ave"] = args.n_ave
if args.warm_steps >= 0:
    recipe["warm_steps"] = args.warm_steps
if args.warm_factor:
    recipe["warm_factor"] = args.warm_factor
if args.cool_steps >= 0:
    recipe["cool_steps"] = args.cool_steps
if args.cool_factor:
    recipe["cool_factor"] = args.cool_factor
t0 = time.time()
wts, frames, curve = tf.train_like(recipe, verbose=False)
print("recipe", recipe, "trained in %.1f s" % (time.time() - t0))
print("curve", [(s, round(l, 1)) for s, l in curve])
first, last = curve[0][1], curve[-1][1]
print("fused loss %.1f -> %.1f (x%.1f down)" % (first, last, first / max(last, 1e-9)))
# margin statistics of the trained net on its frames (exact fp32 kernels)
net = tf.build(wts, "fp32")
stats = []
with torch.no_grad():
    for x, m in frames:
        o = net.forward(torch.from_numpy(x).cuda())[-1].cpu().numpy()
        pred = o > 0
        inter, union = np.logical_and(pred, m > 0.5).sum(), np.logical_or(pred, m > 0.5).sum()
        stats.append({"iou_vs_label": float(inter / max(1, union)), "logit_std": float(o.std()), "frac_abs_logit_below_0p5": float((np.abs(o) < 0.5).mean()),
                      "frac_abs_logit_below_0p1": float((np.abs(o) < 0.1).mean())})
print("margins", json.dumps(stats[:3]))
out = {"recipe": recipe, "curve": curve, "loss_drop": first / max(last, 1e-9), "margins": stats}
if args.oracle_steps:
    from oracle import torch_ref
    from osvos_pytorch_amd.train_common import make_sgd  # noqa: F401
    p = torch_ref.as_leaf_params(tf.initial_weights(recipe), dtype=torch.float64)
    opt = torch.optim.SGD(torch_ref.sgd_groups(p, lr=recipe["lr"], mode="parent"), lr=recipe["lr"], momentum=0.9)
    base = [g["lr"] for g in opt.param_groups]
    ref_curve, it, steps = [], 0, 0
    acc = 0.0
    while steps < args.oracle_steps:
        tf.set_rate(opt, base, tf.rate_factor(recipe, steps, recipe["steps"]))
        x, m = frames[it % len(frames)]
        loss, parts = torch_ref.train_loss(p, torch.from_numpy(x).double(), torch.from_numpy(m).double(), mode="parent", epoch=0, n_epochs=240)
        acc += float(parts[-1].detach())
        (loss / recipe["n_ave"]).backward()
        it += 1
        if it % recipe["n_ave"] == 0:
            opt.step()
            opt.zero_grad()
            steps += 1
            ref_curve.append((steps, acc / recipe["n_ave"]))
            acc = 0.0
    gpu_w, _, gpu_curve = tf.train_like(recipe, steps=args.oracle_steps, record_every=1)
    rel = [abs(a[1] - b[1]) / abs(b[1]) for a, b in zip(gpu_curve, ref_curve)]
    print("float64 oracle over %d steps: max relative loss difference %.2e" % (args.oracle_steps, max(rel)))
    print("   gpu", [(s_, round(l, 1)) for s_, l in gpu_curve][:16])
    print("   f64", [(s_, round(l, 1)) for s_, l in ref_curve][:16])
    out["oracle"] = {"steps": args.oracle_steps, "max_rel_loss_diff": max(rel), "ref_curve": ref_curve}
if args.write:
    with open(os.path.join(REPO, "tests", "golden", "trained_like.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tests/golden/trained_like.json")
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
with open(os.path.join(REPO, "gpurun_out", "trained_like_%g_%d.json" % (recipe["lr"], recipe["steps"])), "w") as f:
    json.dump(out, f, indent=1)
