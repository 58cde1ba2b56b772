#!/usr/bin/env python3
"""Debug: gradients of the 2 x 37 x 53 problem under combinations of OSVOS_MASK_BITS / OSVOS_FUSE_POOL / OSVOS_X3_KSPLIT (one subprocess each)."""
import os, subprocess, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = '''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_net as T
from oracle import synth
from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
res = {}
wts, x, m = synth.calibrated_problem(2, 37, 53, seed=9)
net = T.build_net(wts, "fp32x3")
xg = torch.from_numpy(x).requires_grad_()
outs = net.forward(xg.cuda())
gt = torch.from_numpy(m).cuda()
losses = [cbce(o, gt, size_average=False) for o in outs]
(0.5 * sum(losses[:-1]) + losses[-1]).backward()
for k, v in net.named_parameters():
    if v.grad is not None:
        res["g:" + k] = v.grad.cpu().numpy()
res["dx"] = xg.grad.numpy()
np.savez(sys.argv[1], **res)
''' % (os.path.join(REPO, "tests"), REPO)
variants = {"bits0": dict(OSVOS_MASK_BITS="0", OSVOS_FUSE_POOL="0"), "bits1": dict(OSVOS_MASK_BITS="1", OSVOS_FUSE_POOL="0"),
            "bits1_fuse1": dict(OSVOS_MASK_BITS="1", OSVOS_FUSE_POOL="1"), "bits1_fwdonly": dict(OSVOS_MASK_BITS="1", OSVOS_FUSE_POOL_FWD="1", OSVOS_FUSE_POOL_BWD="0"),
            "bits1_ks1": dict(OSVOS_MASK_BITS="1", OSVOS_FUSE_POOL="0", OSVOS_X3_KSPLIT="1")}
got = {}
tmp = tempfile.mkdtemp()
for tag, env in variants.items():
    out = os.path.join(tmp, tag + ".npz")
    subprocess.run([sys.executable, "-c", CODE, out], check=True, env=dict(os.environ, **env), stdout=subprocess.DEVNULL)
    got[tag] = dict(np.load(out))
ref = got["bits0"]
for tag in variants:
    if tag == "bits0":
        continue
    worst = sorted(((float(np.linalg.norm(got[tag][k] - ref[k]) / (np.linalg.norm(ref[k]) + 1e-30)), k) for k in ref), reverse=True)[:4]
    print(tag, "vs bits0:", ["%.2e %s" % w for w in worst])
