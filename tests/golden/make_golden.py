#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference modules from /root/reference.

Runs only in the build container (the reference is not present on the GPU box).  The fixtures
pin (a) the torch-CPU functional oracle, (b) the plain-C oracle and (c) the HIP path to what
kmaninis/OSVOS-PyTorch itself computes on the same seeded inputs.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Weights / frames are NOT stored (61 MB): they are regenerated from seeds by oracle/synth.py.
Stored per case: the calibrated head parameters, the five logit maps, the losses, and for every
parameter gradient its sum, L2 norm and 24 sampled entries (fp32 run and fp64 run).
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.path.insert(0, REPO)
from oracle import synth  # noqa: E402

# import the reference's own modules (needs its directory first on sys.path because it uses
# top-level package names `networks`, `layers`, `mypath` -- the same names our drop-in uses)
for m in ("networks", "networks.vgg_osvos", "layers", "layers.osvos_layers", "mypath", "util", "util.path_abstract"):
    sys.modules.pop(m, None)
sys.path.insert(0, REF)
import networks.vgg_osvos as ref_vo  # noqa: E402
import layers.osvos_layers as ref_layers  # noqa: E402
assert ref_vo.__file__.startswith(REF), ref_vo.__file__

CASES = [
    # name, N, H, W, frame seed
    ("c16x16", 1, 16, 16, 11),
    ("c37x53_n2", 2, 37, 53, 12),     # odd x odd, batch 2 -> partial pool windows, odd crops
    ("c48x64", 1, 48, 64, 13),
    ("c30x85", 1, 30, 85, 14),
]
N_SAMPLES = 24


def sample_index(shape, key):
    rng = np.random.default_rng(abs(hash_str(key)) % (2 ** 32))
    n = int(np.prod(shape))
    return rng.integers(0, n, size=min(N_SAMPLES, n))


def hash_str(s):
    h = 1469598103934665603
    for ch in s.encode():
        h = ((h ^ ch) * 1099511628211) % (2 ** 64)
    return h


def build_ref_net(weights, dtype):
    net = ref_vo.OSVOS(pretrained=0)
    sd = OrderedDict((k, torch.from_numpy(v.copy())) for k, v in weights.items())
    net.load_state_dict(sd)
    return net.to(dtype)


def ref_forward_fn(weights, x):
    net = build_ref_net(weights, torch.float32)
    with torch.no_grad():
        outs = net.forward(torch.from_numpy(x))
    return [o.numpy() for o in outs]


def grads_summary(named, out, prefix):
    for k, g in named:
        g = g.detach().double().numpy().ravel()
        idx = sample_index(g.shape, k)
        out[prefix + k + "|sum"] = np.float64(g.sum())
        out[prefix + k + "|l2"] = np.float64(np.sqrt((g * g).sum()))
        out[prefix + k + "|idx"] = idx
        out[prefix + k + "|val"] = g[idx]


def run_case(name, n, h, w, seed):
    x = synth.make_frame(n, h, w, seed)
    m = synth.make_mask(n, h, w, seed)
    wts = synth.calibrate_heads(synth.make_weights(1), ref_forward_fn, x)
    out = {"meta": np.array([n, h, w, seed, 1], np.int64)}
    for k in wts:
        if k.startswith("score_dsn") or k.startswith("fuse"):
            out["head|" + k] = wts[k]
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for mode in ("online", "parent"):
            net = build_ref_net(wts, dt)
            xin = torch.from_numpy(x).to(dt)
            xin.requires_grad_()
            gt = torch.from_numpy(m).to(dt)
            outs = net.forward(xin)
            if mode == "online":     # train_online.py:127,140-141
                loss = ref_layers.class_balanced_cross_entropy_loss(outs[-1], gt, size_average=False)
                heads = [loss]
            else:                    # train_parent.py:143-147, epoch 60 of 240
                heads = [ref_layers.class_balanced_cross_entropy_loss(o, gt, size_average=False) for o in outs]
                loss = (1 - 60 / 240) * sum(heads[:-1]) + heads[-1]
            (loss / 5).backward()
            p = "%s|%s|" % (tag, mode)
            if mode == "parent":
                for i, o in enumerate(outs):
                    out["%s|out%d" % (tag, i)] = o.detach().numpy()
            out[p + "loss"] = np.float64(loss.item())
            out[p + "heads"] = np.array([hh.item() for hh in heads], np.float64)
            named = [(k, v.grad) for k, v in net.named_parameters() if v.grad is not None]
            grads_summary(named, out, p + "grad|")
            grads_summary([("input", xin.grad)], out, p + "grad|")
    # SGD trajectory: 2 optimizer steps of the online loop with nAveGrad=2 (train_online.py:79-88,112-149)
    net = build_ref_net(wts, torch.float32)
    lr, wd = 1e-8, 0.0002
    opt = torch.optim.SGD([
        {'params': [pr[1] for pr in net.stages.named_parameters() if 'weight' in pr[0]], 'weight_decay': wd},
        {'params': [pr[1] for pr in net.stages.named_parameters() if 'bias' in pr[0]], 'lr': lr * 2},
        {'params': [pr[1] for pr in net.side_prep.named_parameters() if 'weight' in pr[0]], 'weight_decay': wd},
        {'params': [pr[1] for pr in net.side_prep.named_parameters() if 'bias' in pr[0]], 'lr': lr * 2},
        {'params': [pr[1] for pr in net.upscale.named_parameters() if 'weight' in pr[0]], 'lr': 0},
        {'params': [pr[1] for pr in net.upscale_.named_parameters() if 'weight' in pr[0]], 'lr': 0},
        {'params': net.fuse.weight, 'lr': lr / 100, 'weight_decay': wd},
        {'params': net.fuse.bias, 'lr': 2 * lr / 100},
    ], lr=lr, momentum=0.9)
    w0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    losses = []
    for it in range(4):
        xin = torch.from_numpy(x)
        xin.requires_grad_()
        outs = net.forward(xin)
        loss = ref_layers.class_balanced_cross_entropy_loss(outs[-1], torch.from_numpy(m), size_average=False)
        losses.append(loss.item())
        loss = loss / 2
        loss.backward()
        if it % 2 == 1:
            opt.step()
            opt.zero_grad()
    out["sgd|losses"] = np.array(losses, np.float64)
    delta = [(k, net.state_dict()[k] - w0[k]) for k in w0]
    grads_summary(delta, out, "sgd|delta|")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: float(out[k]) for k in out if k.endswith("|loss")})


def helpers_fixture():
    out = {}
    for k in (3, 4, 5, 8, 16, 32):
        out["filt|%d" % k] = ref_layers.upsample_filt(k)
    for (hin, win, ht, wt) in [(482, 856, 480, 854), (496, 880, 480, 854), (13, 9, 8, 4), (10, 11, 5, 6), (7, 7, 7, 7)]:
        t = torch.arange(hin * win, dtype=torch.float32).reshape(1, 1, hin, win)
        c = ref_layers.center_crop(t, ht, wt)
        out["crop|%d_%d_%d_%d" % (hin, win, ht, wt)] = np.array([c[0, 0, 0, 0].item() // win, c[0, 0, 0, 0].item() % win, c.shape[2], c.shape[3]], np.int64)
    crops_fixture(out)
    rng = np.random.default_rng(7)
    logits = (rng.standard_normal((2, 1, 9, 11)) * 4).astype(np.float32)
    lab = (rng.random((2, 1, 9, 11)) > 0.7).astype(np.float32)
    soft = rng.random((2, 1, 9, 11)).astype(np.float32)
    out["loss|logits"], out["loss|lab"], out["loss|soft"] = logits, lab, soft
    for tag, label in (("bin", lab), ("soft", soft), ("allneg", np.zeros_like(lab)), ("allpos", np.ones_like(lab))):
        for sa, ba in ((True, True), (False, True), (False, False)):
            xin = torch.from_numpy(logits).clone().requires_grad_()
            l = ref_layers.class_balanced_cross_entropy_loss(xin, torch.from_numpy(label), size_average=sa, batch_average=ba)
            l.backward()
            out["loss|%s|%d%d|val" % (tag, sa, ba)] = np.float64(l.item())
            out["loss|%s|%d%d|grad" % (tag, sa, ba)] = xin.grad.numpy()
    out["logit|in"] = np.array([0.01, 0.3, 0.5, 0.99])
    out["logit|out"] = ref_layers.logit(out["logit|in"])
    out["sigmoid|out"] = ref_layers.sigmoid_np(np.array([-3.0, 0.0, 2.5]))
    # state_dict layout (key order + shapes) of the reference module
    net = ref_vo.OSVOS(pretrained=0)
    sd = net.state_dict()
    out["sd|keys"] = np.array(list(sd.keys()))
    out["sd|shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)
    print("wrote helpers")


def adapters_fixture():
    """The reference's two weight-import paths (vgg_osvos.py:92-125) run on seeded synthetic files of the right layout
    (the real vgg_pytorch.pth / vgg_caffe.mat are absent offline): checksums of the 26 trunk tensors the reference ends up
    with, their contiguity, plus what a reference-saved state_dict of the Caffe-initialised net looks like."""
    import tempfile
    out = {"meta": np.array([5, 6], np.int64)}         # seeds of the .pth and the .mat
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "models"))
        synth.write_vgg_pytorch_pth(os.path.join(tmp, "models", "vgg_pytorch.pth"), 5)
        synth.write_vgg_caffe_mat(os.path.join(tmp, "models", "vgg_caffe.mat"), 6)
        os.chdir(tmp)                                   # the reference's Path.models_dir() is "./models"
        try:
            for pretrained, tag in ((1, "pth"), (2, "mat")):
                torch.manual_seed(0)
                net = ref_vo.OSVOS(pretrained=pretrained)
                named = [(k, v) for k, v in net.state_dict().items() if k.startswith("stages.")]
                assert len(named) == 26
                grads_summary(named, out, tag + "|")
                out[tag + "|contiguous"] = np.array([bool(v.is_contiguous()) for _, v in named])
                # the heads keep the default init (normal(0, 0.001) / bilinear): pin two of them as well
                grads_summary([(k, v) for k, v in net.state_dict().items() if k in ("upscale.2.weight", "fuse.bias")], out, tag + "|")
                if pretrained == 2:
                    # a checkpoint the reference writes from this net (train_parent.py:175-176) keeps the transposed strides
                    torch.save(net.state_dict(), os.path.join(tmp, "ref_saved.pth"))
                    back = torch.load(os.path.join(tmp, "ref_saved.pth"))
                    out["mat|saved_contiguous"] = np.array([bool(back[k].is_contiguous()) for k, _ in named])
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "adapters.npz"), **out)
    print("wrote adapters", {k: out[k].tolist() for k in out if k.endswith("contiguous")})


def crops_fixture(out):
    """center_crop with target > input (zero padding) and mixed crop/pad (osvos_layers.py:51-56): full result arrays."""
    for (hin, win, ht, wt) in [(5, 8, 10, 13), (5, 13, 10, 8), (7, 7, 12, 2), (4, 4, 9, 9), (9, 6, 4, 11)]:
        t = torch.arange(1, hin * win + 1, dtype=torch.float32).reshape(1, 1, hin, win)
        out["cropfull|%d_%d_%d_%d" % (hin, win, ht, wt)] = ref_layers.center_crop(t, ht, wt).numpy()


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                              # python make_golden.py helpers adapters  -> only those fixtures
        for what in sys.argv[1:]:
            {"helpers": helpers_fixture, "adapters": adapters_fixture}[what]()
        sys.exit(0)
    helpers_fixture()
    adapters_fixture()
    for c in CASES:
        run_case(*c)
