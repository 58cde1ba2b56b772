"""ORACLE (test infrastructure only) -- functional restatement of the OSVOS hot path on torch CPU.

This file is a *checker*, not a product path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  The shipped HIP path never routes through it.

It restates, as pure functions over a flat ``{state_dict key: tensor}`` mapping, what the
reference expresses as an ``nn.Module`` tree:

* forward wiring ............ reference networks/vgg_osvos.py:59-74
* trunk stage layout ........ reference networks/vgg_osvos.py:19-24,36,136-145
* side / score / deconv ..... reference networks/vgg_osvos.py:41-46,54
* center crop ............... reference layers/osvos_layers.py:51-56
* bilinear filter ........... reference layers/osvos_layers.py:59-67
* class-balanced BCE ........ reference layers/osvos_layers.py:19-48
* SGD parameter groups ...... reference train_online.py:79-88, train_parent.py:87-103

The arithmetic itself lives in PyTorch/ATen (an unpinned third-party dependency of the
reference, README.md:21), so running these ``torch.nn.functional`` calls on the CPU *is* the
reference CPU path.  Parity pin: ``tests/golden/*.npz`` were produced by importing the real
reference modules from /root/reference (tests/golden/make_golden.py) and this restatement is
checked against them in tests/test_oracle_golden.py.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# channel plan of the five trunk stages (vgg_osvos.py:19-24)
STAGE_CHANNELS = [[64, 64], [128, 128], [256, 256, 256], [512, 512, 512], [512, 512, 512]]
STAGE_IN = [3, 64, 128, 256, 512]
SIDE_CH = 16
N_SCALES = 4


def trunk_conv_names():
    """state_dict prefixes of the 13 trunk convs, in execution order.

    Stage 0 is [conv, relu, conv, relu] -> indices 0, 2; later stages start with the pool,
    so their convs sit at 1, 3, (5)  (make_layers_osvos, vgg_osvos.py:136-145)."""
    names = []
    for si, chans in enumerate(STAGE_CHANNELS):
        first = 0 if si == 0 else 1
        names.append(["stages.%d.%d" % (si, first + 2 * j) for j in range(len(chans))])
    return names


def state_dict_spec():
    """(key, shape) in the reference's state_dict order (attribute order vgg_osvos.py:48-54)."""
    spec = []
    for i in range(N_SCALES):
        k = 2 ** (i + 2)
        spec.append(("upscale.%d.weight" % i, (SIDE_CH, SIDE_CH, k, k)))
    for i in range(N_SCALES):
        k = 2 ** (i + 2)
        spec.append(("upscale_.%d.weight" % i, (1, 1, k, k)))
    for si, names in enumerate(trunk_conv_names()):
        cin = STAGE_IN[si]
        for name, cout in zip(names, STAGE_CHANNELS[si]):
            spec.append((name + ".weight", (cout, cin, 3, 3)))
            spec.append((name + ".bias", (cout,)))
            cin = cout
    for i in range(N_SCALES):
        spec.append(("side_prep.%d.weight" % i, (SIDE_CH, STAGE_CHANNELS[i + 1][-1], 3, 3)))
        spec.append(("side_prep.%d.bias" % i, (SIDE_CH,)))
    for i in range(N_SCALES):
        spec.append(("score_dsn.%d.weight" % i, (1, SIDE_CH, 1, 1)))
        spec.append(("score_dsn.%d.bias" % i, (1,)))
    spec.append(("fuse.weight", (1, SIDE_CH * N_SCALES, 1, 1)))
    spec.append(("fuse.bias", (1,)))
    return spec


def bilinear_filter(size):
    """osvos_layers.py:59-67."""
    factor = (size + 1) // 2
    center = factor - 1 if size % 2 == 1 else factor - 0.5
    r = 1.0 - np.abs(np.arange(size) - center) / factor
    return np.outer(r, r)


def crop_to(x, height, width):
    """Same pixels the reference keeps with its negative F.pad (osvos_layers.py:51-56):
    floor(excess/2) dropped on the top/left, ceil(excess/2) on the bottom/right."""
    eh, ew = x.shape[2] - height, x.shape[3] - width
    t, l = eh // 2, ew // 2
    return x[:, :, t:t + height, l:l + width]


def forward(params, x):
    """vgg_osvos.py:59-74.  Returns [side_out0..3, fused], each [N,1,H,W]."""
    H, W = int(x.shape[-2]), int(x.shape[-1])
    names = trunk_conv_names()
    side, side_out = [], []
    for si in range(5):
        if si > 0:
            x = F.max_pool2d(x, kernel_size=2, stride=2, ceil_mode=True)
        for n in names[si]:
            x = F.relu(F.conv2d(x, params[n + ".weight"], params[n + ".bias"], padding=1))
        if si > 0:
            i = si - 1
            s = 2 ** si
            prep = F.conv2d(x, params["side_prep.%d.weight" % i], params["side_prep.%d.bias" % i], padding=1)
            up = F.conv_transpose2d(prep, params["upscale.%d.weight" % i], stride=s)
            side.append(crop_to(up, H, W))
            score = F.conv2d(prep, params["score_dsn.%d.weight" % i], params["score_dsn.%d.bias" % i])
            up1 = F.conv_transpose2d(score, params["upscale_.%d.weight" % i], stride=s)
            side_out.append(crop_to(up1, H, W))
    fused = F.conv2d(torch.cat(side, dim=1), params["fuse.weight"], params["fuse.bias"])
    return side_out + [fused]


def cbce_loss(output, label, size_average=True, batch_average=True):
    """osvos_layers.py:19-48, same operation order."""
    # NB: the reference casts both masks with .float(), so the class weights n_neg/n_tot and
    # n_pos/n_tot are float32 quotients even when the logits are float64 (osvos_layers.py:28-34,41)
    labels = (label >= 0.5).float()
    n_pos = labels.sum()
    n_neg = (1.0 - labels).sum()
    n_tot = n_pos + n_neg
    g = (output >= 0).float()
    val = output * (labels - g) - torch.log(1 + torch.exp(output - 2 * output * g))
    l_pos = (-(labels * val)).sum()
    l_neg = (-((1.0 - labels) * val)).sum()
    final = n_neg / n_tot * l_pos + n_pos / n_tot * l_neg
    if size_average:
        final = final / float(np.prod(label.shape))
    elif batch_average:
        final = final / label.shape[0]
    return final


def sgd_groups(params, lr=1e-8, wd=0.0002, mode="online"):
    """Parameter groups of train_online.py:79-88 (mode 'online') / train_parent.py:87-103
    (mode 'parent'); ``params`` maps state_dict keys to leaf tensors."""
    def pick(prefix, kind):
        return [v for k, v in params.items() if k.startswith(prefix + ".") and k.endswith(kind)]
    groups = [
        {"params": pick("stages", "weight"), "weight_decay": wd},
        {"params": pick("stages", "bias"), "lr": lr * 2},
        {"params": pick("side_prep", "weight"), "weight_decay": wd},
        {"params": pick("side_prep", "bias"), "lr": lr * 2},
    ]
    if mode == "parent":
        groups += [
            {"params": pick("score_dsn", "weight"), "lr": lr / 10, "weight_decay": wd},
            {"params": pick("score_dsn", "bias"), "lr": 2 * lr / 10},
        ]
    groups += [
        {"params": pick("upscale", "weight"), "lr": 0},
        {"params": pick("upscale_", "weight"), "lr": 0},
        {"params": [params["fuse.weight"]], "lr": lr / 100, "weight_decay": wd},
        {"params": [params["fuse.bias"]], "lr": 2 * lr / 100},
    ]
    return groups


def train_loss(params, x, gt, mode="online", epoch=0, n_epochs=240):
    """Loss of one micro-batch: fused head only (train_online.py:127) or the deep-supervision
    mix (train_parent.py:143-147).  Returns (loss, per_head_losses)."""
    outs = forward(params, x)
    if mode == "online":
        l = cbce_loss(outs[-1], gt, size_average=False)
        return l, [l]
    losses = [cbce_loss(o, gt, size_average=False) for o in outs]
    return (1 - epoch / n_epochs) * sum(losses[:-1]) + losses[-1], losses


def as_leaf_params(arrays, dtype=torch.float32, requires_grad=True):
    out = OrderedDict()
    for k, v in arrays.items():
        t = torch.as_tensor(np.asarray(v)).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out
