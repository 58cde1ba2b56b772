"""Drop-in for the reference ``networks/vgg_osvos.py``: the same ``OSVOS(pretrained)`` constructor,
module tree, ``state_dict`` layout and ``forward`` contract, with the whole forward/backward
executed by hand-written gfx950 kernels (``osvos_net_forward`` / ``osvos_net_backward``).

The ``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.MaxPool2d`` / ``nn.ReLU`` children are kept as
*parameter containers* so that ``named_parameters()``, ``isinstance`` checks, ``state_dict()`` and
``load_state_dict()`` (incl. checkpoints written by the reference) behave identically
(reference vgg_osvos.py:36-54, Appendix C of SURVEY.md); their ``forward`` methods are never
called by ``OSVOS.forward``."""
from __future__ import division

import math
import os

import torch
import torch.nn as nn

from ..autograd import NetRuntime, OSVOSNetFunction
from ..layers.osvos_layers import interp_surgery

try:  # the path config is a top-level module in the reference (mypath.py); same here
    from mypath import Path
except Exception:  # pragma: no cover
    Path = None

_STAGES = [[64, 64], ['M', 128, 128], ['M', 256, 256, 256], ['M', 512, 512, 512], ['M', 512, 512, 512]]
_STAGE_IN = [3, 64, 128, 256, 512]


def make_layers_osvos(cfg, in_channels):
    """One trunk stage: optional ceil-mode 2x2 max-pool, then (3x3 conv, ReLU) pairs."""
    mods = []
    c = in_channels
    for v in cfg:
        if v == 'M':
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True))
            continue
        mods.append(nn.Conv2d(c, v, kernel_size=3, padding=1))
        mods.append(nn.ReLU(inplace=True))
        c = v
    return nn.Sequential(*mods)


class OSVOS(nn.Module):
    def __init__(self, pretrained=1):
        super(OSVOS, self).__init__()
        print("Constructing OSVOS architecture..")
        stages, side_prep, score_dsn = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        upscale, upscale_ = nn.ModuleList(), nn.ModuleList()
        for i, cfg in enumerate(_STAGES):
            stages.append(make_layers_osvos(cfg, _STAGE_IN[i]))
            if i == 0:
                continue
            side_prep.append(nn.Conv2d(cfg[-1], 16, kernel_size=3, padding=1))
            score_dsn.append(nn.Conv2d(16, 1, kernel_size=1, padding=0))
            upscale_.append(nn.ConvTranspose2d(1, 1, kernel_size=2 ** (1 + i), stride=2 ** i, bias=False))
            upscale.append(nn.ConvTranspose2d(16, 16, kernel_size=2 ** (1 + i), stride=2 ** i, bias=False))
        # registration order fixes the state_dict order: upscale, upscale_, stages, side_prep,
        # score_dsn, fuse (reference vgg_osvos.py:48-54)
        self.upscale = upscale
        self.upscale_ = upscale_
        self.stages = stages
        self.side_prep = side_prep
        self.score_dsn = score_dsn
        self.fuse = nn.Conv2d(64, 1, kernel_size=1, padding=0)
        self._runtime = NetRuntime()
        print("Initializing weights..")
        self._initialize_weights(pretrained)

    def forward(self, x):
        """x: [N,3,H,W] float32 on the GPU.  Returns ``[side_out0..3, fused]`` logits, each
        [N,1,H,W] (reference vgg_osvos.py:59-74)."""
        outs = OSVOSNetFunction.apply(self._runtime, x, *self.parameters())
        return list(outs)

    def set_precision(self, name):
        """'fp32x3' (default: fp32 tensors, fp32-grade results -- the wide 3x3 convolutions run on the bf16 matrix pipe with three-way split
        operands, six bf16 products per fp32 product), 'fp32' (the same arithmetic on the exact fp32 MFMA kernels, ~1.5x slower) or 'bf16'
        (bf16 MFMA operands and bf16 trunk tensors, fp32 accumulate).  'fp32x2' (round 6): the f32x3 kernels with TWO bf16 pieces per operand,
        three products -- fp32 tensors, 16-bit-significand operands (finer than the TF32 cuDNN defaults to for fp32 convolutions), fp32
        accumulate, ~half the matrix work; logits ~1e-4 std and losses ~1e-5 from fp32, NOT inside every flat fp32 bar.
        'fp32x3b2' / 'fp32x3h2': the forward of 'fp32x3' bit for bit, the BACKWARD convolutions on two bf16 pieces / on two FP16 pieces under block
        exponents (22-23-bit operands) -- three products, every parity number of 'fp32x3' unchanged, 1.2-1.3x its step rate.  'fp32h2': FP16 pairs in
        both passes (activations closer to float64 than 'fp32x3''s; see DESIGN.md 3.1a for what that does and does not buy in training).
        Not part of the reference's API."""
        self._runtime.set_precision(name)
        return self

    def set_inplace_grad_accumulation(self, on=True):
        """Opt in to accumulating parameter gradients straight into existing ``.grad`` tensors inside the backward kernels
        (gradient-accumulation loops: no temporaries, no per-tensor add kernels).  Only valid when every backward through
        this module is a ``loss.backward()`` -- under ``torch.autograd.grad`` leave it off.  Not part of the reference's API."""
        self._runtime.inplace_accumulate = bool(on)
        return self

    def set_deferred_backward_join(self, on=True):
        """Opt in (together with in-place gradient accumulation) to backwards that return before their weight-gradient tail has finished
        on the side streams; call ``join_backward()`` before reading any ``.grad`` (TrainLoop does, before the optimizer step).  Not part
        of the reference's API."""
        self._runtime.defer_join = bool(on)
        return self

    def join_backward(self):
        self._runtime.join_backward()
        return self

    def invalidate_packed_weights(self):
        """Force a re-pack of the parameters on the next forward.  Needed only after writing parameters through ``.data``
        (``p.data.copy_(...)``, ``dist.broadcast(p.data)``): such writes do not bump ``p._version``, which keys the cache."""
        self._runtime.key = None
        return self

    def _initialize_weights(self, pretrained):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.001)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.ConvTranspose2d):
                m.weight.data.zero_()
                m.weight.data = interp_surgery(m)
        if pretrained == 1:
            print("Loading weights from PyTorch VGG")
            self._load_pytorch_vgg(os.path.join(Path.models_dir(), 'vgg_pytorch.pth'))
        elif pretrained == 2:
            print("Loading weights from Caffe VGG")
            self._load_caffe_vgg(os.path.join(Path.models_dir(), 'vgg_caffe.mat'))

    def _trunk_convs(self):
        return [m for stage in self.stages for m in stage if isinstance(m, nn.Conv2d)]

    def _load_pytorch_vgg(self, path):
        """torchvision-style VGG-16 checkpoint: the 13 ``features.<i>.weight/bias`` conv tensors,
        in order, become the trunk (reference vgg_osvos.py:92-109)."""
        sd = torch.load(path, map_location=lambda storage, loc: storage)
        feats = sorted({int(k.split('.')[1]) for k in sd if k.startswith('features.') and k.endswith('.weight') and sd[k].dim() == 4})
        convs = self._trunk_convs()
        if len(feats) != len(convs):
            raise ValueError("expected %d conv layers in %s, found %d" % (len(convs), path, len(feats)))
        for conv, fi in zip(convs, feats):
            conv.weight = nn.Parameter(sd['features.%d.weight' % fi].clone())
            conv.bias = nn.Parameter(sd['features.%d.bias' % fi].clone())

    def _load_caffe_vgg(self, path):
        """Caffe VGG exported to .mat: weights[0][i] is stored [kw,kh,cin,cout] -> transpose to
        OIHW; biases[0][i] is [cout,1] (reference vgg_osvos.py:110-125)."""
        import scipy.io
        mat = scipy.io.loadmat(path)
        for i, conv in enumerate(self._trunk_convs()):
            w = torch.from_numpy(mat['weights'][0][i].transpose().copy())
            b = torch.from_numpy(mat['biases'][0][i][:, 0].copy())
            assert conv.weight.data.shape == w.shape
            assert conv.bias.data.shape == b.shape
            conv.weight.data = w
            conv.bias.data = b


def find_conv_layers(_vgg):
    return [i for i, m in enumerate(_vgg.features) if isinstance(m, nn.Conv2d)]


def make_layers(cfg, batch_norm=False):
    """Plain VGG feature stack (floor-mode pools); only a weight-file adapter, never run here."""
    mods, c = [], 3
    for v in cfg:
        if v == 'M':
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        mods.append(nn.Conv2d(c, v, kernel_size=3, padding=1))
        if batch_norm:
            mods.append(nn.BatchNorm2d(v))
        mods.append(nn.ReLU(inplace=True))
        c = v
    return nn.Sequential(*mods)


class VGG(nn.Module):
    """Classifier shell matching torchvision's VGG-16 state_dict (reference vgg_osvos.py:148-182);
    kept so ``vgg_pytorch.pth`` can be loaded through it.  Not on the hot path."""

    def __init__(self, features, num_classes=1000):
        super(VGG, self).__init__()
        self.features = features
        self.classifier = nn.Sequential(
            nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
            nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(),
            nn.Linear(4096, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.view(x.size(0), -1))
