"""CPU: the weight-import adapters of the drop-in constructor -- ``OSVOS(pretrained=1)`` (torchvision-layout
``vgg_pytorch.pth``) and ``OSVOS(pretrained=2)`` (Caffe-export ``vgg_caffe.mat``), reference
networks/vgg_osvos.py:92-125 -- against what the REFERENCE ends up with after loading the same bytes.

The real weight files are absent offline, so both sides load seeded synthetic files of the exact layouts
(oracle/synth.py: write_vgg_pytorch_pth / write_vgg_caffe_mat).  tests/golden/adapters.npz holds the reference's
result (sum, L2 norm, 24 samples and contiguity of each of the 26 trunk tensors), produced by
tests/golden/make_golden.py importing the real reference; this test regenerates the files from the same seeds and
loads them through the drop-in.  Also here: checkpoints whose tensors are not contiguous load correctly, and
``center_crop`` zero-pads like the reference when the target is larger than the input."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR
from oracle import synth


def _check(g, prefix, key, tensor):
    a = tensor.detach().double().numpy().ravel()
    assert abs(a.sum() - float(g[prefix + key + "|sum"])) <= 1e-9 * max(1.0, abs(float(g[prefix + key + "|sum"]))), key
    assert abs(np.sqrt((a * a).sum()) - float(g[prefix + key + "|l2"])) <= 1e-12 * float(g[prefix + key + "|l2"]) + 1e-30, key
    assert np.array_equal(a[g[prefix + key + "|idx"]], g[prefix + key + "|val"]), key


@pytest.fixture()
def models_dir(tmp_path, monkeypatch):
    d = tmp_path / "models"
    d.mkdir()
    monkeypatch.setenv("OSVOS_MODELS_DIR", str(d))
    return d


@pytest.mark.parametrize("pretrained,tag", [(1, "pth"), (2, "mat")])
def test_pretrained_adapters_match_the_reference(models_dir, pretrained, tag):
    import networks.vgg_osvos as vo
    g = np.load(os.path.join(GOLDEN_DIR, "adapters.npz"))
    seed_pth, seed_mat = [int(v) for v in g["meta"]]
    if pretrained == 1:
        synth.write_vgg_pytorch_pth(str(models_dir / "vgg_pytorch.pth"), seed_pth)
    else:
        synth.write_vgg_caffe_mat(str(models_dir / "vgg_caffe.mat"), seed_mat)
    net = vo.OSVOS(pretrained=pretrained)
    sd = net.state_dict()
    trunk = [(k, v) for k, v in sd.items() if k.startswith("stages.")]
    assert len(trunk) == 26
    for k, v in trunk:
        _check(g, tag + "|", k, v)
        assert v.dtype == torch.float32
    # every parameter the kernels (and FusedSGD) will see is contiguous -- as are the reference's (loadmat returns
    # Fortran-ordered arrays, so the reference's .transpose() view is C-contiguous too; recorded in the fixture)
    assert all(v.is_contiguous() for _, v in trunk) and bool(g[tag + "|contiguous"].all())
    # the loaded tensors are Parameters registered under the reference's names, trainable, in the optimizer groups
    assert [k for k, _ in net.stages.named_parameters()] == [k[len("stages."):] for k, _ in trunk]
    assert all(p.requires_grad for p in net.stages.parameters())
    # the heads keep the default init: frozen bilinear deconvs (same bytes as the reference), zero biases
    _check(g, tag + "|", "upscale.2.weight", sd["upscale.2.weight"])
    _check(g, tag + "|", "fuse.bias", sd["fuse.bias"])
    # and the import equals the file's content
    w0, b0 = synth.vgg_trunk_arrays(seed_pth if pretrained == 1 else seed_mat)[4]
    assert np.array_equal(sd["stages.2.1.weight"].numpy(), w0) and np.array_equal(sd["stages.2.1.bias"].numpy(), b0)


def test_adapter_errors(models_dir):
    import networks.vgg_osvos as vo
    with pytest.raises(FileNotFoundError):
        vo.OSVOS(pretrained=1)                  # no file: loud, like the reference's torch.load
    sd = {"features.0.weight": torch.zeros(64, 3, 3, 3), "features.0.bias": torch.zeros(64)}
    torch.save(sd, str(models_dir / "vgg_pytorch.pth"))
    with pytest.raises(ValueError):
        vo.OSVOS(pretrained=1)                  # not a VGG-16: 1 conv layer instead of 13


def test_checkpoint_with_non_contiguous_tensors_loads(tmp_path):
    """A checkpoint may carry tensors with arbitrary strides (torch.save keeps them): e.g. a net whose weights were
    assigned from transposed numpy views as in vgg_osvos.py:117-121.  load_state_dict must give the same values, and the
    module's parameters must end up contiguous."""
    import networks.vgg_osvos as vo
    wts = synth.make_weights(3)
    sd = {}
    for k, v in wts.items():
        t = torch.from_numpy(v.copy())
        if t.dim() == 4:
            t = t.permute(3, 2, 1, 0).contiguous().permute(3, 2, 1, 0)       # same values, reversed strides
            assert not t.is_contiguous() or min(t.shape) == 1 or t.shape[2] == 1
        sd[k] = t
    path = str(tmp_path / "parent_epoch-239.pth")
    torch.save(sd, path)
    back = torch.load(path, map_location=lambda storage, loc: storage)
    assert not back["stages.2.1.weight"].is_contiguous()
    net = vo.OSVOS(pretrained=0)
    net.load_state_dict(back)
    for k, p in net.state_dict().items():
        assert p.is_contiguous() and np.array_equal(p.numpy(), wts[k]), k
    # a state_dict written by the drop-in loads back into a plain dict with the reference's 52 keys in order
    torch.save(net.state_dict(), path)
    again = torch.load(path)
    assert list(again.keys()) == list(wts.keys())


def test_center_crop_zero_pads_like_the_reference():
    from layers.osvos_layers import center_crop
    h = np.load(os.path.join(GOLDEN_DIR, "helpers.npz"))
    keys = [k for k in h.files if k.startswith("cropfull|")]
    assert len(keys) >= 5
    for key in keys:
        hin, win, ht, wt = [int(v) for v in key[9:].split("_")]
        t = torch.arange(1, hin * win + 1, dtype=torch.float32).reshape(1, 1, hin, win)
        got = center_crop(t, ht, wt)
        assert got.shape == h[key].shape and np.array_equal(got.numpy(), h[key]), key
        assert got.data_ptr() != t.data_ptr()                 # a copy, like F.pad's result
