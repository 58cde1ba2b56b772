"""CPU: pin the torch-CPU functional oracle to the golden vectors made from the real reference."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from golden_util import CASES, check_grad, grad_keys, load_case, GOLDEN_DIR
import os


@pytest.mark.parametrize("name", CASES)
def test_forward_and_loss_match_reference(name):
    g, wts, x, m = load_case(name)
    p = torch_ref.as_leaf_params(wts, requires_grad=False)
    with torch.no_grad():
        outs = torch_ref.forward(p, torch.from_numpy(x))
    for i, o in enumerate(outs):
        ref = g["f32|out%d" % i]
        assert o.shape == ref.shape
        # same ATen kernels, same op order -> bitwise or within a few ulp
        np.testing.assert_allclose(o.numpy(), ref, rtol=0, atol=1e-4 * max(1.0, np.abs(ref).max()))
    heads = [torch_ref.cbce_loss(o, torch.from_numpy(m), size_average=False).item() for o in outs]
    np.testing.assert_allclose(heads, g["f32|parent|heads"], rtol=2e-6)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["online", "parent"])
def test_gradients_match_reference(name, mode):
    g, wts, x, m = load_case(name)
    p = torch_ref.as_leaf_params(wts)
    xin = torch.from_numpy(x).requires_grad_()
    loss, _ = torch_ref.train_loss(p, xin, torch.from_numpy(m), mode=mode, epoch=60, n_epochs=240)
    (loss / 5).backward()
    pre = "f32|%s|" % mode
    assert abs(loss.item() - float(g[pre + "loss"])) <= 2e-6 * abs(float(g[pre + "loss"]))
    keys = grad_keys(g, pre + "grad|")
    have = {k: v.grad for k, v in p.items() if v.grad is not None}
    have["input"] = xin.grad
    assert set(keys) == set(have)
    for k in keys:
        check_grad(g, pre + "grad|", k, have[k].numpy(), 2e-4, what=name)


def test_fp64_oracle_matches_fp64_reference():
    g, wts, x, m = load_case("c37x53_n2")
    p = torch_ref.as_leaf_params(wts, dtype=torch.float64, requires_grad=False)
    with torch.no_grad():
        outs = torch_ref.forward(p, torch.from_numpy(x).double())
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), g["f64|out%d" % i], rtol=1e-10, atol=1e-9)


def test_sgd_trajectory_matches_reference():
    g, wts, x, m = load_case("c48x64")
    p = torch_ref.as_leaf_params(wts)
    opt = torch.optim.SGD(torch_ref.sgd_groups(p, mode="online"), lr=1e-8, momentum=0.9)
    w0 = {k: v.detach().clone() for k, v in p.items()}
    losses = []
    for it in range(4):
        loss, _ = torch_ref.train_loss(p, torch.from_numpy(x).requires_grad_(), torch.from_numpy(m), mode="online")
        losses.append(loss.item())
        (loss / 2).backward()
        if it % 2 == 1:
            opt.step()
            opt.zero_grad()
    np.testing.assert_allclose(losses, g["sgd|losses"], rtol=1e-5)
    for k in w0:
        check_grad(g, "sgd|delta|", k, (p[k].detach() - w0[k]).numpy(), 2e-3, what="sgd")


def test_helpers_match_reference():
    h = np.load(os.path.join(GOLDEN_DIR, "helpers.npz"))
    for k in (3, 4, 5, 8, 16, 32):
        np.testing.assert_allclose(torch_ref.bilinear_filter(k), h["filt|%d" % k], rtol=0, atol=1e-15)
    for key in [f for f in h.files if f.startswith("crop|")]:
        hin, win, ht, wt = [int(v) for v in key[5:].split("_")]
        t = torch.arange(hin * win, dtype=torch.float32).reshape(1, 1, hin, win)
        c = torch_ref.crop_to(t, ht, wt)
        first = int(c[0, 0, 0, 0].item())
        assert [first // win, first % win, c.shape[2], c.shape[3]] == list(h[key])
    logits = torch.from_numpy(h["loss|logits"])
    for tag, lab in (("bin", h["loss|lab"]), ("soft", h["loss|soft"]), ("allneg", np.zeros_like(h["loss|lab"])), ("allpos", np.ones_like(h["loss|lab"]))):
        for sa, ba in ((True, True), (False, True), (False, False)):
            xin = logits.clone().requires_grad_()
            l = torch_ref.cbce_loss(xin, torch.from_numpy(lab), size_average=sa, batch_average=ba)
            l.backward()
            np.testing.assert_allclose(l.item(), float(h["loss|%s|%d%d|val" % (tag, sa, ba)]), rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(xin.grad.numpy(), h["loss|%s|%d%d|grad" % (tag, sa, ba)], rtol=1e-5, atol=1e-9)
    keys = [k for k, _ in torch_ref.state_dict_spec()]
    shapes = [",".join(map(str, s)) for _, s in torch_ref.state_dict_spec()]
    assert keys == list(h["sd|keys"])
    assert shapes == list(h["sd|shapes"])
