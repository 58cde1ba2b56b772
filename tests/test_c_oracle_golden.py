"""CPU: pin the plain-C oracle (oracle/osvos_oracle.c) to the golden vectors from the real reference."""
import os

import numpy as np
import pytest

from oracle import c_oracle
from golden_util import CASES, GOLDEN_DIR, check_grad, grad_keys, load_case


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_forward_loss_grads_fp64(name):
    """fp64 C restatement vs the reference run in fp64: tight tolerance (independent code paths)."""
    g, wts, x, m = load_case(name)
    r = c_oracle.net(wts, x, m, side_w=0.75, loss_scale=0.2, want_grads=True, want_dx=True, dtype=np.float64)
    for i in range(5):
        np.testing.assert_allclose(r["outs"][i], g["f64|out%d" % i], rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(r["losses"], g["f64|parent|heads"], rtol=1e-10)
    pre = "f64|parent|grad|"
    for k in grad_keys(g, pre):
        if k.startswith("upscale"):
            continue                      # frozen deconv weights: the C oracle reports zeros
        arr = r["dx"] if k == "input" else r["grads"][k]
        check_grad(g, pre, k, arr, 1e-8, what=name)


@pytest.mark.parametrize("name", CASES[:2])
def test_c_oracle_online_mode_fp32(name):
    """fp32-storage C restatement vs the reference fp32 run (online: fused head only)."""
    g, wts, x, m = load_case(name)
    r = c_oracle.net(wts, x, m, side_w=0.0, loss_scale=0.2, want_grads=True, want_dx=True, dtype=np.float32)
    for i in range(5):
        ref = g["f32|out%d" % i]
        np.testing.assert_allclose(r["outs"][i], ref, rtol=0, atol=2e-4 * max(1.0, np.abs(ref).max()))
    assert abs(r["losses"][4] - float(g["f32|online|loss"])) <= 1e-5 * abs(float(g["f32|online|loss"]))
    pre = "f32|online|grad|"
    keys = grad_keys(g, pre)
    assert "score_dsn.0.weight" not in keys       # online mode: no gradient reaches score_dsn
    for k in keys:
        if k.startswith("upscale"):
            continue
        arr = r["dx"] if k == "input" else r["grads"][k]
        check_grad(g, pre, k, arr, 5e-4, what=name)
    for i in range(4):
        assert not r["grads"]["score_dsn.%d.weight" % i].any()


def test_c_oracle_helpers():
    h = np.load(os.path.join(GOLDEN_DIR, "helpers.npz"))
    for k in (3, 4, 5, 8, 16, 32):
        np.testing.assert_allclose(c_oracle.upsample_filt(k), h["filt|%d" % k], rtol=0, atol=1e-15)
    logits = h["loss|logits"]
    for tag, lab in (("bin", h["loss|lab"]), ("soft", h["loss|soft"]), ("allneg", np.zeros_like(h["loss|lab"])), ("allpos", np.ones_like(h["loss|lab"]))):
        for mode, (sa, ba) in enumerate(((True, True), (False, True), (False, False))):
            loss, grad = c_oracle.cbce(logits.astype(np.float64), lab.astype(np.float64), mode)
            np.testing.assert_allclose(loss, float(h["loss|%s|%d%d|val" % (tag, sa, ba)]), rtol=2e-6, atol=1e-12)
            np.testing.assert_allclose(grad, h["loss|%s|%d%d|grad" % (tag, sa, ba)], rtol=2e-5, atol=1e-9)
