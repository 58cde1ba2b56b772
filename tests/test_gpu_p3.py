"""GPU parity tests of the P3 storage form of the f32x3 arithmetic (csrc/p3.h, conv3x3_p3.hip, p3_ops.hip, wgrad_f32x3.hip P3IN):
fp32 tensors held as their three bf16 piece planes.  Checked against torch float64 (the reference's ATen semantics, SURVEY 8d bars)
AND bit for bit against the fp32-input f32x3 kernels -- the pieces are the same numbers and the products are accumulated in the same
order, so nothing may move."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from osvos_pytorch_amd import ops
    return ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().float().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).double().cpu()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))


def split3_torch(x):
    """the three bf16 pieces as torch forms them: round-to-nearest-even each time (the definition in csrc/p3.h)"""
    h = x.bfloat16()
    r = x - h.float()
    m = r.bfloat16()
    l = (r - m.float()).bfloat16()
    return h, m, l


def test_p3_round_trip_is_lossless_and_matches_the_definition():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(2, 13, 17, 24, generator=g) * torch.exp(4 * torch.randn(2, 13, 17, 24, generator=g))).cuda()
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.5e-20]).cuda()
    p = ops.f32_to_p3(x)
    assert p.shape == (2, 3, 13, 17, 24) and p.dtype == torch.bfloat16
    h, m, l = split3_torch(x)
    assert torch.equal(p[:, 0], h) and torch.equal(p[:, 1], m) and torch.equal(p[:, 2], l)
    assert torch.equal(ops.p3_to_f32(p), x)                           # hi + mid + lo == v exactly
    # ragged channel count: padded with zeros
    y = torch.randn(1, 5, 7, 3, generator=g).cuda()
    q = ops.f32_to_p3(y, cd=8)
    assert torch.equal(ops.p3_to_f32(q)[..., :3], y) and float(ops.p3_to_f32(q)[..., 3:].abs().max()) == 0.0


P3_SHAPES = [
    # N, H, W, Cin, Cout
    (2, 17, 35, 16, 64),
    (1, 33, 70, 64, 128),
    (1, 40, 45, 32, 96),
    (1, 30, 54, 512, 64),
    (2, 9, 11, 48, 32),
]


@pytest.mark.parametrize("shape", P3_SHAPES)
@pytest.mark.parametrize("tile", list(range(10)) + [100, 104, 107, -1])
def test_conv3x3_p3_all_tiles(shape, tile):
    """every tile config: fp32 and P3 results vs float64 at the fp32 bars, P3 result == split of the fp32 result, and the whole thing
    bit-identical to the fp32-input f32x3 convolution (same pieces, same order of products)"""
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000 + 11)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    xd = nhwc(x)
    wpk3 = ops.pack_x3(wt.cuda())
    y, y3 = ops.conv3x3_p3(ops.f32_to_p3(xd), wpk3, b.cuda(), cout, relu=True, tile=tile)
    emax, el2 = rel_err(nchw(y), ref)
    assert emax < 2e-5 and el2 < 1e-5, (shape, tile, emax, el2)
    assert torch.equal(ops.p3_to_f32(y3), y), (shape, tile)
    y_x3 = ops.conv3x3_x3(xd, wpk3, b.cuda(), cout, relu=True)
    assert torch.equal(y, y_x3), (shape, tile, float((y - y_x3).abs().max()))


@pytest.mark.parametrize("tile", [0, 2, 5, 7])
def test_conv3x3_p3_dgrad_masks_splitk_and_skinny_outputs(tile):
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    n, h, w, cin, cout = 1, 37, 43, 64, 64
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    dy = torch.randn(n, cout, h, w, generator=g)
    act = torch.randn(n, cin, h, w, generator=g)                      # the producer's (pre-ReLU-mask) activation: mask = act > 0
    act_relu = F.relu(act)
    ref = F.conv_transpose2d(dy.double(), wt.double(), padding=1) * (act_relu.double() > 0)
    wpk3d = ops.pack_x3(wt.cuda(), dgrad=True)
    dy3 = ops.f32_to_p3(nhwc(dy))
    for mask in (nhwc(act_relu), ops.f32_to_p3(nhwc(act_relu))):      # fp32 mask, P3 mask (plane 0)
        dx, dx3 = ops.conv3x3_p3(dy3, wpk3d, None, cin, mask=mask, tile=tile)
        emax, el2 = rel_err(nchw(dx), ref)
        assert emax < 2e-5 and el2 < 1e-5, (tile, emax, el2)
        assert torch.equal(ops.p3_to_f32(dx3), dx)
    # split-K: same numbers up to the order of the partial sums, P3 mask and P3 result through the finalize kernel
    n, h, w, cin, cout = 1, 15, 27, 512, 64
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    x3 = ops.f32_to_p3(nhwc(x))
    wpk3 = ops.pack_x3(wt.cuda())
    for ks in (2, 4, 8, 0):
        y, y3 = ops.conv3x3_p3(x3, wpk3, b.cuda(), cout, relu=True, tile=tile if tile < 7 else 2, ksplit=ks)
        emax, el2 = rel_err(nchw(y), ref)
        assert emax < 2e-5 and el2 < 1e-5, (ks, emax, el2)
        assert torch.equal(ops.p3_to_f32(y3), y)
    # skinny outputs (fp32 only): side_prep's 16 couts, the 3-channel input gradient
    for co, ycs in ((16, 16), (3, 4)):
        wt = torch.randn(co, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
        ref = F.conv2d(x.double(), wt.double(), padding=1)
        y, _ = ops.conv3x3_p3(x3, ops.pack_x3(wt.cuda()), None, co, want_p3=False, y_cs=ycs, tile=7)
        emax, el2 = rel_err(nchw(y)[:, :co], ref)
        assert emax < 2e-5 and el2 < 1e-5, (co, emax, el2)
        if ycs > co:
            assert float(y[..., co:].abs().max()) == 0.0


WG_SHAPES = [(1, 24, 32, 64, 64), (2, 17, 21, 128, 64), (1, 30, 54, 64, 128), (1, 13, 9, 64, 192)]


@pytest.mark.parametrize("shape", WG_SHAPES)
def test_wgrad_p3_equals_the_fp32_input_kernel_and_float64(shape):
    ops = _ops()
    from osvos_pytorch_amd._lib import F32_X3
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(31 + h)
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    xr = x.double().requires_grad_(False)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wt, padding=1).backward(dy.double())
    dw3, db3 = ops.conv3x3_wgrad_p3(ops.f32_to_p3(nhwc(x)), ops.f32_to_p3(nhwc(dy)), cin, cout)
    emax, el2 = rel_err(dw3.cpu(), wt.grad)
    assert el2 < 1e-6 and emax < 1e-5, (shape, emax, el2)
    torch.testing.assert_close(db3.cpu().double(), dy.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-4)
    dw, db = ops.conv3x3_wgrad(nhwc(x), nhwc(dy), cin, cout, dtype=F32_X3)
    assert torch.equal(dw, dw3), float((dw - dw3).abs().max())
    torch.testing.assert_close(db, db3, rtol=1e-6, atol=1e-5)        # (column sums taken in another order)


@pytest.mark.parametrize("shape", [(1, 9, 11, 8), (2, 17, 35, 64), (1, 30, 53, 128)])
def test_pools_with_p3_results_equal_the_fp32_pools(shape):
    ops = _ops()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(41)
    x = F.relu(torch.randn(n, h, w, c, generator=g) - 0.3).cuda()      # (ties among the ReLU zeros: the first-maximum rule matters)
    y, y3 = ops.maxpool2x2_p3(x, want_f32=True)
    yr = ops.maxpool2x2(x)
    assert torch.equal(y, yr) and torch.equal(ops.p3_to_f32(y3), yr)
    dy = torch.randn(yr.shape, generator=g).cuda()
    ds = torch.randn(x.shape, generator=g).cuda()
    for side in (None, ds):
        dx, dx3 = ops.maxpool2x2_bwd_p3(x, dy, side, want_f32=True)
        dr = ops.maxpool2x2_bwd(x, dy, side)
        assert torch.equal(dx, dr) and torch.equal(ops.p3_to_f32(dx3), dr)


@pytest.mark.parametrize("shape", [(1, 24, 32, 128), (2, 17, 21, 256), (1, 30, 54, 512), (1, 7, 5, 128)])
def test_skinny_side_prep_wgrad_on_the_bf16_pipe(shape):
    """side_prep's weight gradient (Cout = 16; vgg_osvos.py:41) from the P3 stage output and the P3 head gradient -- the S16 form of
    wgrad_f32x3.hip -- against float64 and against the exact fp32 skinny kernel it replaces in the P3 mode"""
    ops = _ops()
    from osvos_pytorch_amd._lib import F32
    n, h, w, cin = shape
    cout = 16
    g = torch.Generator().manual_seed(51 + h)
    x = F.relu(torch.randn(n, cin, h, w, generator=g))
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, padding=1).backward(dy.double())
    dw3, db3 = ops.conv3x3_wgrad_p3(ops.f32_to_p3(nhwc(x)), ops.f32_to_p3(nhwc(dy), cd=16), cin, cout)
    emax, el2 = rel_err(dw3.cpu(), wt.grad)
    assert el2 < 1e-6 and emax < 1e-5, (shape, emax, el2)
    torch.testing.assert_close(db3.cpu().double(), dy.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-4)
    dw, db = ops.conv3x3_wgrad(nhwc(x), nhwc(dy), cin, cout, dtype=F32)
    e2 = rel_err(dw3, dw)
    assert e2[1] < 1e-6, (shape, e2)


def test_skinny_wgrad_fp32_inputs_equal_the_p3_form():
    """the same S16 kernel fed with fp32 tensors (pieces formed while staging: what the default, non-P3 network runs for side_prep)"""
    ops = _ops()
    from osvos_pytorch_amd._lib import F32_X3
    n, h, w, cin, cout = 1, 30, 54, 256, 16
    g = torch.Generator().manual_seed(77)
    x = F.relu(torch.randn(n, h, w, cin, generator=g)).cuda()
    dy = torch.randn(n, h, w, cout, generator=g).cuda()
    dw3, db3 = ops.conv3x3_wgrad_p3(ops.f32_to_p3(x), ops.f32_to_p3(dy, cd=16), cin, cout)
    dw, db = ops.conv3x3_wgrad(x, dy, cin, cout, dtype=F32_X3)
    assert torch.equal(dw, dw3)
    torch.testing.assert_close(db, db3, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("stage0", ["0", "1"])
def test_whole_network_in_p3_storage_mode_matches_the_reference_goldens(tmp_path, stage0):
    """OSVOS_X3_P3=1 (opt-in): trunk tensors as P3, convolutions by LDS-DMA, P3 weight gradients and pools -- forward, losses and every
    gradient of a golden case of the REAL reference, at the fp32 bars; and within fp32 round-off of the default (fp32-tensor) f32x3 network.
    OSVOS_P3_FROM_STAGE = 0 / 1: with and without stage 0 in P3."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import test_gpu_net as T
        from golden_util import load_case
        from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
        g, wts, x, m = load_case("c37x53_n2")
        net = T.build_net(wts, "fp32x3")
        xin = torch.from_numpy(x).requires_grad_()
        outs = net.forward(xin.cuda())
        gt = torch.from_numpy(m).cuda()
        losses = [cbce(o, gt, size_average=False) for o in outs]
        loss = (1 - 60 / 240) * sum(losses[:-1]) + losses[-1]
        (loss / 5).backward()
        res = {"out%%d" %% i: o.detach().cpu().numpy() for i, o in enumerate(outs)}
        res["loss"] = np.array(loss.item())
        res.update({"g:" + k: v.grad.cpu().numpy() for k, v in net.named_parameters() if v.grad is not None})
        res["g:input"] = xin.grad.numpy()
        np.savez(sys.argv[1], **res)
    ''') % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    got = {}
    for flag in ("0", "1"):
        out = str(tmp_path / ("p%s.npz" % flag))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, OSVOS_X3_P3=flag, OSVOS_P3_FROM_STAGE=stage0), timeout=900)
        got[flag] = dict(np.load(out))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_util import check_grad_either, grad_keys, load_case
    g, _, _, _ = load_case("c37x53_n2")
    a, b = got["0"], got["1"]
    for i in range(5):
        ref = g["f32|out%d" % i]
        assert np.abs(b["out%d" % i] - ref).max() <= 1e-3 * ref.std(), i
        assert np.abs(b["out%d" % i] - a["out%d" % i]).max() <= 2e-5 * ref.std(), i
    assert abs(float(b["loss"]) - float(g["f32|parent|loss"])) <= 1e-5 * abs(float(g["f32|parent|loss"]))
    n_checked = 0
    for k in grad_keys(g, "f32|parent|grad|"):
        if k.startswith("upscale"):
            continue
        check_grad_either(g, "parent", k, b["g:" + k], 1e-3, what="p3")
        n_checked += 1
    assert n_checked > 40
