"""GPU parity tests, op by op, through the C ABI (osvos_pytorch_amd.ops) against CPU oracles
(torch float64 functional ops = the reference's ATen semantics; oracle/c_oracle for the loss).
fp32 tolerances: conv outputs within 2e-5 * rms-scale (fp32 MFMA == fmaf chain), see SURVEY 8d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from osvos_pytorch_amd import ops
    return ops


def nhwc(t):  # cpu NCHW -> cuda NHWC contiguous
    return t.permute(0, 2, 3, 1).contiguous().float().cuda()


def nchw(t):  # cuda NHWC -> cpu NCHW float64
    return t.permute(0, 3, 1, 2).double().cpu()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))


def test_mfma_fragment_layout():
    """v_mfma_f32_32x32x2_f32: A[i][k] lane = i + 32k, B[k][j] lane = j + 32k,
    D[row][col]: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)."""
    out = _ops().debug_mfma_layout().cpu().numpy()
    lane = np.arange(64)[:, None]
    r = np.arange(16)[None, :]
    col = lane & 31
    row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    for v, (wa, wb) in enumerate([((1, 0), (1, 0)), ((0, 3), (0, 7)), ((1, 3), (1, 7))]):
        exp = (row + 1) * (col + 1) * 100.0 * (wa[0] * wb[0] + wa[1] * wb[1])
        np.testing.assert_array_equal(out[v], exp.astype(np.float32))


CONV_SHAPES = [
    # N, H, W, Cin, Cout
    (1, 9, 11, 8, 32),
    (2, 17, 35, 16, 64),
    (1, 33, 70, 64, 128),
    (1, 8, 8, 24, 16),
    (1, 40, 45, 32, 96),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("tile", list(range(15)) + [100, 103, 109, 113, 114, -1])
def test_conv3x3_forward_all_tiles(shape, tile):
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000 + 3)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    y = ops.conv3x3(nhwc(x), ops.pack_fwd(wt.cuda()), b.cuda(), cout, relu=True, tile=tile)
    emax, el2 = rel_err(nchw(y), ref)
    assert emax < 2e-5 and el2 < 1e-5, (shape, tile, emax, el2)


X3_SHAPES = [
    # N, H, W, Cin, Cout  (f32x3 needs Cin % 16 == 0, Cout % 4 == 0 and >= 32)
    (2, 17, 35, 16, 64),
    (1, 33, 70, 64, 128),
    (1, 40, 45, 32, 96),
    (1, 30, 54, 512, 64),
    (2, 9, 11, 48, 32),
]


@pytest.mark.parametrize("shape", X3_SHAPES)
@pytest.mark.parametrize("tile", [200 + k for k in range(18)] + [301, 305, 310, 316])
def test_conv3x3_f32x3_all_tiles(shape, tile):
    """f32x3 (three-way bf16 split on the bf16 matrix pipe) is held to the SAME float64 bars as the exact fp32 MFMA kernel."""
    ops = _ops()
    assert ops.lib().osvos_conv3x3_f32x3_tiles() == 18
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000 + 3)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    y = ops.conv3x3(nhwc(x), ops.pack_fwd(wt.cuda()), b.cuda(), cout, relu=True, tile=tile)
    emax, el2 = rel_err(nchw(y), ref)
    assert emax < 2e-5 and el2 < 1e-5, (shape, tile, emax, el2)


def test_conv3x3_f32x3_is_fp32_grade_and_covers_mask_stride_dgrad_splitk():
    """(1) error against float64 no worse than 2x the exact fp32 kernel's on a K = 9 x 512 reduction with wide-range data;
    (2) mask / channel stride / no bias; (3) the data-gradient pack; (4) split-K; (5) dtype OSVOS_F32_X3 and the process-wide mode."""
    ops = _ops()
    from osvos_pytorch_amd._lib import F32_X3
    g = torch.Generator().manual_seed(77)
    n, h, w, cin, cout = 1, 30, 54, 512, 128
    # activations spread over 6 decades (post-ReLU like: half of them zero), weights He-scaled
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp(torch.randn(n, cin, h, w, generator=g) * 2.0)
    x = x * (torch.rand(n, cin, h, w, generator=g) > 0.5)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cout)) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, padding=1)
    pk = ops.pack_fwd(wt.cuda())
    e_exact = rel_err(nchw(ops.conv3x3(nhwc(x), pk, None, cout, tile=9)), ref)
    e_x3 = rel_err(nchw(ops.conv3x3(nhwc(x), pk, None, cout, tile=201)), ref)
    print("K=4608 conv vs float64: exact fp32 MFMA max %.2e l2 %.2e | f32x3 max %.2e l2 %.2e" % (e_exact + e_x3))
    assert e_x3[1] <= 2.0 * e_exact[1] + 1e-8 and e_x3[0] <= 3.0 * e_exact[0] + 1e-8, (e_exact, e_x3)
    # (2)
    n, h, w, cin, cout = 2, 21, 37, 32, 40
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 17
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    xg, pk = nhwc(x), ops.pack_fwd(wt.cuda())
    y = ops.conv3x3(xg, pk, b.cuda(), cout, relu=False, y_cs=48, tile=205)
    assert rel_err(nchw(y[..., :cout]), ref)[0] < 2e-5
    assert float(y[..., cout:].abs().max()) == 0.0
    m = torch.randn(n, cout, h, w, generator=g)
    ym = ops.conv3x3(xg, pk, b.cuda(), cout, relu=False, mask=nhwc(m), tile=203)
    assert rel_err(nchw(ym), ref * (m > 0))[0] < 2e-5
    # (3) data gradient = conv of dy with the rotated pack: 40 -> 32 channels needs Cin(=cout here) % 16: use 48 -> 32
    cin, cout = 32, 48
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / 17
    dy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64)
    F.conv2d(x, wt, None, padding=1).backward(dy)
    dx = ops.conv3x3(nhwc(dy), ops.pack_dgrad(wt.float().cuda()), None, cin, relu=False, tile=205)
    assert rel_err(nchw(dx), x.grad)[0] < 3e-5
    # (3b) skinny outputs: 16 couts (side_prep) and the ragged 3-channel input gradient written into a 4-channel buffer
    n, h, w, cin = 1, 30, 54, 64
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(16, cin, 3, 3, generator=g) / 24
    b = torch.randn(16, generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    for t in (215, 205, -1):
        y = ops.conv3x3(nhwc(x), ops.pack_fwd(wt.cuda()), b.cuda(), 16, relu=False, tile=t, dtype=F32_X3)
        assert rel_err(nchw(y), ref)[0] < 2e-5, t
    xin = torch.randn(n, 3, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    w1 = torch.randn(64, 3, 3, 3, generator=g, dtype=torch.float64) / 5
    dy = torch.randn(n, 64, h, w, generator=g, dtype=torch.float64)
    F.conv2d(xin, w1, None, padding=1).backward(dy)
    for t in (215, -1):
        dx = ops.conv3x3(nhwc(dy), ops.pack_dgrad(w1.float().cuda()), None, 3, relu=False, y_cs=4, tile=t, dtype=F32_X3)
        assert rel_err(nchw(dx[..., :3]), xin.grad)[0] < 3e-5, t
        assert float(dx[..., 3].abs().max()) == 0.0
    # (4) split-K through the f32x3 kernel (partial sums + finalize with bias / ReLU)
    n, h, w, cin, cout = 1, 15, 27, 256, 64
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 48
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    for ks in (2, 4, 8):
        y = ops.conv3x3_splitk(nhwc(x), ops.pack_fwd(wt.cuda()), b.cuda(), cout, ks, relu=True, tile=205)
        assert rel_err(nchw(y), ref)[0] < 2e-5, ks
    # (5) automatic choice under dtype OSVOS_F32_X3 (the arithmetic is a per-call argument: there is no process-wide mode), also through
    # the split-K entry point
    y3 = ops.conv3x3(nhwc(x), ops.pack_fwd(wt.cuda()), b.cuda(), cout, relu=True, dtype=F32_X3)
    assert rel_err(nchw(y3), ref)[0] < 2e-5
    ym = ops.conv3x3_splitk(nhwc(x), ops.pack_fwd(wt.cuda()), b.cuda(), cout, 0, relu=True, dtype=F32_X3)
    assert rel_err(nchw(ym), ref)[0] < 2e-5
    assert float((ym - y3).abs().max()) <= 4e-6 * float(y3.abs().max())      # the same arithmetic (possibly cut along K): fp32 summation order apart
    ye = ops.conv3x3(nhwc(x), ops.pack_fwd(wt.cuda()), b.cuda(), cout, relu=True)       # dtype F32: the exact kernel, another rounding
    assert rel_err(nchw(ye), ref)[0] < 2e-5 and not torch.equal(ye, y3)


@pytest.mark.parametrize("shape", [(1, 13, 21, 64, 64), (2, 30, 54, 128, 64), (1, 60, 107, 64, 128), (1, 25, 37, 64, 64),
                                   (3, 6, 16, 64, 64), (1, 121, 215, 64, 64), (2, 30, 54, 128, 256), (1, 25, 37, 64, 128), (2, 5, 17, 64, 128)])
def test_wgrad_f32x3(shape):
    """f32x3 weight gradient (three-way bf16 split, pixel-major tiles gathered with ds_read_b64_tr_b16): the SAME float64 bars as the
    exact fp32 kernel (test_conv3x3_dgrad_and_wgrad); odd sizes, batch > 1, accumulation, and the error next to the exact kernel's."""
    ops = _ops()
    from osvos_pytorch_amd._lib import F32_X3
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(19 + h)
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp(torch.randn(n, cin, h, w, generator=g))
    x = x * (torch.rand(n, cin, h, w, generator=g) > 0.4)          # post-ReLU like operand
    dy = torch.randn(n, cout, h, w, generator=g) * torch.exp(torch.randn(n, cout, h, w, generator=g))
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, b, padding=1).backward(dy.double())
    xg, dyg = nhwc(x), nhwc(dy)
    dw, db = ops.conv3x3_wgrad(xg, dyg, cin, cout, dtype=F32_X3)
    dwe, dbe = ops.conv3x3_wgrad(xg, dyg, cin, cout)
    e3, ee = rel_err(dw.cpu(), wt.grad), rel_err(dwe.cpu(), wt.grad)
    print("wgrad %s vs float64: exact fp32 max %.2e l2 %.2e | f32x3 max %.2e l2 %.2e" % ((shape,) + ee + e3))
    assert e3[0] < 3e-5 and e3[1] < 1e-5, (shape, e3)
    assert e3[1] <= 2.0 * ee[1] + 1e-7, (shape, e3, ee)
    assert rel_err(db.cpu(), b.grad)[0] < 3e-5, shape
    dw2, db2 = ops.conv3x3_wgrad(xg, dyg, cin, cout, accumulate_into=(dw.clone(), db.clone()), dtype=F32_X3)
    assert rel_err(dw2.cpu(), 2 * wt.grad)[0] < 3e-5 and rel_err(db2.cpu(), 2 * b.grad)[0] < 3e-5
    # shapes it does not take fall back to the exact kernels under the same dtype (conv1_1: Cin 3 padded to 8; side_prep: Cout 16)
    x8 = torch.randn(1, 9, 11, 8, device="cuda")
    dy64 = torch.randn(1, 9, 11, 64, device="cuda")
    a1, _ = ops.conv3x3_wgrad(x8, dy64, 3, 64, dtype=F32_X3)
    a0, _ = ops.conv3x3_wgrad(x8, dy64, 3, 64)
    assert torch.equal(a1, a0)


@pytest.mark.parametrize("shape", [(2, 17, 35, 16, 64), (1, 33, 70, 64, 128), (1, 40, 45, 32, 96), (1, 30, 54, 512, 64), (1, 21, 37, 64, 16)])
def test_conv3x3_f32x3_presplit_weights_are_bit_identical(shape):
    """the pre-split bf16x3 weight pack (formed once per optimizer step) feeds the SAME pieces to the MFMAs as the in-kernel split of the
    fp32 pack: forward and data gradient must agree bit for bit on every eight-wave tile, and meet the float64 bars"""
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    xg, pk, pk3 = nhwc(x), ops.pack_fwd(wt.cuda()), ops.pack_x3(wt.cuda())
    for t in (10, 12, 14, 15, 16, 17, -1):
        y3 = ops.conv3x3_x3(xg, pk3, b.cuda(), cout, relu=True, tile=t)
        assert rel_err(nchw(y3), ref)[0] < 2e-5, (shape, t)
        if t >= 0:
            assert torch.equal(y3, ops.conv3x3(xg, pk, b.cuda(), cout, relu=True, tile=200 + t)), (shape, t)
    if cout % 16 == 0:
        dy = torch.randn(n, cout, h, w, generator=g)
        dpk, dpk3 = ops.pack_dgrad(wt.cuda()), ops.pack_x3(wt.cuda(), dgrad=True)
        m = torch.randn(n, cin, h, w, generator=g)
        a = ops.conv3x3_x3(nhwc(dy), dpk3, None, cin, mask=nhwc(m), tile=12)
        assert torch.equal(a, ops.conv3x3(nhwc(dy), dpk, None, cin, mask=nhwc(m), tile=212))
        ref_dx = torch.nn.grad.conv2d_input(x.shape, wt.double(), dy.double(), padding=1) * (m > 0)
        assert rel_err(nchw(a), ref_dx)[0] < 3e-5


def test_conv3x3_matches_naive_kernel_and_mask_and_stride():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    n, h, w, cin, cout = 1, 21, 37, 16, 16
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 12
    b = torch.randn(cout, generator=g)
    xg = nhwc(x)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    naive = ops.debug_conv3x3_naive(xg, wt.cuda(), b.cuda())
    assert rel_err(nchw(naive), ref)[0] < 2e-5
    y = ops.conv3x3(xg, ops.pack_fwd(wt.cuda()), b.cuda(), cout, relu=False, y_cs=24)
    assert rel_err(nchw(y[..., :cout]), ref)[0] < 2e-5
    assert float(y[..., cout:].abs().max()) == 0.0            # channels beyond Cout untouched
    m = torch.randn(n, cout, h, w, generator=g)
    ym = ops.conv3x3(xg, ops.pack_fwd(wt.cuda()), b.cuda(), cout, relu=False, mask=nhwc(m))
    assert rel_err(nchw(ym), ref * (m > 0))[0] < 2e-5


@pytest.mark.parametrize("shape", [(1, 9, 11, 8, 32), (2, 17, 35, 16, 64), (1, 20, 33, 3, 64), (1, 12, 13, 64, 16),
                                   (2, 37, 53, 3, 64), (1, 64, 70, 3, 64), (2, 19, 45, 128, 16), (1, 30, 54, 256, 16), (1, 8, 8, 32, 16)])
def test_conv3x3_dgrad_and_wgrad(shape):
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / (3 * cin ** 0.5)).requires_grad_()
    b = torch.randn(cout, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64)
    F.conv2d(x, wt, b, padding=1).backward(dy)
    cin_s = (cin + 7) // 8 * 8
    xp = torch.zeros(n, cin_s, h, w)
    xp[:, :cin] = x.detach().float()
    dyg = nhwc(dy)
    dx = ops.conv3x3(dyg, ops.pack_dgrad(wt.detach().float().cuda()), None, cin, relu=False, y_cs=max(cin, 4) if cin < 8 else cin)
    assert rel_err(nchw(dx[..., :cin]), x.grad)[0] < 3e-5, shape
    dw, db = ops.conv3x3_wgrad(nhwc(xp), dyg, cin, cout)
    assert rel_err(dw.cpu(), wt.grad)[0] < 3e-5, shape
    assert rel_err(db.cpu(), b.grad)[0] < 3e-5, shape
    dw2, db2 = ops.conv3x3_wgrad(nhwc(xp), dyg, cin, cout, accumulate_into=(dw.clone(), db.clone()))
    assert rel_err(dw2.cpu(), 2 * wt.grad)[0] < 3e-5


def test_wgrad_many_patches_and_splits():
    """long reduction (many 32x2 patches, several splits) keeps fp32-level accuracy"""
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    n, h, w, cin, cout = 2, 60, 107, 64, 64
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, None, padding=1).backward(dy.double())
    dw, db = ops.conv3x3_wgrad(nhwc(x), nhwc(dy), cin, cout)
    assert rel_err(dw.cpu(), wt.grad)[1] < 1e-5
    assert rel_err(db.cpu(), dy.double().sum((0, 2, 3)))[0] < 1e-5


@pytest.mark.parametrize("hw", [(8, 8), (9, 13), (1, 1), (7, 2), (30, 54), (61, 107)])
def test_maxpool_forward_backward(hw):
    ops = _ops()
    h, w = hw
    g = torch.Generator().manual_seed(h * 100 + w)
    c = 8
    # quantised positive values -> many exact ties; ReLU-like zeros too
    x = (torch.randint(-2, 4, (2, c, h, w), generator=g).clamp(min=0)).double().requires_grad_()
    y = F.max_pool2d(x, 2, 2, ceil_mode=True)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    yg = ops.maxpool2x2(nhwc(x.detach()))
    assert torch.equal(nchw(yg), y.detach())
    side = torch.randn(x.shape, generator=g, dtype=torch.float64)
    dxg = ops.maxpool2x2_bwd(nhwc(x.detach()), nhwc(dy), nhwc(side))
    ref = (x.grad + side) * (x.detach() > 0)
    assert rel_err(nchw(dxg), ref)[0] < 1e-6
    dxg2 = ops.maxpool2x2_bwd(nhwc(x.detach()), nhwc(dy), None)
    assert rel_err(nchw(dxg2), x.grad * (x.detach() > 0))[0] < 1e-6


def test_maxpool_negative_partial_window():
    """an all-negative clipped window returns the real max (never a padded zero)"""
    ops = _ops()
    x = -torch.rand(1, 4, 5, 5) - 1
    y = ops.maxpool2x2(nhwc(x))
    assert torch.equal(nchw(y), F.max_pool2d(x.double(), 2, 2, ceil_mode=True))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("kind", ["bin", "soft", "allneg", "allpos"])
def test_cbce_loss_and_grad(mode, kind):
    from oracle import c_oracle
    ops = _ops()
    rng = np.random.default_rng(3)
    logits = (rng.standard_normal((2, 1, 37, 53)) * 4).astype(np.float32)
    lab = {"bin": (rng.random(logits.shape) > 0.8), "soft": rng.random(logits.shape),
           "allneg": np.zeros(logits.shape), "allpos": np.ones(logits.shape)}[kind].astype(np.float32)
    ref_loss, ref_grad = c_oracle.cbce(logits.astype(np.float64), lab.astype(np.float64), mode)
    loss, grad = ops.cbce(torch.from_numpy(logits).cuda(), torch.from_numpy(lab).cuda(), mode)
    assert abs(loss.item() - ref_loss) <= 1e-5 * abs(ref_loss) + 1e-12
    np.testing.assert_allclose(grad.cpu().numpy(), ref_grad, rtol=2e-5, atol=2e-8 * np.abs(ref_grad).max())
    if kind == "allneg":
        assert loss.item() == 0.0


def test_fused_sgd_matches_torch():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    p = torch.randn(1000, generator=g)
    ref = p.clone().requires_grad_()
    opt = torch.optim.SGD([ref], lr=1e-2, momentum=0.9, weight_decay=2e-4)
    pg, buf = p.clone().cuda(), torch.zeros(1000).cuda()
    for it in range(3):
        gr = torch.randn(1000, generator=g)
        ref.grad = gr.clone()
        opt.step()
        ops.sgd_step(pg, gr.cuda(), buf, 1e-2, 0.9, 2e-4, it == 0)
    np.testing.assert_allclose(pg.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)


BF16_SHAPES = [(1, 9, 11, 8, 32), (2, 17, 35, 16, 64), (1, 33, 70, 64, 128), (1, 8, 8, 24, 16), (1, 40, 45, 32, 96), (1, 20, 24, 96, 64)]


@pytest.mark.parametrize("shape", BF16_SHAPES)
@pytest.mark.parametrize("tile", list(range(12)) + [100, 106, 108, 110, -1])
def test_conv3x3_bf16_mfma_forward_all_tiles(shape, tile):
    """bf16-operand path: with inputs that are already bf16-representable the only difference to a float64
    convolution is the fp32 accumulation order -> tight tolerance; this pins layout/indexing, not precision"""
    from osvos_pytorch_amd._lib import F32_BF16MFMA
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000 + 17)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16().float()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).bfloat16().float()
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    y = ops.conv3x3(nhwc(x), ops.pack_fwd(wt.cuda(), F32_BF16MFMA), b.cuda(), cout, relu=True, tile=tile, dtype=F32_BF16MFMA)
    emax, el2 = rel_err(nchw(y), ref)
    assert emax < 3e-5 and el2 < 1e-5, (shape, tile, emax, el2)


def test_conv3x3_bf16_mfma_rounding_and_dgrad():
    """unrounded fp32 inputs: result equals the convolution of the RNE-bf16-rounded operands; the data-gradient pack
    (rotated / transposed filter) goes through the same kernel"""
    from osvos_pytorch_amd._lib import F32_BF16MFMA
    ops = _ops()
    g = torch.Generator().manual_seed(23)
    n, h, w, cin, cout = 2, 19, 27, 64, 32
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 24
    ref = F.conv2d(x.bfloat16().double(), wt.bfloat16().double(), None, padding=1)
    y = ops.conv3x3(nhwc(x), ops.pack_fwd(wt.cuda(), F32_BF16MFMA), None, cout, dtype=F32_BF16MFMA)
    assert rel_err(nchw(y), ref)[0] < 3e-5
    exact = F.conv2d(x.double(), wt.double(), None, padding=1)
    assert 1e-4 < rel_err(nchw(y), exact)[1] < 2e-2          # it really is bf16 arithmetic
    dy = torch.randn(n, cout, h, w, generator=g)
    xr = x.double().requires_grad_()
    F.conv2d(xr, wt.bfloat16().double(), None, padding=1).backward(dy.bfloat16().double())
    m = torch.randn(n, cin, h, w, generator=g)
    dx = ops.conv3x3(nhwc(dy), ops.pack_dgrad(wt.cuda(), F32_BF16MFMA), None, cin, mask=nhwc(m), dtype=F32_BF16MFMA)
    assert rel_err(nchw(dx), xr.grad * (m > 0))[0] < 3e-5


@pytest.mark.parametrize("shape", [(1, 9, 11, 64, 64), (2, 17, 35, 64, 128), (1, 33, 70, 128, 64), (1, 60, 107, 192, 128), (3, 8, 40, 64, 64)])
def test_wgrad_bf16_mfma(shape):
    """bf16-operand weight gradient: with bf16-representable inputs only the fp32 accumulation order differs from
    float64 (pins the transposed staging, the in-register tap shifts and the slab layout); bias gradient is exact fp32"""
    from osvos_pytorch_amd._lib import F32_BF16MFMA
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g).bfloat16().float()
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, None, padding=1).backward(dy.bfloat16().double())
    dw, db = ops.conv3x3_wgrad(nhwc(x), nhwc(dy), cin, cout, dtype=F32_BF16MFMA)
    assert rel_err(dw.cpu(), wt.grad)[0] < 3e-5, shape
    assert rel_err(db.cpu(), dy.double().sum((0, 2, 3)))[0] < 1e-5
    dw2, _ = ops.conv3x3_wgrad(nhwc(x), nhwc(dy), cin, cout, accumulate_into=(dw.clone(), db.clone()), dtype=F32_BF16MFMA)
    assert rel_err(dw2.cpu(), 2 * wt.grad)[0] < 3e-5


@pytest.mark.parametrize("ksplit", [0, 2, 3, 4, 8])
def test_conv3x3_split_k(ksplit):
    """K cut into parts + finalize kernel (bias, ReLU, mask) equals the single-pass result up to fp32 summation order"""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    n, h, w, cin, cout = 2, 30, 54, 256, 64
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 48
    b = torch.randn(cout, generator=g)
    m = torch.randn(n, cout, h, w, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    y = ops.conv3x3_splitk(nhwc(x), ops.pack_fwd(wt.cuda()), b.cuda(), cout, ksplit, relu=True, tile=9)
    assert rel_err(nchw(y), ref)[0] < 2e-5
    ym = ops.conv3x3_splitk(nhwc(x), ops.pack_fwd(wt.cuda()), None, cout, ksplit, relu=False, mask=nhwc(m))
    ref2 = F.conv2d(x.double(), wt.double(), None, padding=1) * (m > 0)
    assert rel_err(nchw(ym), ref2)[0] < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 9, 11, 8, 32), (2, 17, 35, 16, 64), (1, 33, 70, 64, 128), (1, 8, 8, 24, 16), (1, 20, 24, 96, 64), (1, 36, 40, 128, 256)])
def test_conv3x3_bf16io_bf16_input_and_copy(shape):
    """bf16 activations in HBM: same numbers as the fp32-input kernel fed with the same (bf16-representable) values,
    for every tile built for bf16 input; the bf16 output copy is the RNE rounding of the fp32 output"""
    from osvos_pytorch_amd._lib import F32_BF16MFMA
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000 + 29)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16().float()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).bfloat16().float()
    b = torch.randn(cout, generator=g)
    m = torch.randn(n, cout, h, w, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1)) * (m > 0)
    wpk = ops.pack_fwd(wt.cuda(), F32_BF16MFMA)
    xg = nhwc(x)
    tiles = ops.conv3x3_bf16io_tiles()
    assert 1 in tiles and 8 in tiles
    assert all(t in tiles for t in (30, 31, 32, 33, 34, 35, 36, 37))      # the LDS-DMA staged kernel (36, 37: resident filter, Cin = 64 only)
    for tile in tiles + [-1, 101, 108, 130, 132, 135]:
        if tile % 100 in (36, 37) and cin != 64:
            with pytest.raises(RuntimeError):
                ops.conv3x3_bf16io(xg.bfloat16(), wpk, b.cuda(), cout, tile=tile)
            continue
        if tile % 100 in (30, 31, 32, 33, 34, 35, 36, 37) and (cin % 16 != 0 or cout % 8 != 0):
            with pytest.raises(RuntimeError):
                ops.conv3x3_bf16io(xg.bfloat16(), wpk, b.cuda(), cout, tile=tile)
            continue
        y16, yb16 = ops.conv3x3_bf16io(xg.bfloat16(), wpk, b.cuda(), cout, relu=True, mask=nhwc(m), tile=tile)
        assert rel_err(nchw(y16), ref)[0] < 3e-5, (shape, tile)
        assert torch.equal(yb16, y16.bfloat16()), (shape, tile)
        if tile % 100 in (30, 31, 32, 33, 34, 35, 36, 37):  # DMA staging exists for bf16 activations only
            with pytest.raises(RuntimeError):
                ops.conv3x3_bf16io(xg, wpk, b.cuda(), cout, tile=tile)
            continue
        y32, yb32 = ops.conv3x3_bf16io(xg, wpk, b.cuda(), cout, relu=True, mask=nhwc(m), tile=tile)
        assert torch.equal(y16, y32), (shape, tile)                  # identical arithmetic, only the staging differs
        assert torch.equal(yb32, y32.bfloat16()), (shape, tile)
    with pytest.raises(RuntimeError):
        ops.conv3x3_bf16io(xg.bfloat16(), wpk, b.cuda(), cout, tile=12)     # no such tile


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 33, 70, 64, 128), (2, 24, 40, 128, 64), (1, 20, 24, 256, 256), (1, 16, 40, 16, 64)])
def test_f32x3_kernels_with_two_pieces_per_operand(shape):
    """Precision 'fp32x2' at op level (round 6): the f32x3 convolution (every production tile, forward + masked data gradient) and the f32x3 weight
    gradient with TWO bf16 pieces per operand -- three products ah*bh + ah*bm + am*bh.  Against float64 (F.conv2d, vgg_osvos.py:41,142-143 and its
    autograd): each operand is short of its fp32 value by < 2^-16 relative, so the result is within 2^-15 sum|a||b| of the truth (held at half of
    that), a rel-L2 of a few 1e-6 -- several times the three-piece kernel's error (the switch really takes the other path) and 1000x below bf16's."""
    from osvos_pytorch_amd._lib import F32_X3
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    m = torch.randn(n, cout, h, w, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    mag = F.conv2d(x.abs().double(), wt.abs().double(), b.abs().double(), padding=1)          # sum |a||b| per output
    dy = torch.randn(n, cout, h, w, generator=g)
    wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wref, padding=1).backward(dy.double())
    pk, pk3, dpk3 = ops.pack_fwd(wt.cuda()), ops.pack_x3(wt.cuda()), ops.pack_x3(wt.cuda(), dgrad=True)
    res = {}
    try:
        for pieces in (3, 2):
            ops.set_x3_pieces(pieces)
            errs = []
            for tile in (10, 12, 14, 16, -1):      # pre-split packs (what the network runs) and the fp32 pack split in the kernel: the same bits
                y3 = ops.conv3x3_x3(nhwc(x), pk3, b.cuda(), cout, relu=True, tile=tile)
                if tile >= 0:
                    assert torch.equal(y3, ops.conv3x3(nhwc(x), pk, b.cuda(), cout, relu=True, tile=200 + tile)), (shape, pieces, tile)
                y = nchw(y3).cpu().double()
                assert float(((y - ref).abs() / (mag * 2.0 ** -16 + 1e-30)).max()) <= 1.0, (shape, pieces, tile)
                errs.append(float((y - ref).norm() / ref.norm()))
            d = nchw(ops.conv3x3_x3(nhwc(dy), dpk3, None, cin, mask=None, tile=-1)).cpu().double() if cout % 16 == 0 else None
            res[pieces] = (max(errs), ops.conv3x3_wgrad(nhwc(x), nhwc(dy), cin, cout, dtype=F32_X3)[0].cpu().double(), d)
    finally:
        ops.set_x3_pieces(3)
    if res[2][2] is not None:      # data gradient (rotated pack): un-masked here, against conv_transpose2d
        dfull = F.conv_transpose2d(dy.double(), wt.double(), padding=1)
        d3, d2 = [float((res[k][2] - dfull).norm() / dfull.norm()) for k in (3, 2)]
        assert d3 < 2e-6 and d2 < 2e-5, (shape, d3, d2)
    e3, e2 = res[3][0], res[2][0]
    w3 = float((res[3][1] - wref.grad).norm() / wref.grad.norm())
    w2 = float((res[2][1] - wref.grad).norm() / wref.grad.norm())
    print("f32x3 pieces 3 | 2: conv rel-L2 %.1e | %.1e, wgrad rel-L2 %.1e | %.1e" % (e3, e2, w3, w2))
    assert e3 < 2e-6 and 4 * e3 < e2 < 2e-5, (shape, e3, e2)          # (measured 3.6e-7 .. 7.2e-7 and 4.4e-6)
    assert w3 < 5e-6 and w2 < 5e-5 and (w2 > 2 * w3 or cin < 64), (shape, w3, w2)      # (Cin = 16 takes the exact fp32 skinny weight gradient in both modes)


@pytest.mark.gpu
@pytest.mark.parametrize("spread", ["unit", "tiny", "huge", "lognormal", "sparse"])
@pytest.mark.parametrize("shape", [(1, 33, 70, 64, 128), (2, 24, 40, 128, 64), (1, 20, 24, 256, 256), (1, 18, 40, 512, 512), (1, 40, 64, 128, 16)])
def test_f32x3_kernels_with_fp16_pairs(shape, spread):
    """Precision 'fp32h2' at op level (round 6; csrc/h2split.h): the f32x3 convolution (every production tile; forward with bias + ReLU, and the
    data-gradient pack) and the f32x3 weight gradient with TWO FP16 pieces per operand under block exponents -- three products on
    v_mfma_f32_32x32x16_f16.  Against float64 (F.conv2d, vgg_osvos.py:41,142-143, and its autograd), NEXT TO the exact fp32 MFMA kernel on the same
    inputs: every output within 2^-19 sum|a||b| of the truth, and the rel-L2 error no larger than the exact fp32 kernel's or 3.5e-7 (three fp32 roundings
    of the result; 5e-7 for the K = 4608 data gradient; measured: dense operands 1.3-1.9e-7 against the exact kernel's 2.0-3.0e-7 -- sixteen products enter one rounding instead of two --, 95 %
    zeros 1.6-2.9e-7 against 1.0-1.9e-7).  Operand magnitudes are swept over what the block exponents must absorb: ~1e-9 (gradients), ~1e+6, log-normal
    magnitudes across pixels (5 decades inside one tile: values below 2^-17 of their tile's maximum keep an ABSOLUTE error of 2^-40 of that maximum,
    which is where the per-output bound is 2^-19 and not 2^-21), and 95 % exact zeros (post-ReLU maps)."""
    from osvos_pytorch_amd._lib import F32_X3
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(sum(shape) + len(spread))
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    if spread == "tiny":
        x, dy, wt = x * 3e-9, dy * 7e-10, wt * 1e-3
    elif spread == "huge":
        x, dy, wt = x * 2.5e6, dy * 4e5, wt * 30.0
    elif spread == "lognormal":
        x = x * torch.exp(2.5 * torch.randn(n, 1, h, w, generator=g))
        dy = dy * torch.exp(2.5 * torch.randn(n, 1, h, w, generator=g)) * 1e-6
        wt = wt * torch.exp(1.5 * torch.randn(cout, cin, 1, 1, generator=g))
    elif spread == "sparse":
        x = x * (torch.rand(n, cin, h, w, generator=g) > 0.95)
        dy = dy * (torch.rand(n, cout, h, w, generator=g) > 0.95)
    b = torch.randn(cout, generator=g) * float(x.abs().mean() * wt.abs().mean() * cin)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    mag = F.conv2d(x.abs().double(), wt.abs().double(), b.abs().double(), padding=1)
    wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wref, padding=1).backward(dy.double())
    wmag = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.abs().double(), wmag, padding=1).backward(dy.abs().double())
    dfull = F.conv_transpose2d(dy.double(), wt.double(), padding=1)
    dmag = F.conv_transpose2d(dy.abs().double(), wt.abs().double(), padding=1)
    rel = lambda a, r: float((a.cpu().double() - r).norm() / r.norm())
    xg, dyg, bg = nhwc(x), nhwc(dy), b.cuda()
    # the exact fp32 MFMA kernels on the same inputs
    pk, dpk = ops.pack_fwd(wt.cuda()), ops.pack_dgrad(wt.cuda())
    e_exact = rel(nchw(ops.conv3x3(xg, pk, bg, cout, relu=True)), ref)
    d_exact = rel(nchw(ops.conv3x3(dyg, dpk, None, cin)), dfull) if cout % 16 == 0 and cin % 4 == 0 else None
    w_exact = rel(ops.conv3x3_wgrad(xg, dyg, cin, cout)[0], wref.grad)
    try:
        ops.set_x3_pieces(22)
        pk3, dpk3 = ops.pack_x3(wt.cuda()), (ops.pack_x3(wt.cuda(), dgrad=True) if cout % 16 == 0 else None)
        errs = []
        for tile in ((15, -1) if cout <= 32 else (10, 11, 12, 13, 14, 16, 17, -1)):
            y = nchw(ops.conv3x3_x3(xg, pk3, bg, cout, relu=True, tile=tile)).cpu().double()
            assert torch.isfinite(y).all(), (shape, spread, tile)
            assert float(((y - ref).abs() / (mag * 2.0 ** -19 + 1e-300)).max()) <= 1.0, (shape, spread, tile)
            errs.append(float((y - ref).norm() / ref.norm()))
        e_h2 = max(errs)
        if dpk3 is not None:
            d = nchw(ops.conv3x3_x3(dyg, dpk3, None, cin, tile=-1)).cpu().double()
            assert float(((d - dfull).abs() / (dmag * 2.0 ** -19 + 1e-300)).max()) <= 1.0, (shape, spread)
            d_h2 = rel(d, dfull)
        gw = ops.conv3x3_wgrad(xg, dyg, cin, cout, dtype=F32_X3)[0].cpu().double()
        assert float(((gw - wref.grad).abs() / (wmag.grad * 2.0 ** -19 + 1e-300)).max()) <= 1.0, (shape, spread)
        w_h2 = rel(gw, wref.grad)
        with pytest.raises(RuntimeError):                     # no fp16-pair form of the four-wave tiles
            ops.conv3x3_x3(xg, pk3, bg, cout, relu=True, tile=3)
    finally:
        ops.set_x3_pieces(3)
    print("h2 vs exact fp32 (rel-L2 against float64) %s %s: conv %.2e | %.2e, dgrad %s, wgrad %.2e | %.2e" %
          (shape, spread, e_h2, e_exact, "%.2e | %.2e" % (d_h2, d_exact) if dpk3 is not None and d_exact is not None else "-", w_h2, w_exact))
    assert e_h2 <= max(e_exact, 3.5e-7), (shape, spread, e_h2, e_exact)
    if dpk3 is not None and d_exact is not None:
        assert d_h2 <= max(d_exact, 5e-7), (shape, spread, d_h2, d_exact)
    assert w_h2 <= max(w_exact, 3e-7), (shape, spread, w_h2, w_exact)


def _bits_of(t_nhwc):
    """[N,H,W,C] -> int64 [N,H,W,C/32]: bit b of word g = (t[..., 32 g + b] > 0) (csrc/maskbits.h)"""
    n, h, w, c = t_nhwc.shape
    pos = (t_nhwc > 0).long().reshape(n, h, w, c // 32, 32)
    return (pos << torch.arange(32)).sum(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 40, 70, 64, 64), (1, 33, 45, 64, 128), (3, 16, 32, 64, 64), (1, 7, 9, 64, 64), (6, 100, 140, 64, 64), (1, 36, 40, 128, 64),
                                   (2, 17, 35, 32, 64)])
def test_conv3x3_bf16act_fused_epilogues_every_tile(shape):
    """The bf16-store trunk convolution as the network launches it, on EVERY tile built for bf16 activations -- among them the round-6
    resident-filter persistent forms 36 / 37 (Cin = 64: conv1_2, conv2_1, conv1_2's data gradient) and the LDS-DMA tiles'
    new pool-code epilogue: result vs float64 on bf16-representable operands (RNE of a value within fp32 summation noise of the truth), the
    sign-bit words, the fused 2x2 ceil-mode pool and its code bytes EXACTLY what the separate pooling kernel makes of the same result, and
    the data-gradient form (one-bit ReLU mask in, no ReLU).  Reference: vgg_osvos.py:136-145 (conv, ReLU, MaxPool2d ceil_mode) and autograd."""
    ops = _ops()
    from osvos_pytorch_amd._lib import F32_BF16MFMA
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).bfloat16().float()
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    wpk = ops.pack_fwd(wt.cuda(), F32_BF16MFMA)
    xg = nhwc(x.float()).bfloat16()
    m = torch.randn(n, cout, h, w, generator=g)
    mbits = _bits_of(nhwc(m).cpu()).to(torch.int32).cuda()         # (values >= 2^31 wrap to the same 32 bits)
    ref_masked = F.conv2d(x.double(), wt.double(), None, padding=1) * (m > 0)
    tiles = [t for t in ops.conv3x3_bf16io_tiles() if (t < 36 or cin == 64) and (t < 30 or (cin % 16 == 0 and cout % 8 == 0))]
    assert (36 in tiles) == (cin == 64)
    # 38 / 138: the round-6 Cin = 64 kernel with the deferred + skewed packed epilogue (conv3x3_bf16_p64.hip): sign bits and the pool in separate calls
    for tile in tiles + [-1] + ([136, 137, 38, 138] if cin == 64 else [130, 135]):
        y, bits, _, _ = ops.conv3x3_bf16act_fused(xg, wpk, b.cuda(), cout, relu=True, want_bits=True, tile=tile)
        yf = nchw(y.float())
        err = (yf.double().cpu() - ref).abs()
        assert float((err / (ref.abs() * 2.0 ** -8 + 1e-3)).max()) <= 1.0, (shape, tile, float(err.max()))      # within bf16 rounding of the truth
        assert torch.equal(_bits_of(y.float().cpu()), bits.cpu().long() & 0xFFFFFFFF), (shape, tile)
        y1, _, pooled, code = ops.conv3x3_bf16act_fused(xg, wpk, b.cuda(), cout, relu=True, want_pool=True, tile=tile)
        assert torch.equal(y, y1), (shape, tile)
        p_ref, c_ref = ops.maxpool2x2_bf16act_code(y)
        assert torch.equal(pooled, p_ref), (shape, tile)
        assert torch.equal(code, c_ref), (shape, tile)
        # plain (an op-level caller without the extras): same result bits
        y2, _, _, _ = ops.conv3x3_bf16act_fused(xg, wpk, b.cuda(), cout, relu=True, tile=tile)
        assert torch.equal(y, y2), (shape, tile)
        if tile % 100 != 38:      # sign bits AND the pool from one launch (the other tiles' epilogue takes any combination)
            y3, bits3, pooled3, code3 = ops.conv3x3_bf16act_fused(xg, wpk, b.cuda(), cout, relu=True, want_bits=True, want_pool=True, tile=tile)
            assert torch.equal(y, y3) and torch.equal(bits, bits3) and torch.equal(pooled, pooled3) and torch.equal(code, code3), (shape, tile)
        # data-gradient form: one-bit mask, no ReLU, no bias
        d, _, _, _ = ops.conv3x3_bf16act_fused(xg, wpk, None, cout, relu=False, mask_bits=mbits, tile=tile)
        derr = (nchw(d.float()).double().cpu() - ref_masked).abs()
        assert float((derr / (ref_masked.abs() * 2.0 ** -8 + 1e-3)).max()) <= 1.0, (shape, tile, float(derr.max()))
        assert bool(((nchw(d.float()).cpu() != 0) <= (m > 0)).all()), (shape, tile)
        if tile == 8:
            d8, y8 = d, y
    if cin == 64:      # the same arithmetic (fp32 bias add last, RNE, ReLU): the new kernel's bits equal the workhorse tile's up to fp32 summation order
        d38, _, _, _ = ops.conv3x3_bf16act_fused(xg, wpk, None, cout, relu=False, mask_bits=mbits, tile=38)
        y38, _, _, _ = ops.conv3x3_bf16act_fused(xg, wpk, b.cuda(), cout, relu=True, tile=38)
        assert float((d38.float() - d8.float()).abs().max()) <= 2.0 ** -7 * float(d8.float().abs().max()), shape
        assert float((y38 != y8).float().mean()) < 0.02, shape      # (a different k-order moves a value across a rounding boundary now and then)
    if cin != 64:
        with pytest.raises(RuntimeError):
            ops.conv3x3_bf16act_fused(xg, wpk, b.cuda(), cout, relu=True, tile=36)


@pytest.mark.gpu
def test_bf16copy_outputs_of_pool_and_layout_kernels():
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 16, 9, 13, generator=g)
    xg = nhwc(x)
    y, yb = ops.maxpool2x2_bf16copy(xg)
    assert torch.equal(y, ops.maxpool2x2(xg)) and torch.equal(yb, y.bfloat16())
    dy = torch.randn(2, 16, 5, 7, generator=g)
    ds = torch.randn(2, 16, 9, 13, generator=g)
    dx, dxb = ops.maxpool2x2_bwd_bf16copy(xg, nhwc(dy), nhwc(ds))
    assert torch.equal(dx, ops.maxpool2x2_bwd(xg, nhwc(dy), nhwc(ds))) and torch.equal(dxb, dx.bfloat16())
    img = torch.randn(2, 3, 7, 5, generator=g).cuda()
    a, ab = ops.nchw_to_nhwc_bf16copy(img, 8)
    assert torch.equal(a, ops.nchw_to_nhwc(img, 8)) and torch.equal(ab, a.bfloat16())


@pytest.mark.gpu
def test_fused_sgd_matches_torch_optim_sgd_and_shares_state_dict():
    """8 parameter groups as the reference builds them (train_online.py:79-88): bit-identical trajectory to
    torch.optim.SGD over several steps, interchangeable state_dict, version counters bumped"""
    from osvos_pytorch_amd.optim import FusedSGD
    g = torch.Generator().manual_seed(41)
    shapes = [(64, 3, 3, 3), (64,), (128, 64, 3, 3), (128,), (16, 128, 3, 3), (16,), (1, 16, 1, 1), (1,), (1, 64, 1, 1), (1,), (5000,)]
    lr, wd = 1e-3, 2e-4
    def groups(ps):
        return [{"params": ps[0:4:2], "weight_decay": wd, "lr": lr}, {"params": ps[1:4:2], "lr": 2 * lr},
                {"params": ps[4:6], "weight_decay": wd, "lr": lr}, {"params": ps[6:8], "lr": lr / 10, "weight_decay": wd},
                {"params": ps[8:10], "lr": lr / 100}, {"params": ps[10:], "lr": 0.0}]
    init = [torch.randn(s, generator=g) for s in shapes]
    pa = [t.clone().cuda().requires_grad_() for t in init]
    pb = [t.clone().cuda().requires_grad_() for t in init]
    oa = torch.optim.SGD(groups(pa), lr=lr, momentum=0.9)
    ob = FusedSGD(groups(pb), lr=lr, momentum=0.9)
    for step in range(4):
        if step == 2:          # swap optimizer state through state_dict mid-run
            ob2 = FusedSGD(groups(pb), lr=lr, momentum=0.9)
            import copy
            ob2.load_state_dict(copy.deepcopy(oa.state_dict()))      # (load_state_dict aliases tensors that need no cast)
            for x, y in zip(pa, pb):
                assert torch.equal(oa.state[x]["momentum_buffer"], ob2.state[y]["momentum_buffer"])
            ob = ob2
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).cuda()
            x.grad = gr.clone() if not (step == 1 and x.numel() == 16) else None      # a tensor without gradient is skipped
            y.grad = None if x.grad is None else gr.clone()
        v0 = [y._version for y in pb]
        oa.step(); ob.step()
        for i, (x, y) in enumerate(zip(pa, pb)):
            assert torch.equal(x.detach(), y.detach()), (step, i)
            if y.grad is not None:
                assert y._version > v0[i]
    assert set(ob.state_dict()["state"][0].keys()) == {"momentum_buffer"}
    with pytest.raises(ValueError):
        FusedSGD(groups(pb), lr=lr, momentum=0.9, nesterov=True)
    with pytest.raises(RuntimeError):
        cpu = [torch.zeros(3, requires_grad=True)]
        o = FusedSGD(cpu, lr=0.1)
        cpu[0].grad = torch.ones(3)
        o.step()


@pytest.mark.gpu
def test_bf16act_pooling_matches_fp32_rule_on_bf16_values():
    """bf16 tensors in, bf16 out: forward = same elements the fp32 kernel picks; backward = fp32 arithmetic of the
    fp32 kernel on the same (bf16-representable) values, rounded once"""
    ops = _ops()
    g = torch.Generator().manual_seed(43)
    for shape in [(2, 16, 9, 13), (1, 64, 8, 8), (1, 8, 1, 1), (1, 24, 7, 2)]:
        n, c, h, w = shape
        x = F.relu(torch.randn(shape, generator=g)).bfloat16()
        xg = nhwc(x.float()).bfloat16()
        y = ops.maxpool2x2_bf16act(xg)
        assert torch.equal(y.float(), ops.maxpool2x2(xg.float()))
        dy = torch.randn(n, c, (h + 1) // 2, (w + 1) // 2, generator=g).bfloat16()
        ds = torch.randn(shape, generator=g).bfloat16()
        dx = ops.maxpool2x2_bwd_bf16act(xg, nhwc(dy.float()).bfloat16(), nhwc(ds.float()).bfloat16())
        ref = ops.maxpool2x2_bwd(xg.float(), nhwc(dy.float()), nhwc(ds.float()))
        assert torch.equal(dx, ref.bfloat16())
        dx0 = ops.maxpool2x2_bwd_bf16act(xg, nhwc(dy.float()).bfloat16())
        assert torch.equal(dx0, ops.maxpool2x2_bwd(xg.float(), nhwc(dy.float())).bfloat16())


@pytest.mark.gpu
def test_conv3x3_bf16io_bf16_mask_and_bf16_only_output():
    from osvos_pytorch_amd._lib import F32_BF16MFMA
    ops = _ops()
    g = torch.Generator().manual_seed(47)
    n, h, w, cin, cout = 2, 19, 27, 64, 64
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 24
    m = F.relu(torch.randn(n, cout, h, w, generator=g)).bfloat16()           # a post-ReLU activation as the mask
    wpk = ops.pack_dgrad(wt.cuda(), F32_BF16MFMA) if False else ops.pack_fwd(wt.cuda(), F32_BF16MFMA)
    xg, mg = nhwc(x.float()).bfloat16(), nhwc(m.float()).bfloat16()
    for tile in (-1, 1, 8):
        y_ref, yb_ref = ops.conv3x3_bf16io(xg, wpk, None, cout, mask=mg.float(), tile=tile)   # (accumulation order differs between tiles)
        y, yb = ops.conv3x3_bf16io(xg, wpk, None, cout, mask=mg, tile=tile)                   # bf16 mask
        assert torch.equal(y, y_ref) and torch.equal(yb, yb_ref)
        y2, yb2 = ops.conv3x3_bf16io(xg, wpk, None, cout, mask=mg, tile=tile, want_f32=False)   # bf16-only result
        assert y2 is None and torch.equal(yb2, yb_ref)
    ref = F.conv2d(x.double(), wt.bfloat16().double(), None, padding=1) * (m > 0)
    assert rel_err(nchw(y_ref), ref)[0] < 3e-5
    with pytest.raises(RuntimeError):
        ops.conv3x3_bf16io(xg, wpk, None, cout, want_f32=False, want_bf16=False)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 9, 11, 64, 64), (2, 17, 35, 64, 128), (1, 33, 70, 128, 64), (3, 8, 40, 64, 64)])
def test_wgrad_bf16act_equals_fp32_input_kernel_on_bf16_values(shape):
    """bf16 x and dy in HBM: the weight gradient is bit-identical to the bf16-MFMA kernel fed with the same values as fp32
    (same products, same accumulation order); the bias gradient is the fp32 sum of the bf16 dy"""
    from osvos_pytorch_amd._lib import F32_BF16MFMA
    ops = _ops()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(53)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16()
    dy = torch.randn(n, cout, h, w, generator=g).bfloat16()
    xg, dyg = nhwc(x.float()), nhwc(dy.float())
    dw32, db32 = ops.conv3x3_wgrad(xg, dyg, cin, cout, dtype=F32_BF16MFMA)
    dw16, db16 = ops.conv3x3_wgrad_bf16act(xg.bfloat16(), dyg.bfloat16(), cin, cout)
    assert torch.equal(dw16, dw32)
    assert rel_err(db16.cpu(), dy.double().sum((0, 2, 3)))[0] < 1e-5
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, None, padding=1).backward(dy.double())
    assert rel_err(dw16.cpu(), wt.grad)[0] < 3e-5


@pytest.mark.gpu
def test_bf16_matrix_pipe_probe_runs_and_depends_on_the_operand_bits():
    """osvos_debug_mfma_peak_bf16 (what bench.py's roofline.pipe_sustained times): the register-only MFMA loop returns the exact sums
    (zeros stay zero; constant operands give iters x 8 accumulators x 16 k x a x b in every accumulator element) and sustains a
    plausible rate -- above 0.8 PFLOP/s on any operands, and not slower on zeros than on noise (the clock follows the toggling bits)."""
    import ctypes as C
    from osvos_pytorch_amd import _lib
    blocks, iters = 1024, 500
    out = torch.empty(blocks * 512, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(seed_vals, reps=1):
        seed = seed_vals.to(torch.bfloat16).cuda()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.check(_lib.lib().osvos_debug_mfma_peak_bf16(C.c_void_p(seed.data_ptr()), C.c_void_p(out.data_ptr()), blocks, iters, st))
        e0.record()
        for _ in range(reps):
            _lib.check(_lib.lib().osvos_debug_mfma_peak_bf16(C.c_void_p(seed.data_ptr()), C.c_void_p(out.data_ptr()), blocks, iters, st))
        e1.record()
        e1.synchronize()
        return reps * blocks * 8 * iters * 8 * 2.0 * 32 * 32 * 16 / (e0.elapsed_time(e1) * 1e-3) / 1e12

    tz = run(torch.zeros(128 * 8), reps=4)
    assert float(out.abs().max()) == 0.0
    run(torch.full((128 * 8,), 0.5))
    # every accumulator element: sum over iters of 16 k-products of 0.5 * 0.5; a thread sums its 8 accumulators x 16 registers
    assert torch.all(out == iters * 16 * 0.25 * 8 * 16)
    tn = run(torch.rand(128 * 8, generator=torch.Generator().manual_seed(3)) - 0.5, reps=4)
    print("bf16 MFMA-only loop: %.0f TFLOP/s on zeros, %.0f on noise" % (tz, tn))
    assert tz > 800 and tn > 800 and tz > 0.95 * tn


def test_lds_dma_layout_probe():
    """buffer_load_dwordx4 ... lds: lane l of a wave instruction lands in LDS slot (M0 base)/16 + l; lanes whose buffer offset
    is out of range land as zeros -- the two facts the DMA-staged convolution kernel is built on"""
    import ctypes as C
    from osvos_pytorch_amd import _lib
    n = 97
    src = torch.arange(n * 4, dtype=torch.float32).cuda()
    out = torch.empty(8 * 64 * 4, device="cuda")
    _lib.check(_lib.lib().osvos_debug_lds_dma(C.c_void_p(src.data_ptr()), n, C.c_void_p(out.data_ptr()), None), "lds dma")
    got = out.cpu().view(8, 64, 4)
    for w in range(4):
        for i in range(2):
            for l in range(64):
                exp = torch.zeros(4) if l == 5 else src.cpu().view(n, 4)[(l * 7 + 3 * w + i) % n]
                assert torch.equal(got[2 * w + i, l], exp), (w, i, l, got[2 * w + i, l], exp)


def _bytescale_scipy11(data):
    """scipy 1.1 misc.bytescale(data, cmin=None, cmax=None, high=255, low=0) as toimage() calls it for mode 'L' (float32 input)"""
    cmin, cmax = data.min(), data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = np.float32(float(255 - 0) / float(cscale))
    bytedata = (data - cmin) * scale + np.float32(0)
    return (bytedata.clip(0, 255) + np.float32(0.5)).astype(np.uint8)


@pytest.mark.gpu
def test_result_writer_bytes_png_and_jaccard(tmp_path):
    """train_online.py:181-187 restated: sigmoid -> scipy<=1.1 imsave.  Device bytes == numpy restatement except where float32
    exp differs in the last bit right at a rounding boundary (<= 1 grey level, < 0.1 % of pixels); PNG decodes to the same bytes;
    J counts are exact integers"""
    from PIL import Image
    from osvos_pytorch_amd import results
    g = torch.Generator().manual_seed(61)
    logits = (torch.randn(3, 1, 37, 53, generator=g) * 3 - 1)
    logits[2] = 0.25                                            # constant frame: cscale == 0 branch
    got = results.mask_bytes(logits.cuda()).cpu().numpy()
    for n in range(3):
        pred = np.squeeze(1 / (1 + np.exp(-logits[n].numpy().transpose(1, 2, 0))))       # the reference's float32 numpy sigmoid
        exp = _bytescale_scipy11(pred)
        d = np.abs(got[n].astype(np.int32) - exp.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3, (n, d.max(), (d != 0).mean())
    p = str(tmp_path / "m.png")
    results.write_png(p, got[0])
    assert np.array_equal(np.array(Image.open(p)), got[0]) and Image.open(p).mode == "L"
    results.save_masks(logits.cuda(), [str(tmp_path / ("f%d.png" % i)) for i in range(3)])
    assert np.array_equal(np.array(Image.open(str(tmp_path / "f1.png"))), got[1])
    gt = (torch.rand(3, 1, 37, 53, generator=g) > 0.6).float()
    gt[2] = 0
    js = results.jaccard(logits.cuda(), gt.cuda())
    for n in range(3):
        pm, gm = logits[n].numpy() > 0, gt[n].numpy() > 0.5
        u = np.logical_or(pm, gm).sum()
        assert js[n] == (1.0 if u == 0 else np.logical_and(pm, gm).sum() / u)
    assert results.jaccard(torch.full((1, 1, 4, 4), -5.0).cuda(), torch.zeros(1, 1, 4, 4).cuda()) == [1.0]      # both empty
    st = results.davis_statistics([0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2])
    assert abs(st["mean"] - 0.55) < 1e-12 and abs(st["recall"] - 0.5) < 1e-12 and abs(st["decay"] - 0.6) < 1e-12
    with pytest.raises(RuntimeError):
        results.mask_bytes(logits)


@pytest.mark.gpu
def test_wgrad_bf16_forms_are_bit_identical(tmp_path):
    """The two forms of the bf16 weight gradient that ship (OSVOS_WGRAD_FORM 0: first staging form, four waves / 64-cout tiles; 3, the default:
    pixel-major tiles read with ds_read_b64_tr_b16, eight waves / 128-cout tiles where Cout allows): same patches, same k-order -> the weight
    gradient must be bit-identical, the bias gradient equal up to fp32 summation order.  (The retired forms 1, 2, 4, 5 live in
    tools/native/wgrad_bf16_forms.inc and are built into the probe harness only.)  Shapes cover ragged right / bottom edges, one and two
    128-cout tiles, 64-cout tiles and several images.  The switch is read once per process: every form runs in a subprocess."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        import osvos_pytorch_amd.ops as ops
        res = {}
        for n, h, w, cin, cout in [(1, 9, 11, 64, 64), (2, 17, 35, 64, 128), (1, 33, 70, 128, 64), (3, 8, 40, 64, 64), (2, 30, 54, 256, 256), (1, 60, 107, 128, 256)]:
            g = torch.Generator().manual_seed(1000 + h)
            x = torch.randn(n, h, w, cin, generator=g).bfloat16().cuda()
            dy = torch.randn(n, h, w, cout, generator=g).bfloat16().cuda()
            dw, db = ops.conv3x3_wgrad_bf16act(x, dy, cin, cout)
            res["dw_%%dx%%dx%%d_%%d_%%d" %% (n, h, w, cin, cout)] = dw.cpu().numpy()
            res["db_%%dx%%dx%%d_%%d_%%d" %% (n, h, w, cin, cout)] = db.cpu().numpy()
        np.savez(sys.argv[1], **res)
    ''') % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    got = {}
    for form in ("3", "0"):
        out = str(tmp_path / ("f%s.npz" % form))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, OSVOS_WGRAD_FORM=form), timeout=600)
        got[form] = dict(np.load(out))
    for form in ("0",):
        for k, ref in got["3"].items():
            if k.startswith("dw_"):
                assert np.array_equal(got[form][k], ref), (form, k)
            else:
                assert np.allclose(got[form][k], ref, rtol=1e-5, atol=1e-4), (form, k)


@pytest.mark.parametrize("HW", [(480, 854), (1080, 1920), (37, 53), (101, 135)])
def test_head_lowres_upsample_and_backward_at_the_baseline_crop_sets(HW):
    """Op-level check of the commuted head at the 854x480 crop set ((1,1,1,1),(3,3,2,2),(5,5,4,4),(13,13,8,8)), the 1080p one
    ((1,1,1,1),(2,2,2,2),(4,4,4,4),(8,8,12,12)) and two odd sizes, against the reference's ops in float64: score_dsn 1x1 conv,
    ConvTranspose2d(k = 2s, stride s) of the 1- and the 16-channel maps, center_crop, cat + fuse (vgg_osvos.py:68-72,
    osvos_layers.py:51-56) -- and their adjoints."""
    ops = _ops()
    H, W = HW
    g = torch.Generator().manual_seed(H)
    n = 1 if H >= 480 else 2
    hs, ws = [], []
    h, w = H, W
    for i in range(4):
        h, w = (h + 1) // 2, (w + 1) // 2
        hs.append(h)
        ws.append(w)
    preps = [torch.randn(n, 16, hs[i], ws[i], generator=g, dtype=torch.float64, requires_grad=True) for i in range(4)]
    wd = [torch.randn(1, 16, 1, 1, generator=g, dtype=torch.float64, requires_grad=True) for _ in range(4)]
    bd = [torch.randn(1, generator=g, dtype=torch.float64, requires_grad=True) for _ in range(4)]
    wf = torch.randn(1, 64, 1, 1, generator=g, dtype=torch.float64, requires_grad=True)
    bf = torch.randn(1, generator=g, dtype=torch.float64)

    def filt(k):          # upsample_filt (osvos_layers.py:59-67)
        f = (k + 1) // 2
        c = f - 1 if k % 2 == 1 else f - 0.5
        r = 1 - (torch.arange(k, dtype=torch.float64) - c).abs() / f
        return r[:, None] * r[None, :]

    def crop(t):          # center_crop (osvos_layers.py:51-56): floor(excess / 2) dropped at the top / left
        eh, ew = t.shape[2] - H, t.shape[3] - W
        return t[:, :, eh // 2:eh // 2 + H, ew // 2:ew // 2 + W]
    f1 = [filt(4 << i) * (1.0 + 0.1 * i) for i in range(4)]          # upscale_[i]: arbitrary (here scaled bilinear) 1 -> 1 filter
    f16 = [filt(4 << i) for i in range(4)]
    sides, side_outs = [], []
    for i in range(4):
        s = 2 << i
        w16 = torch.zeros(16, 16, 2 * s, 2 * s, dtype=torch.float64)
        for c in range(16):
            w16[c, c] = f16[i]
        sides.append(crop(F.conv_transpose2d(preps[i], w16, stride=s)))
        side_outs.append(crop(F.conv_transpose2d(F.conv2d(preps[i], wd[i], bd[i]), f1[i][None, None], stride=s)))
    fused = F.conv2d(torch.cat(sides, 1), wf, bf)
    dsides = [torch.randn(n, 1, H, W, generator=g, dtype=torch.float64) for _ in range(4)]
    dfused = torch.randn(n, 1, H, W, generator=g, dtype=torch.float64)
    (sum((o * d).sum() for o, d in zip(side_outs, dsides)) + (fused * dfused).sum()).backward()

    cu = lambda t: t.detach().float().cuda().contiguous()      # noqa: E731
    scores, fparts = [], []
    for i in range(4):
        sc, fp = ops.head_lowres(nhwc(preps[i].detach()), cu(wd[i].flatten()), cu(bd[i]), cu(wf.flatten()[16 * i:16 * i + 16]))
        ref_sc = F.conv2d(preps[i].detach(), wd[i].detach(), bd[i].detach())[:, 0]
        assert rel_err(sc.cpu(), ref_sc)[0] < 2e-6
        scores.append(sc)
        fparts.append(fp)
    outs = ops.head_upsample(scores, fparts, [cu(f.flatten()) for f in f1], [cu(f.flatten()) for f in f16], cu(bf), H, W)
    for i in range(4):
        assert rel_err(outs[i].cpu(), side_outs[i].detach())[0] < 3e-6, (HW, i)
        # the border rows / columns are where a wrong crop offset shows
        assert rel_err(outs[i].cpu()[..., :3, :], side_outs[i].detach()[..., :3, :])[0] < 3e-6 and rel_err(outs[i].cpu()[..., -3:], side_outs[i].detach()[..., -3:])[0] < 3e-6
    assert rel_err(outs[4].cpu(), fused.detach())[0] < 3e-6, HW
    for i in range(4):
        dprep, dwf, dwd, dbd = ops.head_bwd(nhwc(preps[i].detach()), cu(dsides[i]), cu(dfused), cu(f1[i].flatten()), cu(f16[i].flatten()),
                                            cu(wd[i].flatten()), cu(wf.flatten()[16 * i:16 * i + 16]), H, W, i)
        assert rel_err(nchw(dprep), preps[i].grad)[0] < 1e-5, (HW, i)
        assert rel_err(dwf.cpu(), wf.grad.flatten()[16 * i:16 * i + 16])[0] < 1e-5
        assert rel_err(dwd.cpu(), wd[i].grad.flatten())[0] < 1e-5
        assert abs(float(dbd) - float(bd[i].grad)) <= 1e-5 * abs(float(bd[i].grad)) + 1e-3


@pytest.mark.parametrize("shape", [(1, 9, 11, 64), (2, 17, 35, 64), (1, 40, 70, 16), (1, 8, 33, 32)])
def test_input_gradient_of_the_first_convolution(shape):
    """dgrad_c3.hip: dx = conv_transpose(dy, W) for Cin = 3 straight into NCHW, against float64"""
    ops = _ops()
    n, h, w, cout = shape
    g = torch.Generator().manual_seed(91 + w)
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.randn(cout, 3, 3, 3, generator=g) / 5
    ref = F.conv_transpose2d(dy.double(), wt.double(), padding=1)
    dx = ops.conv3x3_dgrad_c3(nhwc(dy), wt.cuda())
    assert dx.shape == (n, 3, h, w)
    emax, el2 = rel_err(dx.cpu(), ref)
    assert emax < 2e-5 and el2 < 1e-5, (shape, emax, el2)


@pytest.mark.gpu
def test_cbce_step_multi_equals_the_single_head_calls():
    """osvos_cbce_step_multi (the parent loop's five losses in three launches, class counts formed once) against five osvos_cbce_step calls: the
    per-head arithmetic is the same kernel body, so gradients are bit-identical and losses / running sums agree to the last float32 bit or
    two (the double partial sums meet in another order); odd element counts (scalar tail) and a batch."""
    from osvos_pytorch_amd.layers.osvos_layers import class_balanced_cross_entropy_loss_step as one
    from osvos_pytorch_amd.layers.osvos_layers import class_balanced_cross_entropy_loss_step_multi as multi
    g = torch.Generator(device="cuda").manual_seed(11)
    for (n, h, w) in [(1, 37, 53), (2, 64, 66), (3, 16, 17)]:
        outs = [torch.randn(n, 1, h, w, device="cuda", generator=g) * 3 - 1 for _ in range(5)]
        lab = (torch.rand(n, 1, h, w, device="cuda", generator=g) > 0.8).float()
        scales = [0.07, 0.07, 0.07, 0.07, 0.1]
        r1 = [torch.full((), 2.5, device="cuda") for _ in range(5)]
        r2 = [torch.full((), 2.5, device="cuda") for _ in range(5)]
        ref = [one(o, lab, size_average=False, grad_scale=s, running=r) for o, s, r in zip(outs, scales, r1)]
        losses, grads = multi(outs, lab, size_average=False, grad_scales=scales, running=r2)
        torch.cuda.synchronize()
        for k in range(5):
            assert torch.equal(grads[k], ref[k][1]), k
            assert abs(float(losses[k]) - float(ref[k][0])) <= 2e-7 * abs(float(ref[k][0])), k
            assert abs(float(r2[k]) - float(r1[k])) <= 2e-7 * abs(float(r1[k])), k


def test_cbce_scratch_is_left_zero_and_the_zeroed_promise_changes_nothing():
    """Round 6: the loss call is two launches -- the workgroup that arrives last at the sweep forms the losses (no final kernel) and clears the
    scratch, and with OSVOS_CBCE_SCRATCH_ZEROED the call enqueues no memset (osvos_layers.py:19-48 arithmetic unchanged).  (1) after ANY call the
    scratch reads zero; (2) forty back-to-back calls on ONE never-re-zeroed buffer with the promise == forty calls on freshly poisoned buffers
    without it: gradients bit-identical, losses within the fp64 summation order; (3) the training-loop wrapper (persistent buffer per stream)
    agrees with the C oracle over repeated calls, per-image and five-head forms included."""
    import ctypes as C
    from osvos_pytorch_amd import _lib
    from osvos_pytorch_amd.autograd import CBCE_PER_IMAGE, CBCE_SCRATCH_ZEROED
    from osvos_pytorch_amd.layers.osvos_layers import class_balanced_cross_entropy_loss_step_multi as multi
    from oracle import c_oracle
    l = _lib.lib()
    vp = C.c_void_p
    g = torch.Generator(device="cuda").manual_seed(5)

    def call(outs, lab, scratch, flags, n_img):
        n = len(outs)
        losses = torch.full((n,), -7.0, device="cuda")
        grads = [torch.empty_like(o) for o in outs]
        a_out = (vp * n)(*[vp(o.data_ptr()) for o in outs])
        a_loss = (vp * n)(*[vp(losses.data_ptr() + 4 * k) for k in range(n)])
        a_grad = (vp * n)(*[vp(x.data_ptr()) for x in grads])
        a_scale = (C.c_float * n)(*([0.2] * n))
        _lib.check(l.osvos_cbce_step_ex(a_out, vp(lab.data_ptr()), a_loss, a_grad, vp(scratch.data_ptr()), lab.numel(), n_img, 2, flags, None, n, a_scale, None,
                                        vp(torch.cuda.current_stream().cuda_stream)), "cbce")
        return losses, grads
    for (n_img, h, w, heads, per_image) in [(1, 48, 85, 1, False), (3, 40, 52, 5, False), (4, 24, 36, 1, True), (2, 33, 35, 5, True)]:
        fl = CBCE_PER_IMAGE if per_image else 0
        nb = int(l.osvos_cbce_scratch_bytes(heads, n_img, fl))
        persistent = torch.zeros(nb, device="cuda", dtype=torch.uint8)
        for it in range(40 if heads == 1 else 6):
            outs = [torch.randn(n_img, 1, h, w, device="cuda", generator=g) * 3 - 1 for _ in range(heads)]
            lab = (torch.rand(n_img, 1, h, w, device="cuda", generator=g) > 0.7).float()
            la, ga = call(outs, lab, persistent, fl | CBCE_SCRATCH_ZEROED, n_img)
            poisoned = torch.full((nb,), 0xA5, device="cuda", dtype=torch.uint8)
            lb, gb = call(outs, lab, poisoned, fl, n_img)
            torch.cuda.synchronize()
            assert int(persistent.count_nonzero()) == 0 and int(poisoned.count_nonzero()) == 0, (n_img, heads, it)
            for k in range(heads):
                assert torch.equal(ga[k], gb[k]), (heads, k, it)
                assert abs(float(la[k]) - float(lb[k])) <= 2e-7 * abs(float(lb[k])), (heads, k, it)
            if not per_image:
                ref, _ = c_oracle.cbce(outs[0].cpu().numpy().astype(np.float64).reshape(n_img, -1), lab.cpu().numpy().astype(np.float64).reshape(n_img, -1), 2)
                assert abs(float(la[0]) - ref) <= 1e-5 * abs(ref), (n_img, it)
    # the wrapper of the training loops: one persistent buffer per (device, stream, size); a side stream gets its own
    outs = [torch.randn(2, 1, 30, 40, device="cuda", generator=g) for _ in range(5)]
    lab = (torch.rand(2, 1, 30, 40, device="cuda", generator=g) > 0.6).float()
    first = multi(outs, lab, size_average=False, grad_scales=[1.0] * 5)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        other = multi(outs, lab, size_average=False, grad_scales=[1.0] * 5)
    torch.cuda.current_stream().wait_stream(side)
    again = multi(outs, lab, size_average=False, grad_scales=[1.0] * 5)
    torch.cuda.synchronize()
    for a, b in ((first, other), (first, again)):
        assert all(torch.equal(x, y) for x, y in zip(a[1], b[1])) and float((a[0] - b[0]).abs().max()) <= 2e-7 * float(a[0].abs().max())


@pytest.mark.parametrize("shape", [(1, 24, 32, 128), (2, 17, 21, 256), (1, 30, 54, 512), (1, 7, 5, 128)])
def test_skinny_side_prep_wgrad_on_the_bf16_pipe(shape):
    """side_prep's weight gradient (Cout = 16; vgg_osvos.py:41) -- the S16 form of wgrad_f32x3.hip, what the default fp32x3 network runs --
    against float64 and against the exact fp32 skinny kernel it replaced (0.17 of the fp32 roofline in round 2)"""
    from osvos_pytorch_amd import ops
    from osvos_pytorch_amd._lib import F32, F32_X3
    n, h, w, cin = shape
    cout = 16
    g = torch.Generator().manual_seed(51 + h)
    x = F.relu(torch.randn(n, cin, h, w, generator=g))
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, padding=1).backward(dy.double())
    xd, dyd = x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda()
    dw3, db3 = ops.conv3x3_wgrad(xd, dyd, cin, cout, dtype=F32_X3)
    d = dw3.cpu().double() - wt.grad
    emax, el2 = float(d.abs().max() / wt.grad.abs().max()), float(d.norm() / wt.grad.norm())
    assert el2 < 1e-6 and emax < 1e-5, (shape, emax, el2)
    torch.testing.assert_close(db3.cpu().double(), dy.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-4)
    dw, db = ops.conv3x3_wgrad(xd, dyd, cin, cout, dtype=F32)
    assert float((dw3.double() - dw.double()).norm() / dw.double().norm()) < 1e-6, shape


SK_CASES = [
    # N, H, W, Cin, Cout, tile, grid   (grid 0 = automatic; small forced grids cut every tile into several parts)
    (1, 37, 53, 64, 64, 12, 5),
    (1, 37, 53, 64, 64, 12, 7),
    (2, 30, 54, 512, 512, 12, 256),
    (2, 30, 54, 512, 512, 14, 256),
    (1, 61, 107, 256, 512, 14, 256),
    (1, 120, 214, 256, 256, 10, 256),
    (1, 120, 214, 256, 256, 10, 240),
    (1, 96, 160, 128, 128, 10, 13),
    (1, 30, 54, 512, 512, -1, 0),
    (1, 120, 214, 128, 256, -1, 0),
]


@pytest.mark.parametrize("case", SK_CASES)
def test_conv3x3_f32x3_streamk(case):
    """stream-K form of the f32x3 convolution (conv3x3_f32x3.hip): persistent workgroups walk equal shares of the (tile, K chunk) units,
    shared tiles are summed by the last arriver in contributor order.  Held to the float64 bars of the plain kernel, within fp32 summation
    order of it, DETERMINISTIC (two runs bit-equal, also under a different arrival order: another launch running beside it), the fused
    pool equal to the pooling kernel on the same y, ReLU mask / no-bias forms included."""
    ops = _ops()
    n, h, w, cin, cout, tile, grid = case
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    xg, pk3 = nhwc(x), ops.pack_x3(wt.cuda())
    y, pooled = ops.conv3x3_x3_streamk(xg, pk3, b.cuda(), cout, relu=True, tile=tile, grid=grid, want_pooled=True)
    emax, el2 = rel_err(nchw(y), ref)
    assert emax < 2e-5 and el2 < 1e-5, (case, emax, el2)
    plain = ops.conv3x3_x3(xg, pk3, b.cuda(), cout, relu=True, tile=tile)
    assert float((y - plain).abs().max()) <= 4e-6 * float(plain.abs().max()), case
    assert torch.equal(pooled, ops.maxpool2x2(y)), case
    # determinism: same bits again, and with a second stream keeping part of the chip busy (another arrival order)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(3):
            junk = junk @ junk * 1e-3
    y2 = ops.conv3x3_x3_streamk(xg, pk3, b.cuda(), cout, relu=True, tile=tile, grid=grid)
    torch.cuda.synchronize()
    assert torch.equal(y, y2), case
    y3 = ops.conv3x3_x3_streamk(xg, pk3, b.cuda(), cout, relu=True, tile=tile, grid=grid)
    assert torch.equal(y, y3), case
    # data-gradient form: rotated pack, ReLU mask of the producer, no bias
    dy = torch.randn(n, cout, h, w, generator=g)
    m = torch.randn(n, cin, h, w, generator=g)
    dpk3 = ops.pack_x3(wt.cuda(), dgrad=True)
    dx = ops.conv3x3_x3_streamk(nhwc(dy), dpk3, None, cin, mask=nhwc(m), tile=tile, grid=grid)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, wt.double(), dy.double(), padding=1) * (m > 0)
    assert rel_err(nchw(dx), ref_dx)[0] < 3e-5, case
    # the tickets are back at zero: the workspace is ready for the next launch
    ws = ops.streamk_workspace(xg.device)
    assert int(ws[:ops.lib().osvos_conv3x3_x3_streamk_ticket_bytes()].view(torch.int32).abs().max()) == 0


def test_cbce_per_image_counts_external_counts_and_offset_views():
    """osvos_cbce_step_ex: (1) per-image mode == the sequential single-image calls of an accumulation window (gradients bit-identical, the
    summed loss to fp32 round-off), incl. image sizes that are not multiples of four (element-wise sweep); (2) external counts: two shards
    with the GLOBAL counts give the whole batch's loss (sum) and gradients (concatenation), and equal the torch expression of
    parallel.cbce_with_counts on the CPU; (3) contiguous but 4-byte-offset views (used to be refused) give the numbers of their
    aligned copies."""
    from osvos_pytorch_amd.layers.osvos_layers import class_balanced_cross_entropy_loss_step as one
    from osvos_pytorch_amd.layers.osvos_layers import class_balanced_cross_entropy_loss_step_multi as multi
    from osvos_pytorch_amd.layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
    from osvos_pytorch_amd import parallel
    g = torch.Generator(device="cuda").manual_seed(23)
    for (n, h, w) in [(5, 37, 53), (4, 32, 48), (3, 7, 9)]:
        out = torch.randn(n, 1, h, w, device="cuda", generator=g) * 3 - 1
        lab = (torch.rand(n, 1, h, w, device="cuda", generator=g) > torch.linspace(0.5, 0.95, n, device="cuda").view(n, 1, 1, 1)).float()
        # (1) the window in one call vs image by image
        run_a, run_b = torch.zeros((), device="cuda"), torch.zeros((), device="cuda")
        seq = [one(out[j:j + 1].clone(), lab[j:j + 1].clone(), size_average=False, grad_scale=1.0 / n, running=run_a) for j in range(n)]
        loss, grad = one(out, lab, size_average=False, grad_scale=1.0 / n, running=run_b, per_image=True)
        torch.cuda.synchronize()
        for j in range(n):
            assert torch.equal(grad[j:j + 1], seq[j][1]), (n, h, w, j)
        tot = sum(float(s[0]) for s in seq)
        assert abs(float(loss) - tot) <= 3e-7 * abs(tot) and abs(float(run_b) - float(run_a)) <= 3e-7 * abs(tot), (float(loss), tot)
        # all five heads at once
        outs5 = [torch.randn(n, 1, h, w, device="cuda", generator=g) for _ in range(5)]
        l5, g5 = multi(outs5, lab, size_average=False, grad_scales=[0.1] * 5, per_image=True)
        for k in range(5):
            lk, gk = one(outs5[k], lab, size_average=False, grad_scale=0.1, per_image=True)
            assert torch.equal(g5[k], gk) and abs(float(l5[k]) - float(lk)) <= 3e-7 * abs(float(lk))
        # (2) two shards + global counts == the whole batch
        if n >= 2:
            whole_l, whole_g = one(out, lab, size_average=False, grad_scale=1.0)
            cnts = parallel.global_class_counts(lab)
            k = n // 2
            la, ga = one(out[:k].contiguous(), lab[:k].contiguous(), size_average=False, counts=cnts)
            lb, gb = one(out[k:].contiguous(), lab[k:].contiguous(), size_average=False, counts=cnts)
            assert torch.equal(torch.cat([ga, gb]), whole_g)
            assert abs(float(la) + float(lb) - float(whole_l)) <= 3e-7 * abs(float(whole_l))
            o2 = out[:k].clone().requires_grad_()
            lc = parallel.cbce_with_counts(o2, lab[:k], *cnts)            # HIP path, differentiable
            (lc * 0.5).backward()
            oc = out[:k].cpu().clone().requires_grad_()
            lcpu = parallel.cbce_with_counts(oc, lab[:k].cpu(), *[c.cpu() for c in cnts])      # torch expression
            (lcpu * 0.5).backward()
            assert abs(float(lc) - float(lcpu)) <= 1e-5 * abs(float(lcpu))
            assert float((o2.grad.cpu() - oc.grad).abs().max()) <= 1e-5 * float(oc.grad.abs().max())
        # (3) offset views
        if (h * w) % 4 != 0:
            base_o, base_l = out.reshape(-1), lab.reshape(-1)
            vo, vl = base_o[1:1 + h * w].view(1, 1, h, w), base_l[1:1 + h * w].view(1, 1, h, w)
            assert vo.data_ptr() % 16 != 0
            xa = vo.clone().requires_grad_()
            cbce(xa, vl.clone(), size_average=False).backward()
            xv = vo.detach().requires_grad_()
            lv = cbce(xv, vl, size_average=False)
            lv.backward()
            l2, g2 = one(vo, vl, size_average=False)
            assert torch.equal(xv.grad, xa.grad) and torch.equal(g2, xa.grad)


@pytest.mark.parametrize("cout,cin", [(64, 64), (128, 64), (512, 512), (40, 32), (64, 3), (16, 48)])
@pytest.mark.parametrize("dgrad", [False, True])
def test_pack_x3_layout_and_pieces(cout, cin, dgrad):
    """The pre-split pack (round 4: whole contiguous runs of the OIHW filter turned through LDS) against its definition:
    pack[piece][tap][cg][m][e] with the three bf16 pieces of W[m][8 cg + e][tap] (forward) or W[8 cg + e][m][8 - tap] (data gradient),
    output channels zero padded to a multiple of 32.  The pieces are checked through what they must satisfy: piece 0 is the value's bf16
    truncation-or-rounding (within one bf16 ulp), the three pieces sum back to the fp32 value within 2^-22 relative, padding is zero."""
    ops = _ops()
    k, m = (cout, cin) if dgrad else (cin, cout)
    if k % 16:
        pytest.skip("reduction channels must be a multiple of 16")
    g = torch.Generator().manual_seed(cout * 7 + cin + int(dgrad))
    wt = torch.randn(cout, cin, 3, 3, generator=g)
    buf = ops.pack_x3(wt.cuda(), dgrad=dgrad)
    mp, cg = (m + 31) // 32 * 32, k // 8
    pk = buf.view(torch.int16).view(3, 9, cg, mp, 8).cpu()
    pieces = (pk.to(torch.int32) << 16).view(torch.float32).double()          # a bf16 is the upper half of its fp32
    want = wt.flip(2, 3).permute(1, 0, 2, 3) if dgrad else wt                  # [m][k][3][3] with tap -> 8 - tap for the data gradient
    want = want.reshape(m, cg, 8, 9).permute(3, 1, 0, 2).double()              # [tap][cg][m][e]
    assert float(pieces[:, :, :, m:, :].abs().max()) == 0.0 if mp > m else True
    got = pieces[:, :, :, :m, :]
    scale = want.abs().clamp_min(1e-30)
    assert float(((got[0] - want).abs() / scale).max()) <= 2.0 ** -7
    assert float(((got.sum(0) - want).abs() / scale).max()) <= 2.0 ** -22


@pytest.mark.parametrize("shape", [(1, 9, 11), (2, 17, 35), (1, 16, 32), (3, 40, 70), (1, 33, 64)])
def test_input_gradient_of_the_first_convolution_on_the_matrix_pipe(shape):
    """dgrad_c3_mfma_kernel (the bf16-store mode's input gradient): bf16 dy x bf16 filter, fp32 accumulation.  Against float64 on the ROUNDED
    operands the only error is the fp32 accumulation order (1e-5); against the unrounded filter it is the filter's bf16 rounding."""
    ops = _ops()
    n, h, w = shape
    g = torch.Generator().manual_seed(17 + w)
    dyb = torch.randn(n, 64, h, w, generator=g).bfloat16()
    wt = torch.randn(64, 3, 3, 3, generator=g) / 5
    dx = ops.conv3x3_dgrad_c3(dyb.permute(0, 2, 3, 1).contiguous().cuda(), wt.cuda())
    assert dx.shape == (n, 3, h, w)
    ref_r = F.conv_transpose2d(dyb.double(), wt.bfloat16().double(), padding=1)
    emax, el2 = rel_err(dx.cpu(), ref_r)
    assert emax < 2e-5 and el2 < 1e-5, (shape, emax, el2)
    emax, el2 = rel_err(dx.cpu(), F.conv_transpose2d(dyb.double(), wt.double(), padding=1))
    assert emax < 1e-2 and el2 < 4e-3, (shape, "vs the unrounded filter", emax, el2)


@pytest.mark.gpu
def test_pool_code_bytes_and_the_backward_that_reads_them():
    """Round 5: the bf16 pooling writes one code byte per pooled element (first-maximum position in scan order + four "input > 0" bits) and
    maxpool_bwd_code_kernel reads it instead of the pool's input.  The bytes against their definition in numpy (ties, exact zeros, all-zero
    windows, clipped windows at odd sizes), the pooled values unchanged, and the backward BIT-IDENTICAL to the kernel that recomputes the
    argmax from the input -- with and without a side gradient."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    for shape in [(2, 16, 9, 13), (1, 64, 8, 8), (1, 8, 1, 1), (1, 24, 7, 2), (2, 128, 30, 54)]:
        n, c, h, w = shape
        x = F.relu(torch.randn(shape, generator=g)).bfloat16()
        x[:, : c // 4] = (x[:, : c // 4] * 2).round() / 2           # plenty of exact ties
        x[:, c // 4: c // 2, ::2, ::2] = 0                              # zeros inside windows
        x[:, -1] = 0                                                    # all-zero windows
        xg = nhwc(x.float()).bfloat16()
        y, code = ops.maxpool2x2_bf16act_code(xg)
        assert torch.equal(y, ops.maxpool2x2_bf16act(xg))
        ho, wo = (h + 1) // 2, (w + 1) // 2
        xp = torch.full((n, 2 * ho, 2 * wo, c), -1.0)                   # (-1: outside the image -- never a maximum, never positive)
        xp[:, :h, :w] = xg.float().cpu()
        win = torch.stack([xp[:, 0::2, 0::2], xp[:, 0::2, 1::2], xp[:, 1::2, 0::2], xp[:, 1::2, 1::2]], dim=-1)      # [n, ho, wo, c, 4] in scan order
        best = win.argmax(dim=-1)                                       # torch.argmax returns the FIRST maximum
        assert bool((win.gather(-1, best[..., None])[..., 0] == win.max(dim=-1).values).all())
        first = (win == win.max(dim=-1, keepdim=True).values).float().argmax(dim=-1)      # first index holding the maximum, explicitly
        pos = (win > 0).long()
        want = first + 4 * pos[..., 0] + 8 * pos[..., 1] + 16 * pos[..., 2] + 32 * pos[..., 3]
        assert torch.equal(code.cpu().long(), want), shape
        dy = nhwc(torch.randn(n, c, ho, wo, generator=g)).bfloat16()
        ds = nhwc(torch.randn(shape, generator=g)).bfloat16()
        for side in (ds, None):
            a = ops.maxpool2x2_bwd_bf16act_code(code, dy, (h, w), side)
            b = ops.maxpool2x2_bwd_bf16act(xg, dy, side)
            assert torch.equal(a, b), (shape, side is None)


@pytest.mark.gpu
@pytest.mark.parametrize("cout,cin", [(64, 3), (64, 64), (128, 64), (512, 512), (16, 256), (40, 24)])
@pytest.mark.parametrize("dgrad", [False, True])
def test_pack_bf16_layout_bit_exact(cout, cin, dgrad):
    """The bf16 packs (round 5: whole contiguous runs of the OIHW filter turned through LDS, all of a network's packs in one launch) against
    their definition, BIT FOR BIT: pack[tap][cg][m][e] = bf16_rne(W[m][8 cg + e][tap]) (forward) or bf16_rne(W[8 cg + e][m][8 - tap]) (data
    gradient); reduction channels zero padded to a multiple of 32 (conv1_1: 3 -> 32), output channels to a multiple of 32."""
    from osvos_pytorch_amd._lib import F32_BF16MFMA
    ops = _ops()
    g = torch.Generator().manual_seed(cout * 5 + cin + int(dgrad))
    wt = torch.randn(cout, cin, 3, 3, generator=g)
    buf = (ops.pack_dgrad if dgrad else ops.pack_fwd)(wt.cuda(), F32_BF16MFMA)
    k, m = (cout, cin) if dgrad else (cin, cout)
    kp, mp = (k + 31) // 32 * 32, (m + 31) // 32 * 32
    pk = buf.view(torch.int16)[: 9 * (kp // 8) * mp * 8].view(9, kp // 8, mp, 8).cpu()
    want = torch.zeros(9, kp, mp)
    src = wt.flip(2, 3).permute(1, 0, 2, 3) if dgrad else wt                   # [m][k][3][3], tap -> 8 - tap for the data gradient
    want[:, :k, :m] = src.reshape(m, k, 9).permute(2, 1, 0)
    want = want.view(9, kp // 8, 8, mp).permute(0, 1, 3, 2).contiguous().bfloat16().view(torch.int16)
    assert torch.equal(pk, want)
