"""Per-op wrappers over the C ABI, taking torch CUDA tensors.  These are thin (pointer + shape
marshalling only) and exist for the parity tests and for users who want single ops; the network
itself runs through the two whole-network calls in ``autograd.py``.

Tensors are NHWC ("channels last" memory, shape [N,H,W,C]) unless a name says nchw."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import F32, F32_BF16MFMA, F32_X3, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("osvos_pytorch_amd ops need CUDA (ROCm) tensors; there is no CPU fallback")


def nchw_to_nhwc(x, cpad):
    _need_cuda(x)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    y = torch.empty((n, h, w, cpad), device=x.device, dtype=torch.float32)
    check(lib().osvos_nchw_to_nhwc(_p(x), _p(y), n, c, h, w, cpad, F32, _stream()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x, c):
    _need_cuda(x)
    n, h, w, cs = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    check(lib().osvos_nhwc_to_nchw(_p(x), _p(y), n, c, h, w, cs, F32, _stream()), "nhwc_to_nchw")
    return y


def pack_fwd(w_oihw, dtype=F32):
    _need_cuda(w_oihw)
    w = w_oihw.contiguous().float()
    cout, cin = w.shape[:2]
    buf = torch.empty(lib().osvos_wpack_bytes(cout, cin, dtype) // 4, device=w.device, dtype=torch.float32)
    check(lib().osvos_pack_conv3x3_fwd(_p(w), _p(buf), cout, cin, dtype, _stream()), "pack_fwd")
    return buf


def pack_dgrad(w_oihw, dtype=F32):
    _need_cuda(w_oihw)
    w = w_oihw.contiguous().float()
    cout, cin = w.shape[:2]
    buf = torch.empty(lib().osvos_wpack_dgrad_bytes(cout, cin, dtype) // 4, device=w.device, dtype=torch.float32)
    check(lib().osvos_pack_conv3x3_dgrad(_p(w), _p(buf), cout, cin, dtype, _stream()), "pack_dgrad")
    return buf


def conv3x3_splitk(x, wpk, bias, cout, ksplit, relu=False, mask=None, tile=-1, dtype=F32):
    """fp32 conv cut into `ksplit` K parts (0 = automatic) + finalize kernel; dtype F32_X3: in the f32x3 arithmetic where it applies."""
    _need_cuda(x, wpk, bias, mask)
    n, h, w, cin = x.shape
    y = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    part = torch.empty(lib().osvos_conv3x3_splitk_ws_bytes(n, h, w, cout, F32), device=x.device, dtype=torch.uint8)
    check(lib().osvos_conv3x3_splitk(_p(x), _p(wpk), _p(bias), _p(mask), _p(y), n, h, w, cin, cout, cout, int(relu), dtype, tile,
                                     ksplit, _p(part), _stream()), "conv3x3_splitk")
    return y


def conv3x3(x, wpk, bias, cout, relu=False, mask=None, y_cs=None, tile=-1, dtype=F32):
    """x [N,H,W,Cin] (Cin % 8 == 0), wpk from pack_fwd / pack_dgrad -> [N,H,W,y_cs]."""
    _need_cuda(x, wpk, bias, mask)
    n, h, w, cin = x.shape
    y_cs = y_cs or cout
    y = torch.zeros((n, h, w, y_cs), device=x.device, dtype=torch.float32) if y_cs != cout else \
        torch.empty((n, h, w, y_cs), device=x.device, dtype=torch.float32)
    check(lib().osvos_conv3x3(_p(x), _p(wpk), _p(bias), _p(mask), _p(y), n, h, w, cin, cout, y_cs, int(relu), dtype, tile, _stream()), "conv3x3")
    return y


def conv3x3_bf16io(x, wpk, bias, cout, relu=False, mask=None, y_cs=None, tile=-1, want_bf16=True, want_f32=True):
    """bf16-MFMA convolution with explicit formats: x and mask fp32 or torch.bfloat16 [N,H,W,C]; returns (y fp32 | None, y bf16 | None)."""
    _need_cuda(x, wpk, bias, mask)
    n, h, w, cin = x.shape
    y_cs = y_cs or cout
    y = torch.zeros((n, h, w, y_cs), device=x.device, dtype=torch.float32) if want_f32 else None
    yb = torch.zeros((n, h, w, y_cs), device=x.device, dtype=torch.bfloat16) if want_bf16 else None
    check(lib().osvos_conv3x3_bf16io(_p(x), int(x.dtype == torch.bfloat16), _p(wpk), _p(bias), _p(mask),
                                     int(mask is not None and mask.dtype == torch.bfloat16), _p(y), _p(yb), n, h, w, cin, cout,
                                     y_cs, int(relu), tile, _stream()), "conv3x3_bf16io")
    return y, yb


def set_x3_pieces(pieces):
    """bf16 pieces per operand of the OSVOS_F32_X3 kernels called from this thread through ops.*: 3 (default, six products) or 2 (three products:
    what precision 'fp32x2' runs).  Callers restore 3 when they are done."""
    check(lib().osvos_set_x3_pieces(int(pieces)), "set_x3_pieces")


def conv3x3_bf16act_fused(x, wpk, bias, cout, relu=False, mask_bits=None, want_bits=False, want_pool=False, tile=-1):
    """The bf16-store trunk convolution with its fused epilogues (osvos_conv3x3_bf16act_fused): x torch.bfloat16 [N,H,W,Cin];
    returns (y bf16, y_bits int32 [N,H,W,cout/32] | None, pooled bf16 | None, pool_code uint8 | None)."""
    _need_cuda(x, wpk, bias, mask_bits)
    assert x.dtype == torch.bfloat16
    n, h, w, cin = x.shape
    y = torch.zeros((n, h, w, cout), device=x.device, dtype=torch.bfloat16)
    bits = torch.zeros((n, h, w, cout // 32), device=x.device, dtype=torch.int32) if want_bits else None
    pooled = torch.zeros((n, (h + 1) // 2, (w + 1) // 2, cout), device=x.device, dtype=torch.bfloat16) if want_pool else None
    code = torch.zeros((n, (h + 1) // 2, (w + 1) // 2, cout), device=x.device, dtype=torch.uint8) if want_pool else None
    check(lib().osvos_conv3x3_bf16act_fused(_p(x), _p(wpk), _p(bias), _p(mask_bits), _p(y), _p(bits), _p(pooled), _p(code), n, h, w, cin, cout,
                                            int(relu), tile, _stream()), "conv3x3_bf16act_fused")
    return y, bits, pooled, code


def conv3x3_wgrad_bf16act(x, dy, cin, cout, want_bias=True):
    """x, dy torch.bfloat16 NHWC (wide layers) -> (dW fp32 [cout,cin,3,3], db fp32)"""
    _need_cuda(x, dy)
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
    n, h, w, cin_s = x.shape
    ws = torch.empty(lib().osvos_wgrad_ws_bytes(n, h, w, cin_s, cout, F32_BF16MFMA), device=x.device, dtype=torch.uint8)
    dw = torch.empty((cout, cin, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.empty((cout,), device=x.device, dtype=torch.float32) if want_bias else None
    check(lib().osvos_conv3x3_wgrad_bf16act(_p(x), _p(dy), _p(ws), _p(dw), _p(db), n, h, w, cin, cin_s, cout, dy.shape[3], 0, _stream()), "wgrad_bf16act")
    return dw, db


def maxpool2x2_bf16act(x):
    _need_cuda(x)
    assert x.dtype == torch.bfloat16
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), device=x.device, dtype=torch.bfloat16)
    check(lib().osvos_maxpool2x2_bf16act(_p(x), _p(y), n, h, w, c, _stream()), "maxpool_bf16act")
    return y


def maxpool2x2_bf16act_code(x):
    """bf16 pooling that also returns the pool-code bytes [N,Ho,Wo,C] (uint8) for maxpool2x2_bwd_bf16act_code"""
    _need_cuda(x)
    assert x.dtype == torch.bfloat16
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), device=x.device, dtype=torch.bfloat16)
    code = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), device=x.device, dtype=torch.uint8)
    check(lib().osvos_maxpool2x2_bf16act_code(_p(x), _p(y), _p(code), n, h, w, c, _stream()), "maxpool_bf16act_code")
    return y, code


def maxpool2x2_bwd_bf16act_code(code, dy, hw, dside=None):
    """the pool's backward from the code bytes; hw = (H, W) of the pool's input"""
    _need_cuda(code, dy, dside)
    assert code.dtype == torch.uint8 and dy.dtype == torch.bfloat16 and (dside is None or dside.dtype == torch.bfloat16)
    n, _, _, c = dy.shape
    h, w = hw
    dx = torch.empty((n, h, w, c), device=dy.device, dtype=torch.bfloat16)
    check(lib().osvos_maxpool2x2_bwd_bf16act_code(_p(code), _p(dy), _p(dside), _p(dx), n, h, w, c, _stream()), "maxpool_bwd_bf16act_code")
    return dx


def maxpool2x2_bwd_bf16act(x, dy, dside=None):
    _need_cuda(x, dy, dside)
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and (dside is None or dside.dtype == torch.bfloat16)
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    check(lib().osvos_maxpool2x2_bwd_bf16act(_p(x), _p(dy), _p(dside), _p(dx), n, h, w, c, _stream()), "maxpool_bwd_bf16act")
    return dx


def conv3x3_bf16io_tiles():
    import ctypes
    buf = (ctypes.c_int * 64)()
    n = lib().osvos_conv3x3_bf16io_tiles(buf, 64)
    return [buf[i] for i in range(n)]


def nchw_to_nhwc_bf16copy(x, cpad):
    _need_cuda(x)
    n, c, h, w = x.shape
    y = torch.empty((n, h, w, cpad), device=x.device, dtype=torch.float32)
    yb = torch.empty((n, h, w, cpad), device=x.device, dtype=torch.bfloat16)
    check(lib().osvos_nchw_to_nhwc_bf16copy(_p(x.contiguous()), _p(y), _p(yb), n, c, h, w, cpad, _stream()), "nchw_to_nhwc_bf16copy")
    return y, yb


def maxpool2x2_bf16copy(x):
    _need_cuda(x)
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), device=x.device, dtype=torch.float32)
    yb = torch.empty_like(y, dtype=torch.bfloat16)
    check(lib().osvos_maxpool2x2_bf16copy(_p(x), _p(y), _p(yb), n, h, w, c, _stream()), "maxpool_bf16copy")
    return y, yb


def maxpool2x2_bwd_bf16copy(x, dy, dside=None):
    _need_cuda(x, dy, dside)
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    dxb = torch.empty_like(x, dtype=torch.bfloat16)
    check(lib().osvos_maxpool2x2_bwd_bf16copy(_p(x), _p(dy), _p(dside), _p(dx), _p(dxb), n, h, w, c, _stream()), "maxpool_bwd_bf16copy")
    return dx, dxb


def conv3x3_wgrad(x, dy, cin, cout, want_bias=True, accumulate_into=None, dtype=F32):
    """x [N,H,W,Cin_s], dy [N,H,W,Cout_s] -> (dW [cout,cin,3,3], db [cout])."""
    _need_cuda(x, dy)
    n, h, w, cin_s = x.shape
    cout_s = dy.shape[3]
    ws = torch.empty(lib().osvos_wgrad_ws_bytes(n, h, w, cin_s, cout, dtype), device=x.device, dtype=torch.uint8)
    if accumulate_into is not None:
        dw, db = accumulate_into
        acc = 1
    else:
        dw = torch.empty((cout, cin, 3, 3), device=x.device, dtype=torch.float32)
        db = torch.empty((cout,), device=x.device, dtype=torch.float32) if want_bias else None
        acc = 0
    check(lib().osvos_conv3x3_wgrad(_p(x), _p(dy), _p(ws), _p(dw), _p(db), n, h, w, cin, cin_s, cout, cout_s, acc, dtype, _stream()), "wgrad")
    return dw, db


def maxpool2x2(x):
    _need_cuda(x)
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), device=x.device, dtype=torch.float32)
    check(lib().osvos_maxpool2x2(_p(x), _p(y), n, h, w, c, F32, _stream()), "maxpool")
    return y


def maxpool2x2_bwd(x, dy, dside=None):
    _need_cuda(x, dy, dside)
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    check(lib().osvos_maxpool2x2_bwd(_p(x), _p(dy), _p(dside), _p(dx), n, h, w, c, F32, _stream()), "maxpool_bwd")
    return dx


def cbce(output, label, mode=1, want_grad=True):
    """Returns (loss 0-dim tensor, grad or None).  mode 0 size_average / 1 batch_average / 2 none."""
    _need_cuda(output, label)
    out = output.contiguous().float()
    lab = label.contiguous().float()
    loss = torch.empty((), device=out.device, dtype=torch.float32)
    grad = torch.empty_like(out) if want_grad else None
    scratch = torch.empty(4, device=out.device, dtype=torch.float64)
    check(lib().osvos_cbce(_p(out), _p(lab), _p(loss), _p(grad), _p(scratch), out.numel(), out.shape[0], mode, _stream()), "cbce")
    return loss, grad


def debug_conv3x3_naive(x, w_oihw, bias, relu=False):
    _need_cuda(x, w_oihw, bias)
    n, h, w, cin_s = x.shape
    cout, cin = w_oihw.shape[:2]
    y = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    check(lib().osvos_debug_conv3x3_naive(_p(x), _p(w_oihw.contiguous()), _p(bias), _p(y), n, h, w, cin, cin_s, cout, int(relu), _stream()), "naive conv")
    return y


def debug_mfma_layout(device="cuda"):
    out = torch.empty((4, 64, 16), device=device, dtype=torch.float32)
    check(lib().osvos_debug_mfma_layout(_p(out), _stream()), "mfma layout")
    return out


def sgd_step(p, g, buf, lr, momentum, weight_decay, first):
    _need_cuda(p, g, buf)
    check(lib().osvos_sgd_step(_p(p), _p(g), _p(buf), p.numel(), lr, momentum, weight_decay, int(first), _stream()), "sgd_step")


HEAD_MAX_BLOCKS = 1024      # OSVOS_HEAD_MAX_BLOCKS (include/osvos_hip.h): rows of 34 double partials per head_bwd launch


def head_lowres(prep, wd, bd, wf16):
    """prep [N,h,w,16] -> (score = bd + wd . prep, fpart = wf16 . prep), each [N,h,w]   (vgg_osvos.py:69 score_dsn; fuse commuted)"""
    _need_cuda(prep, wd, bd, wf16)
    n, h, w, _ = prep.shape
    score = torch.empty((n, h, w), device=prep.device, dtype=torch.float32)
    fpart = torch.empty((n, h, w), device=prep.device, dtype=torch.float32)
    check(lib().osvos_head_lowres(_p(prep), _p(wd), _p(bd), _p(wf16), _p(score), _p(fpart), n, h, w, F32, _stream()), "head_lowres")
    return score, fpart


def head_upsample(scores, fparts, f1s, f16s, fuse_bias, H, W):
    """four scales of (score, fpart) [N,h_i,w_i] + their k x k filters -> the five full-resolution logit maps [N,1,H,W]
    (transposed conv k = 2s, stride s, center crop, sum over scales + fuse bias: vgg_osvos.py:68-72, osvos_layers.py:51-56)"""
    _need_cuda(*scores, *fparts, *f1s, *f16s, fuse_bias)
    n = scores[0].shape[0]
    outs = [torch.empty((n, 1, H, W), device=scores[0].device, dtype=torch.float32) for _ in range(5)]
    arr = lambda ts: (C.c_void_p * len(ts))(*[C.c_void_p(t.data_ptr()) for t in ts])      # noqa: E731
    hs = (C.c_int * 4)(*[int(t.shape[1]) for t in scores])
    ws = (C.c_int * 4)(*[int(t.shape[2]) for t in scores])
    check(lib().osvos_head_upsample(arr(scores), arr(fparts), arr(f1s), arr(f16s), _p(fuse_bias), arr(outs), n, H, W, hs, ws, _stream()), "head_upsample")
    return outs


def head_bwd(prep, dside, dfused, f1, f16, wd, wf16, H, W, scale_idx):
    """adjoint of one scale of the head: (dprep [N,h,w,16], dwf16 [16], dwd [16], dbd) from the upstream gradients of side output
    `scale_idx` (dside, may be None) and of the fused map (dfused, may be None), both [N,1,H,W]"""
    _need_cuda(prep, dside, dfused, f1, f16, wd, wf16)
    n, h, w, _ = prep.shape
    dprep = torch.empty_like(prep)
    acc = torch.zeros((HEAD_MAX_BLOCKS, 34), device=prep.device, dtype=torch.float64)
    check(lib().osvos_head_bwd(_p(prep), _p(dside), _p(dfused), _p(f1), _p(f16), _p(wd), _p(wf16), _p(dprep), _p(acc), n, H, W, h, w,
                               scale_idx, F32, _stream()), "head_bwd")
    tot = acc.sum(0)
    return dprep, tot[0:16].float(), tot[16:32].float(), tot[32].float()


def pack_x3(w_oihw, dgrad=False):
    """pre-split bf16x3 pack of an OIHW fp32 filter for the f32x3 convolution (forward, or the data gradient's rotated filter)"""
    _need_cuda(w_oihw)
    cout, cin = int(w_oihw.shape[0]), int(w_oihw.shape[1])
    buf = torch.empty(lib().osvos_wpack_x3_bytes_abi(cout, cin, int(dgrad)), device=w_oihw.device, dtype=torch.uint8)
    check(lib().osvos_pack_conv3x3_x3(_p(w_oihw.contiguous()), _p(buf), cout, cin, int(dgrad), _stream()), "pack_x3")
    return buf


def conv3x3_x3(x, wpk3, bias, cout, relu=False, mask=None, y_cs=None, tile=-1):
    """f32x3 convolution with pre-split weights: x [N,H,W,Cin] fp32 -> [N,H,W,y_cs] fp32"""
    _need_cuda(x, wpk3, bias, mask)
    n, h, w, cin = x.shape
    y_cs = y_cs or cout
    y = torch.zeros((n, h, w, y_cs), device=x.device, dtype=torch.float32) if y_cs != cout else torch.empty((n, h, w, y_cs), device=x.device, dtype=torch.float32)
    check(lib().osvos_conv3x3_x3(_p(x), _p(wpk3), _p(bias), _p(mask), _p(y), n, h, w, cin, cout, y_cs, int(relu), tile, _stream()), "conv3x3_x3")
    return y





_SK_WS = {}


def streamk_workspace(device):
    """per-device stream-K workspace (tickets zeroed once; every launch leaves them zero)"""
    key = str(device)
    if key not in _SK_WS:
        ws = torch.empty(lib().osvos_conv3x3_x3_streamk_ws_bytes(), device=device, dtype=torch.uint8)
        ws[:lib().osvos_conv3x3_x3_streamk_ticket_bytes()].zero_()
        _SK_WS[key] = ws
    return _SK_WS[key]


def conv3x3_x3_streamk(x, wpk3, bias, cout, relu=False, mask=None, tile=-1, grid=0, want_pooled=False):
    """stream-K form of conv3x3_x3 (grid 0 = automatic, > 0 = that many persistent workgroups); returns y or (y, pooled)"""
    _need_cuda(x, wpk3, bias, mask)
    n, h, w, cin = x.shape
    y = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    pooled = torch.empty((n, (h + 1) // 2, (w + 1) // 2, cout), device=x.device, dtype=torch.float32) if want_pooled else None
    check(lib().osvos_conv3x3_x3_streamk(_p(x), _p(wpk3), _p(bias), _p(mask), _p(y), _p(pooled), n, h, w, cin, cout, cout, int(relu), tile, grid,
                                         _p(streamk_workspace(x.device)), _stream()), "conv3x3_x3_streamk")
    return (y, pooled) if want_pooled else y


def conv3x3_dgrad_c3(dy, w_oihw):
    """input gradient of a 3-input-channel convolution: dy [N,H,W,Cout] fp32, or bf16 with Cout = 64 (the bf16 mode: bf16 filter, MFMA);
    filter [Cout,3,3,3] -> dx fp32 NCHW [N,3,H,W]"""
    _need_cuda(dy, w_oihw)
    n, h, w, cout = dy.shape
    dx = torch.empty((n, 3, h, w), device=dy.device, dtype=torch.float32)
    if dy.dtype == torch.bfloat16:
        check(lib().osvos_conv3x3_dgrad_c3_bf16mma(_p(dy.contiguous()), _p(pack_dgrad(w_oihw, F32_BF16MFMA)), _p(dx), n, h, w, cout, _stream()),
              "dgrad_c3 (bf16 mfma)")
    else:
        check(lib().osvos_conv3x3_dgrad_c3(_p(dy.contiguous()), _p(pack_dgrad(w_oihw)), _p(dx), n, h, w, cout, _stream()), "dgrad_c3")
    return dx
