"""autograd glue: one Function for the whole network (forward = osvos_net_forward, backward =
osvos_net_backward) and one for the loss.  PyTorch is plumbing here: it owns device memory,
streams and the autograd tape; every FLOP runs in libosvos_hip.so."""
from __future__ import annotations

import ctypes as C
import os

import torch

from ._lib import DEFER_JOIN, F32, F32_BF16MFMA, F32_X3, GENERIC_DECONV, INFERENCE, NPARAMS, X3_HALF_PIECES, X3_HALF_PIECES_BWD, X3_TWO_PIECES, check, lib, ptr_array

# indices (state_dict order) of the frozen transposed-conv weights: lr 0 in both reference
# scripts (train_online.py:84-85, train_parent.py:99-100); their gradients are never formed
_FROZEN = set(range(8))
# precision name -> (library dtype, pieces per operand of the f32x3 kernels in the forward, ... in the backward); see OSVOS.set_precision.
# Pieces: 3 = three bf16 (six products, the default), 2 = two bf16 (three products), 22 = two FP16 under block exponents (three products; csrc/h2split.h)
PRECISIONS = {"fp32": (F32, 3, 3), "bf16": (F32_BF16MFMA, 3, 3), "fp32x3": (F32_X3, 3, 3), "fp32x2": (F32_X3, 2, 2), "fp32x3b2": (F32_X3, 3, 2),
              "fp32h2": (F32_X3, 22, 22), "fp32x3h2": (F32_X3, 3, 22)}
_PIECE_FLAG = {3: 0, 2: X3_TWO_PIECES, 22: X3_HALF_PIECES}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class NetRuntime:
    """Packed-parameter cache of one OSVOS module (wbuf of include/osvos_hip.h).  Re-packs when
    any parameter's (storage, version) changed, e.g. after optimizer.step()."""

    def __init__(self):
        self.wbuf = None
        self.key = None
        self.deconv_key = None
        # 'fp32x3' (default: fp32 tensors and fp32-grade results, the wide 3x3 convolutions on the bf16 matrix pipe with three-way
        # split operands), 'fp32' (the same on the exact fp32 MFMA kernels) or 'bf16' (bf16 MFMA operands and bf16 trunk tensors,
        # fp32 accumulate; head and loss stay fp32).  OSVOS_PRECISION sets the initial value.
        self.dtype, self.pieces_fwd, self.pieces_bwd = F32_X3, 3, 3
        self.set_precision(os.environ.get("OSVOS_PRECISION", "fp32x3").lower() if os.environ.get("OSVOS_PRECISION", "fp32x3").lower() in PRECISIONS else "fp32x3")
        self.aux_stream = None        # second HIP stream: wgrad kernels overlap the dgrad kernels
        self.aux2_stream = None       # third: the slab reduces of the weight gradients
        self.auxf_stream = None       # forward side branches (own stream: a forward pipelined under the previous backward must not queue
                                      # behind that backward's weight-gradient kernels)
        self.two_streams = os.environ.get("OSVOS_TWO_STREAMS", "1") != "0"
        # in-place accumulation into existing .grad tensors inside backward (see OSVOSNetFunction.backward): explicit opt-in
        self.inplace_accumulate = os.environ.get("OSVOS_INPLACE_GRAD", "0") == "1"
        self.generic_head = False     # upscale[i].weight not diagonal / shared-filter: generic transposed-convolution head
        self.grad_events = None       # hipEvent_t handles for the NEXT backward (GradientAllReducer.arm), consumed by it
        self.grad_events_recorded = False     # did the last backward record them (it does only when it accumulated in place)
        # Deferred join (opt-in, gradient-accumulation loops: TrainLoop / bench.py with OSVOS_DEFER_JOIN=1): a backward that accumulates in
        # place returns as soon as its data-gradient chain is enqueued and leaves the weight-gradient tail running on the side streams --
        # under the NEXT micro-batch's forward, whose 256-pixel tiles leave ~18 % of the CUs idle at batch 1.  `join_backward` must run
        # before anything reads a parameter gradient on the main stream (optimizer step, all-reduce).
        self.defer_join = False
        self.pending_join = None              # device of the un-joined backward

    # The side streams are shared by every OSVOS module of the process (one set per device).  ROCm maps HIP streams onto a handful
    # of hardware queues (GPU_MAX_HW_QUEUES); a second module with three more streams of its own ends up sharing queues with the
    # first one's, two of the backward's streams serialise and the step loses its overlap (measured: 130 -> 98 frames/s for the
    # second net built in one process).  Work of different modules on the same stream is ordered by the stream, as it should be.
    _shared_streams = {}

    @classmethod
    def _stream_for(cls, device, name):
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), name)
        st = cls._shared_streams.get(key)
        if st is None:
            prio = os.environ.get("OSVOS_AUX_PRIORITY", "")       # tuning: HIP priority of the weight-gradient / reduce streams (lower = more urgent)
            try:
                st = torch.cuda.Stream(device=device, priority=int(prio)) if prio and name in ("wgrad", "reduce") else torch.cuda.Stream(device=device)
            except Exception:
                st = torch.cuda.Stream(device=device)
            cls._shared_streams[key] = st
        return st

    def join_backward(self):
        """Make the current stream wait for the weight-gradient work a deferred-join backward left on the side streams."""
        if self.pending_join is None:
            return
        dev, self.pending_join = self.pending_join, None
        with torch.cuda.device(dev):
            check(lib().osvos_net_join(_stream(), C.c_void_p(self.aux_stream.cuda_stream) if self.aux_stream is not None else None,
                                       C.c_void_p(self.aux2_stream.cuda_stream) if self.aux2_stream is not None else None), "net_join")

    def aux(self, device):
        if not self.two_streams:
            return None
        self.aux_stream = self._stream_for(device, "wgrad")
        return C.c_void_p(self.aux_stream.cuda_stream)

    def auxf(self, device):
        if not self.two_streams or os.environ.get("OSVOS_SIDE_STREAM", "1") == "0":
            return None
        self.auxf_stream = self._stream_for(device, "side")
        return C.c_void_p(self.auxf_stream.cuda_stream)

    def aux2(self, device):
        if not self.two_streams or os.environ.get("OSVOS_THREE_STREAMS", "1") == "0":
            return None
        self.aux2_stream = self._stream_for(device, "reduce")
        return C.c_void_p(self.aux2_stream.cuda_stream)

    def set_precision(self, name):
        # round 6: the f32x3 kernels can take TWO bf16 pieces per operand (three products instead of six) -- same dtype, packs and workspaces, one flag
        # on the network calls: 'fp32x2' in both passes, 'fp32x3b2' in the BACKWARD only (forward = 'fp32x3' bit for bit)
        # 'fp32h2': TWO FP16 pieces under block exponents in both passes (fp32-grade at three products: the packs change format, hence the re-pack);
        # 'fp32x3h2': the forward of 'fp32x3' bit for bit, the backward on FP16 pairs
        dt, pf, pb = PRECISIONS[name]
        if (pf == 22) != (self.pieces_fwd == 22) or (pb == 22) != (self.pieces_bwd == 22):
            self.key = None
        self.pieces_fwd, self.pieces_bwd = pf, pb
        if dt != self.dtype:
            self.dtype, self.wbuf, self.key = dt, None, None

    def ensure_packed(self, params):
        dev = params[0].device
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self.wbuf is not None and key == self.key and self.wbuf.device == dev:
            return
        l = lib()
        if self.wbuf is None or self.wbuf.device != dev:
            self.wbuf = torch.empty(l.osvos_net_wbuf_bytes(self.dtype), device=dev, dtype=torch.uint8)
            self.deconv_key = None
        dkey = key[:4]
        if dkey != self.deconv_key:
            # (round 6: the generic head runs on the bf16 trunk too -- the head is fp32 in every precision, its backward now also writes the bf16
            #  copy of the side_prep output gradient the bf16-store mode's convolutions read)
            self.generic_head = not self._deconv_is_diagonal(params[:4])
            self.deconv_key = dkey
        check(l.osvos_net_pack(ptr_array([p.data_ptr() for p in params]), C.c_void_p(self.wbuf.data_ptr()),
                               self.cdtype() | (X3_HALF_PIECES if self.pieces_fwd == 22 else 0) | (X3_HALF_PIECES_BWD if self.pieces_bwd == 22 else 0),
                               1, _stream()), "net_pack")
        self.key = key

    def cdtype(self):
        """dtype argument of the osvos_net_* calls: precision, plus the generic-deconvolution flag when the upscale weights need it."""
        return self.dtype | (GENERIC_DECONV if self.generic_head else 0)

    def cdtype_fwd(self):
        return self.cdtype() | _PIECE_FLAG[self.pieces_fwd]

    def cdtype_bwd(self):
        return self.cdtype() | _PIECE_FLAG[self.pieces_bwd]

    @staticmethod
    def _deconv_is_diagonal(ups):
        """True when every upscale[i].weight is diagonal with ONE shared filter -- what interp_surgery writes (osvos_layers.py:72-85)
        and what both reference scripts freeze with lr 0: the commuted head (dot-16, then one bilinear gather) is exact for it.
        Anything else (un-frozen / re-initialised deconvs; the reference runs arbitrary [16,16,k,k] weights, vgg_osvos.py:46,68) takes
        the generic transposed-convolution head (csrc/head_generic.hip), which also forms the deconv weight gradients."""
        l = lib()
        res = torch.empty((4, 2), device=ups[0].device, dtype=torch.float32)
        for i, w in enumerate(ups):
            check(l.osvos_deconv_diag_check(C.c_void_p(w.data_ptr()), w.shape[0], w.shape[2],
                                            C.c_void_p(res[i].data_ptr()), _stream()), "deconv_diag_check")
        return float(res.max().item()) == 0.0


class OSVOSNetFunction(torch.autograd.Function):
    """(x [N,3,H,W], 52 parameters) -> 5 logit maps [N,1,H,W]   (vgg_osvos.py:59-74)."""

    @staticmethod
    def forward(ctx, rt, x, *params):
        ctx.set_materialize_grads(False)      # unused heads arrive as None, not as zero maps
        if not x.is_cuda:
            raise RuntimeError("OSVOS (osvos_pytorch_amd) runs on the GPU only: move the module and the input to "
                               "'cuda'; there is no CPU fallback")
        if len(params) != NPARAMS:
            raise RuntimeError("expected %d parameters, got %d" % (NPARAMS, len(params)))
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("expected input of shape [N,3,H,W], got %s" % (tuple(x.shape),))
        l = lib()
        xin = x.detach().contiguous().float()
        ps = [p.detach().contiguous() for p in params]
        for p in ps:
            if p.dtype != torch.float32 or p.device != xin.device:
                raise RuntimeError("parameters must be float32 on the input's device")
        rt.ensure_packed(ps)
        n, _, h, w = xin.shape
        need_bwd = any(ctx.needs_input_grad)       # False under torch.no_grad(): forward-only workspace
        nbytes = l.osvos_net_ws_bytes(n, h, w, rt.dtype) if need_bwd else l.osvos_net_ws_bytes_infer(n, h, w, rt.dtype)
        ctx.cdtype = rt.cdtype()
        ctx.cdtype_bwd = rt.cdtype_bwd()
        ws = torch.empty(nbytes, device=xin.device, dtype=torch.uint8)
        outs = [torch.empty((n, 1, h, w), device=xin.device, dtype=torch.float32) for _ in range(5)]
        check(l.osvos_net_forward(C.c_void_p(xin.data_ptr()), C.c_void_p(rt.wbuf.data_ptr()), C.c_void_p(ws.data_ptr()),
                                  ptr_array([o.data_ptr() for o in outs]), n, h, w, rt.cdtype_fwd() | (0 if need_bwd else INFERENCE), _stream(),
                                  rt.auxf(xin.device)), "net_forward")
        # the activation workspace rides in autograd's saved-tensor slot: released right after a plain backward, kept under
        # retain_graph=True (a second backward recomputes every gradient buffer from the untouched forward half: no buffer of the
        # forward is written by the backward), and a second backward WITHOUT retain_graph raises autograd's own error, like the reference
        ctx.save_for_backward(ws)
        ctx.rt, ctx.shape = rt, (n, h, w)
        ctx.param_meta = [(tuple(p.shape), p.device) for p in ps]
        ctx.params = params          # for in-place gradient accumulation in backward
        ctx.pack_key = rt.key
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        rt = ctx.rt
        (ws,) = ctx.saved_tensors      # (freed by a previous backward without retain_graph=True: autograd raises its usual error here)
        n, h, w = ctx.shape
        if rt.key != ctx.pack_key:
            raise RuntimeError("parameters changed between forward and backward of the same graph")
        l = lib()
        dev = ws.device
        d = [None if g is None else g.contiguous().float() for g in douts]
        wanted = []
        for i in range(len(ctx.param_meta)):
            # the deconv weights (params 0..7): frozen by lr 0 in both reference scripts (train_online.py:84-85) -- the commuted head
            # forms no gradient for them; the generic head (non-diagonal weights = somebody trains them) does
            need = ctx.needs_input_grad[2 + i] and (i not in _FROZEN or bool(ctx.cdtype & GENERIC_DECONV))
            if need and 4 <= i < 8 and d[i - 4] is None:
                need = False       # upscale_[i] only sees the side head i
            if need and 42 <= i < 50 and all(g is None for g in d[:4]):
                need = False       # score_dsn gets no gradient when only the fused head is used
            wanted.append(need)
        # The weight-gradient kernels produce a layer's weight AND bias gradient in one launch, and the C side launches them
        # when the layer's weight target is non-NULL: a layer with exactly one of (weight, bias) trainable still needs both
        # targets.  The unwanted half goes to a scratch tensor that autograd never sees.
        launch = list(wanted)
        for wi in range(8, 42, 2):                       # (weight, bias) pairs of the 13 trunk + 4 side_prep convs
            launch[wi] = launch[wi + 1] = wanted[wi] or wanted[wi + 1]
        # Gradient accumulation (loss /= nAveGrad; backward; ... step every nAveGrad): when every wanted
        # parameter already holds a dense .grad, the slab-reduce kernels add into it directly and autograd
        # is handed None (no 52 extra add kernels, no 61 MB of temporaries per micro-batch).  OPT-IN
        # (OSVOS.set_inplace_grad_accumulation(True); TrainLoop and bench.py do it): a custom Function cannot tell
        # loss.backward() from torch.autograd.grad(...), and under the latter .grad must stay untouched and the gradients
        # must be RETURNED -- so the default is the plain autograd contract.
        inplace = rt.inplace_accumulate and all(
            (not w) or (p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.device == dev)
            for w, p in zip(wanted, ctx.params))

        def scratch(i):
            return torch.empty(ctx.param_meta[i][0], device=dev, dtype=torch.float32)
        if inplace:
            targets = [(p.grad if w else scratch(i)) if l else None for i, (l, w, p) in enumerate(zip(launch, wanted, ctx.params))]
            grads = [None] * len(wanted)
        else:
            targets = [scratch(i) if l else None for i, l in enumerate(launch)]
            grads = [t if w else None for t, w in zip(targets, wanted)]
        dx = torch.empty((n, 3, h, w), device=dev, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        ev, rt.grad_events = getattr(rt, "grad_events", None), None
        rt.grad_events_recorded = bool(ev and inplace)
        if ev and inplace:         # data-parallel overlap (parallel.GradientAllReducer.arm): one ready-event per completion group
            check(l.osvos_net_arm_grad_events(ptr_array(ev), len(ev)), "net_arm_grad_events")
        aux, aux2 = rt.aux(dev), rt.aux2(dev)
        # deferred join: only when nothing of this call is handed back to autograd (in-place accumulation) and no ready-events are armed
        defer = bool(rt.defer_join and inplace and not ev and aux is not None)
        check(l.osvos_net_backward(C.c_void_p(rt.wbuf.data_ptr()), C.c_void_p(ws.data_ptr()),
                                   ptr_array([None if g is None else g.data_ptr() for g in d]),
                                   ptr_array([None if g is None else g.data_ptr() for g in targets]),
                                   C.c_void_p(dx.data_ptr()) if dx is not None else None,
                                   n, h, w, ctx.cdtype_bwd | (DEFER_JOIN if defer else 0), 1 if inplace else 0, _stream(), aux, aux2), "net_backward")
        if defer:
            # the side streams still read the workspace (activations, upstream gradients, slabs) and write the scratch halves: the caching
            # allocator must not hand that memory out again before they are done with it
            for st in (rt.aux_stream, rt.aux2_stream):
                if st is not None:
                    ws.record_stream(st)
                    for t in targets:
                        if t is not None:
                            t.record_stream(st)
            rt.pending_join = dev
        return (None, dx) + tuple(grads)


class CBCELossFunction(torch.autograd.Function):
    """class_balanced_cross_entropy_loss (osvos_layers.py:19-48); loss and dLoss/dOutput come out
    of the same kernel pass, the backward only scales by the (device-resident) upstream gradient."""

    @staticmethod
    def forward(ctx, output, label, mode):
        if not output.is_cuda:
            raise RuntimeError("class_balanced_cross_entropy_loss (osvos_pytorch_amd) needs CUDA tensors; no CPU fallback")
        l = lib()
        out = output.detach().contiguous().float()
        lab = label.detach().to(device=out.device, dtype=torch.float32).contiguous()
        if lab.numel() != out.numel():
            raise RuntimeError("output and label must have the same number of elements")
        loss = torch.empty((), device=out.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        grad = torch.empty_like(out) if need else None
        scratch = torch.empty(4, device=out.device, dtype=torch.float64)
        check(l.osvos_cbce(C.c_void_p(out.data_ptr()), C.c_void_p(lab.data_ptr()), C.c_void_p(loss.data_ptr()),
                           C.c_void_p(grad.data_ptr()) if need else None, C.c_void_p(scratch.data_ptr()),
                           out.numel(), out.shape[0], int(mode), _stream()), "cbce")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        if grad is None:
            return None, None, None
        g = g.detach().to(torch.float32).contiguous()
        gx = torch.empty_like(grad)
        check(lib().osvos_scale(C.c_void_p(grad.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(gx.data_ptr()),
                                grad.numel(), _stream()), "scale")
        return gx, None, None


class CBCECountsFunction(torch.autograd.Function):
    """class-balanced BCE of a SHARD of a batch with the global batch's (n_pos, n_total, n_images) as arguments (parallel.cbce_with_counts):
    the HIP loss kernel of ``CBCELossFunction`` with external counts."""

    @staticmethod
    def forward(ctx, output, label, counts, mode):
        loss, grad = cbce_step(output, label, mode, 1.0, None, counts=counts)
        ctx.grad = grad if ctx.needs_input_grad[0] else None
        return loss

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        if grad is None:
            return None, None, None, None
        g = g.detach().to(torch.float32).contiguous()
        gx = torch.empty_like(grad)
        check(lib().osvos_scale(C.c_void_p(grad.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(gx.data_ptr()), grad.numel(), _stream()), "scale")
        return gx, None, None, None


CBCE_PER_IMAGE = 1      # include/osvos_hip.h OSVOS_CBCE_PER_IMAGE
CBCE_SCRATCH_ZEROED = 2  # include/osvos_hip.h OSVOS_CBCE_SCRATCH_ZEROED

# The loss call's scratch (class counts, loss sums, arrival ticket): every call leaves it zero, so ONE zero-initialised buffer per (device, stream,
# size) serves every call of a training loop and the call enqueues no memset (a 32-byte hipMemsetAsync is two fill kernels: 14 us between the
# forward and the backward at batch 1).  Keyed by stream: calls on one stream are ordered, calls on different streams get different buffers.
_CBCE_SCRATCH = {}


def _cbce_scratch(dev, nbytes):
    if os.environ.get("OSVOS_CBCE_PERSISTENT", "1") == "0":      # A/B switch: a fresh buffer and a memset per call, as rounds 1-5
        return torch.empty(nbytes, device=dev, dtype=torch.uint8), 0
    if torch.cuda.is_current_stream_capturing():      # a graph gets a buffer of its own, zeroed by a node of the graph (no sharing with eager calls)
        return torch.zeros(nbytes, device=dev, dtype=torch.uint8), 0
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream, nbytes)
    buf = _CBCE_SCRATCH.get(key)
    if buf is None:
        if len(_CBCE_SCRATCH) > 64:
            _CBCE_SCRATCH.clear()
        buf = _CBCE_SCRATCH[key] = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
    return buf, CBCE_SCRATCH_ZEROED


def _counts_tensor(counts, dev):
    """(n_pos, n_total, n_images) of a global batch -> device fp32[3] (the class weights are float32 quotients in the reference too)"""
    if counts is None:
        return None
    if torch.is_tensor(counts):
        t = counts.detach().to(device=dev, dtype=torch.float32).contiguous()
    else:
        t = torch.stack([c.detach().to(device=dev, dtype=torch.float32).reshape(()) if torch.is_tensor(c) else torch.tensor(float(c), device=dev)
                         for c in counts])
    if t.numel() != 3:
        raise RuntimeError("counts must hold (n_pos, n_total, n_images)")
    return t


def cbce_step(output, label, mode, grad_scale=1.0, running=None, per_image=False, counts=None):
    """One training-loop use of the class-balanced BCE without the autograd scalar chain behind it: returns ``(loss, grad)`` where ``loss``
    is the plain 0-dim loss (detached; what the reference adds to ``running_loss``, train_online.py:128) and ``grad`` =
    ``grad_scale * dLoss/dOutput`` -- ready for ``torch.autograd.backward([output], [grad])``.  ``grad_scale`` is the upstream gradient the
    reference's ``loss /= nAveGrad; loss.backward()`` hands this loss (train_online.py:140-141); ``running`` (0-dim fp32 CUDA tensor) gets
    ``+= loss`` inside the kernel.  Same roundings as ``class_balanced_cross_entropy_loss(...)``, ``/=``, ``.backward()``; five small
    ATen launches (ones, div, its backward, add, scale) per micro-batch fewer between the loss and the head's backward.
    ``per_image``: every image of the batch is its own reference batch of one (own class weights; the losses are summed) -- the micro-batches
    of an accumulation window in one call.  ``counts``: ``(n_pos, n_total, n_images)`` of the GLOBAL batch this tensor is a shard of
    (``parallel.global_class_counts``): weights and divisors come from them."""
    losses, grads = cbce_step_multi([output], label, mode, [grad_scale], [running], per_image=per_image, counts=counts)
    return losses[0], grads[0]


def cbce_step_multi(outputs, label, mode, grad_scales, runnings=None, per_image=False, counts=None):
    """``cbce_step`` for several heads against one label (the parent loop's five losses, train_parent.py:143-147) in ONE library call: the
    label's class counts are formed once and the loss / gradient sweep of all heads is one launch.  Returns ``(losses, grads)``: ``losses`` a
    float32 CUDA tensor of ``len(outputs)`` plain losses (detached), ``grads[k] = grad_scales[k] * dLoss_k/dOutput_k``; ``runnings`` (list of
    0-dim fp32 CUDA tensors or Nones) get ``+= loss_k`` on the device.  Per head the arithmetic is ``cbce_step``'s."""
    n = len(outputs)
    if not (1 <= n <= 8) or len(grad_scales) != n or (runnings is not None and len(runnings) != n):
        raise RuntimeError("cbce_step_multi: 1..8 heads with one grad_scale (and one running entry) each")
    if not all(o.is_cuda for o in outputs):
        raise RuntimeError("class_balanced_cross_entropy_loss (osvos_pytorch_amd) needs CUDA tensors; no CPU fallback")
    if per_image and counts is not None:
        raise RuntimeError("per_image and counts exclude each other")
    outs = [o.detach().contiguous().float() for o in outputs]
    dev = outs[0].device
    lab = label.detach().to(device=dev, dtype=torch.float32).contiguous()
    if any(o.numel() != lab.numel() for o in outs):
        raise RuntimeError("every output and the label must have the same number of elements")
    for r in (runnings or []):
        if r is not None and not (r.is_cuda and r.dtype == torch.float32 and r.numel() == 1):
            raise RuntimeError("running entries must be one-element float32 CUDA tensors")
    n_img = int(outs[0].shape[0])
    flags = CBCE_PER_IMAGE if per_image else 0
    cnt = _counts_tensor(counts, dev)
    losses = torch.empty(n, device=dev, dtype=torch.float32)
    grads = [torch.empty_like(o) for o in outs]
    scratch, zeroed = _cbce_scratch(dev, int(lib().osvos_cbce_scratch_bytes(n, n_img, flags)))
    flags |= zeroed
    vp = C.c_void_p
    a_out = (vp * n)(*[vp(o.data_ptr()) for o in outs])
    a_loss = (vp * n)(*[vp(losses.data_ptr() + 4 * k) for k in range(n)])
    a_grad = (vp * n)(*[vp(g.data_ptr()) for g in grads])
    a_run = (vp * n)(*[vp(r.data_ptr()) if r is not None else None for r in (runnings or [None] * n)])
    a_scale = (C.c_float * n)(*[float(s) for s in grad_scales])
    check(lib().osvos_cbce_step_ex(a_out, vp(lab.data_ptr()), a_loss, a_grad, vp(scratch.data_ptr()), lab.numel(), n_img, int(mode), flags,
                                   vp(cnt.data_ptr()) if cnt is not None else None, n, a_scale, a_run, _stream()), "cbce_step")
    return losses, [g.view_as(o) for g, o in zip(grads, outputs)]
