"""Result writer and evaluator for the per-sequence test loop (reference train_online.py:181-189).

The reference turns the fused logit map into a probability (``1 / (1 + exp(-x))``), hands it to ``scipy.misc.imsave``
(scipy <= 1.1: min-max byte scaling, PIL mode 'L') and leaves the DAVIS evaluation to an external toolkit.  Here the
sigmoid + byte scaling run on the device (``osvos_mask_to_bytes``: one byte per pixel crosses PCIe instead of four),
the PNG is written by a small zlib encoder (no PIL / scipy dependency), and the DAVIS region measure J (Jaccard index
of the thresholded mask) with its mean / recall / decay statistics is computed from device-side pixel counts.
"""
import ctypes as C
import struct
import zlib

import numpy as np
import torch

from ._lib import check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def mask_bytes(logits):
    """logits: float32 CUDA tensor [N,1,H,W] or [N,H,W] -> uint8 CUDA tensor [N,H,W] (imsave's byte image per frame)."""
    if not logits.is_cuda:
        raise RuntimeError("osvos_pytorch_amd.results needs CUDA (ROCm) tensors; there is no CPU fallback")
    x = logits.detach().float().contiguous()
    n = x.shape[0]
    count = x.numel() // n
    out = torch.empty((n,) + tuple(x.shape[-2:]), device=x.device, dtype=torch.uint8)
    scratch = torch.empty(2 * n, device=x.device, dtype=torch.int32)
    check(lib().osvos_mask_to_bytes(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()), count, n, _stream()),
          "mask_to_bytes")
    return out


def write_png(path, img):
    """8-bit grayscale PNG of a [H,W] uint8 array (what PIL writes for mode 'L'; any decoder reads the same pixels)."""
    a = np.ascontiguousarray(img, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("write_png expects a 2-D uint8 array, got shape %r" % (a.shape,))
    h, w = a.shape
    raw = np.empty((h, w + 1), dtype=np.uint8)
    raw[:, 0] = 0                       # filter type 0 (None) per scanline
    raw[:, 1:] = a

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) \
        + chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def save_masks(fused_logits, paths):
    """Test-loop body of train_online.py:181-187 for a batch: one PNG per frame."""
    b = mask_bytes(fused_logits).cpu().numpy()
    for img, p in zip(b, paths):
        write_png(p, img)


def jaccard(logits, gts, threshold=0.5):
    """DAVIS region measure per frame: J = |P & G| / |P | G| with P = sigmoid(logit) > threshold, G = gt > 0.5
    (J = 1 when both are empty).  logits, gts: CUDA tensors of N frames."""
    if not (logits.is_cuda and gts.is_cuda):
        raise RuntimeError("osvos_pytorch_amd.results needs CUDA (ROCm) tensors; there is no CPU fallback")
    if not 0.0 < threshold < 1.0:
        raise ValueError("threshold must be a probability in (0, 1)")
    x = logits.detach().float().contiguous()
    g = gts.detach().to(device=x.device, dtype=torch.float32).contiguous()
    if g.numel() != x.numel():
        raise ValueError("logits and ground truth differ in size: %r vs %r" % (tuple(x.shape), tuple(g.shape)))
    n = x.shape[0]
    counts = torch.empty(2 * n, device=x.device, dtype=torch.int64)
    thr = float(np.log(threshold / (1.0 - threshold)))
    check(lib().osvos_mask_iou_counts(C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(counts.data_ptr()), x.numel() // n, n, thr,
                                      _stream()), "mask_iou_counts")
    c = counts.cpu().numpy().reshape(n, 2)
    return [1.0 if u == 0 else float(i) / float(u) for i, u in c]


def davis_statistics(js):
    """mean, recall (fraction of frames with J > 0.5) and decay (mean of the first quarter minus mean of the last
    quarter of the frames) of a sequence's per-frame J, the three numbers DAVIS reports for the region measure."""
    j = np.asarray(js, dtype=np.float64)
    if j.size == 0:
        raise ValueError("no frames")
    bins = np.array_split(np.arange(j.size), 4) if j.size >= 4 else [np.arange(j.size)] * 4
    decay = float(j[bins[0]].mean() - j[bins[3]].mean()) if bins[0].size and bins[3].size else 0.0
    return {"mean": float(j.mean()), "recall": float((j > 0.5).mean()), "decay": decay}
