"""ctypes binding of libosvos_hip.so (C ABI: include/osvos_hip.h).  Fails loudly: there is no
CPU or PyTorch fallback for any compute entry point."""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
import threading

# (GPU_MAX_HW_QUEUES -- how many hardware queues ROCm maps the process's HIP streams onto -- is the CALLER's choice and must be made before
# the first GPU call: the scripts and bench.py ask for 8; see the note at the top of train_parent.py.)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libosvos_hip.so")
CSRC = os.path.join(_HERE, "csrc")
_lock = threading.Lock()
_lib = None

F32, BF16, F32_BF16MFMA, F32_X3 = 0, 1, 2, 3
DEFER_JOIN = 0x200         # OSVOS_FLAG_DEFER_JOIN: osvos_net_backward leaves the side streams un-joined (autograd.NetRuntime.join_backward)
X3_TWO_PIECES = 0x800       # OSVOS_FLAG_X3_TWO_PIECES: precision 'fp32x2' (two bf16 pieces per operand, three products)
X3_HALF_PIECES = 0x1000     # OSVOS_FLAG_X3_HALF_PIECES: precision 'fp32h2' (two FP16 pieces under block exponents, three products); on net_pack: forward packs
X3_HALF_PIECES_BWD = 0x2000 # OSVOS_FLAG_X3_HALF_PIECES_BWD: net_pack only -- data-gradient packs in the FP16-pair format
INFERENCE = 0x400           # OSVOS_FLAG_INFERENCE: osvos_net_forward writes nothing only a backward would read (sign bits, pool codes)
GENERIC_DECONV = 0x100      # OSVOS_FLAG_GENERIC_DECONV: OR-ed into the dtype of the osvos_net_* calls
NPARAMS = 52

_vp, _i, _l, _sz, _f = C.c_void_p, C.c_int, C.c_long, C.c_size_t, C.c_float

# name -> (restype, argtypes); every symbol declared in include/osvos_hip.h
PROTOTYPES = {
    "osvos_version": (_i, []),
    "osvos_last_error": (C.c_char_p, []),
    "osvos_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_wpack_bytes": (_sz, [_i, _i, _i]),
    "osvos_pack_conv3x3_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "osvos_wpack_dgrad_bytes": (_sz, [_i, _i, _i]),
    "osvos_pack_conv3x3_dgrad": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "osvos_conv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_conv3x3_num_tiles": (_i, []),
    "osvos_conv3x3_f32x3_tiles": (_i, []),
    "osvos_wpack_x3_bytes_abi": (_sz, [_i, _i, _i]),
    "osvos_pack_conv3x3_x3": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "osvos_conv3x3_x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_conv3x3_x3_streamk_ws_bytes": (_sz, []),
    "osvos_conv3x3_x3_streamk_ticket_bytes": (_sz, []),
    "osvos_conv3x3_x3_streamk": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "osvos_conv3x3_splitk_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "osvos_conv3x3_splitk": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "osvos_conv3x3_bf16io": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_conv3x3_wgrad_bf16act": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_maxpool2x2_bf16act": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_maxpool2x2_bwd_bf16act": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_maxpool2x2_bf16act_code": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_maxpool2x2_bwd_bf16act_code": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_conv3x3_bf16act_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_conv3x3_bf16io_tiles": (_i, [_vp, _i]),
    "osvos_set_x3_pieces": (_i, [_i]),
    "osvos_nchw_to_nhwc_bf16copy": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "osvos_maxpool2x2_bf16copy": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_maxpool2x2_bwd_bf16copy": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_conv3x3_dgrad_c3": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_conv3x3_dgrad_c3_bf16mma": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "osvos_conv3x3_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_maxpool2x2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "osvos_maxpool2x2_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "osvos_head_lowres": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "osvos_head_upsample": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "osvos_head_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_deconv_diag_check": (_i, [_vp, _i, _i, _vp, _vp]),
    "osvos_cbce": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "osvos_cbce_step": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _f, _vp, _vp]),
    "osvos_cbce_step_multi": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp]),
    "osvos_cbce_scratch_bytes": (_sz, [_i, _i, _i]),
    "osvos_cbce_step_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "osvos_scale": (_i, [_vp, _vp, _vp, _l, _vp]),
    "osvos_net_wbuf_bytes": (_sz, [_i]),
    "osvos_net_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "osvos_net_ws_bytes_infer": (_sz, [_i, _i, _i, _i]),
    "osvos_net_pack": (_i, [_vp, _vp, _i, _i, _vp]),
    "osvos_net_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "osvos_net_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "osvos_net_arm_grad_events": (_i, [_vp, _i]),
    "osvos_net_join": (_i, [_vp, _vp, _vp]),
    "osvos_comm_unique_id": (_i, [_vp]),
    "osvos_comm_init": (_i, [_vp, _i, _i, _vp]),
    "osvos_comm_allreduce_f32": (_i, [_vp, _vp, _sz, _vp]),
    "osvos_comm_allreduce_f64": (_i, [_vp, _vp, _sz, _vp]),
    "osvos_comm_allreduce_chunks_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "osvos_comm_destroy": (_i, [_vp]),
    "osvos_net_ws_query": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "osvos_net_ws_format": (_i, [_i, _i]),
    "osvos_augment_frame": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "osvos_mask_to_bytes": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "osvos_mask_iou_counts": (_i, [_vp, _vp, _vp, _l, _i, _f, _vp]),
    "osvos_sgd_step": (_i, [_vp, _vp, _vp, _l, _f, _f, _f, _i, _vp]),
    "osvos_sgd_step_multi": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp]),
    "osvos_prof_start": (_i, [_i]),
    "osvos_prof_stop": (_i, [_vp, _vp, _vp]),
    "osvos_prof_pause": (_i, [_i]),
    "osvos_debug_conv3x3_naive": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "osvos_debug_mfma_layout": (_i, [_vp, _vp]),
    "osvos_debug_mfma_peak": (_i, [_vp, _i, _i, _vp]),
    "osvos_debug_mfma_peak_bf16": (_i, [_vp, _vp, _i, _i, _vp]),
    "osvos_debug_set_c3_bf16": (_i, [_i]),
    "osvos_debug_lds_dma": (_i, [_vp, _i, _vp, _vp]),
}


def build(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("libosvos_hip.so is not built and hipcc was not found; run `make -C %s`" % CSRC)
    cmd = ["make", "-C", CSRC, "-j8", "-s", "HIPCC=" + hipcc]
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lk:      # ranks of one torchrun launch must not race in make
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if force:
                subprocess.check_call(cmd + ["clean"])
            subprocess.check_call(cmd)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return SO_PATH


def _stale() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".cpp", ".h")) and os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return os.path.getmtime(os.path.join(_HERE, "..", "include", "osvos_hip.h")) > t


def lib() -> C.CDLL:
    """The loaded library.  Builds it first when it is missing or older than its sources and a
    compiler is available; raises RuntimeError otherwise (never falls back to another path)."""
    global _lib
    with _lock:
        if _lib is None:
            # torch first: the process must end up with ONE HIP runtime (the libamdhip64 torch
            # ships); loading ours before torch's makes hipMemsetAsync & co see no device
            import torch  # noqa: F401
            if _stale() and os.environ.get("OSVOS_AUTOBUILD", "1") != "0":
                build()
            try:
                l = C.CDLL(SO_PATH)
            except OSError as e:  # pragma: no cover
                raise RuntimeError("cannot load %s: %s" % (SO_PATH, e))
            for name, (res, args) in PROTOTYPES.items():
                fn = getattr(l, name)          # AttributeError = ABI mismatch: fail loudly
                fn.restype = res
                fn.argtypes = args
            _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().osvos_last_error().decode(errors="replace")
        raise RuntimeError("libosvos_hip %s failed (rc=%d): %s" % (what, rc, msg))


def ptr_array(ptrs):
    """Host array of device pointers (ints or None)."""
    arr = (C.c_void_p * len(ptrs))(*[C.c_void_p(p) if p else None for p in ptrs])
    return arr
