"""ORACLE (test infrastructure only): ctypes binding of oracle/libosvos_oracle.so (plain C)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .torch_ref import state_dict_spec

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libosvos_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("osvos_oracle.c", "osvos_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.osvos_oracle_cbce_f32.restype = C.c_double
        _LIB.osvos_oracle_cbce_f64.restype = C.c_double
    return _LIB


def _suf(dtype):
    return "_f32" if np.dtype(dtype) == np.float32 else "_f64"


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def cbce(out, label, mode=1, want_grad=True):
    """mode 0 size_average / 1 batch_average / 2 none.  Returns (loss, grad)."""
    dt = out.dtype
    out = np.ascontiguousarray(out)
    label = np.ascontiguousarray(label, dtype=dt)
    grad = np.empty_like(out) if want_grad else None
    n = out.shape[0]
    f = getattr(lib(), "osvos_oracle_cbce" + _suf(dt))
    loss = f(_ptr(out), _ptr(label), _ptr(grad), C.c_int(n), C.c_long(out.size // n), C.c_int(mode))
    return loss, grad


def conv_fwd(x, w, b, relu):
    dt = x.dtype
    n, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    y = np.empty((n, cout, h, wd), dt)
    getattr(lib(), "osvos_oracle_conv_fwd" + _suf(dt))(
        _ptr(np.ascontiguousarray(x)), _ptr(np.ascontiguousarray(w)), _ptr(None if b is None else np.ascontiguousarray(b)),
        _ptr(y), n, cin, h, wd, cout, k, int(relu))
    return y


def conv_bwd(x, w, dy, need_dx=True):
    dt = x.dtype
    n, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    dx = np.empty_like(x) if need_dx else None
    dw = np.empty_like(w)
    db = np.empty((cout,), dt)
    getattr(lib(), "osvos_oracle_conv_bwd" + _suf(dt))(
        _ptr(np.ascontiguousarray(x)), _ptr(np.ascontiguousarray(w)), _ptr(np.ascontiguousarray(dy)),
        _ptr(dx), _ptr(dw), _ptr(db), n, cin, h, wd, cout, k)
    return dx, dw, db


def pool_fwd(x):
    dt = x.dtype
    n, c, h, w = x.shape
    y = np.empty((n, c, (h + 1) // 2, (w + 1) // 2), dt)
    arg = np.empty(y.shape, np.int32)
    getattr(lib(), "osvos_oracle_pool_fwd" + _suf(dt))(_ptr(np.ascontiguousarray(x)), _ptr(y), _ptr(arg), n, c, h, w)
    return y, arg


def pool_bwd(dy, arg, h, w):
    dt = dy.dtype
    n, c = dy.shape[:2]
    dx = np.zeros((n, c, h, w), dt)
    getattr(lib(), "osvos_oracle_pool_bwd_acc" + _suf(dt))(_ptr(np.ascontiguousarray(dy)), _ptr(arg), _ptr(dx), n, c, h, w)
    return dx


def deconv_fwd(x, w, s):
    dt = x.dtype
    n, cin, h, wd = x.shape
    _, cout, k, _ = w.shape
    y = np.empty((n, cout, (h - 1) * s + k, (wd - 1) * s + k), dt)
    getattr(lib(), "osvos_oracle_deconv_fwd" + _suf(dt))(_ptr(np.ascontiguousarray(x)), _ptr(np.ascontiguousarray(w)), _ptr(y), n, cin, cout, h, wd, k, s)
    return y


def net(weights, x, label=None, side_w=1.0, loss_scale=1.0, want_grads=False, want_dx=False, dtype=np.float32):
    """Whole network.  weights: mapping state_dict key -> array.  Returns dict with
    outs (5 arrays), losses (5,), grads (dict) and dx."""
    dt = np.dtype(dtype)
    spec = state_dict_spec()
    arrs = [np.ascontiguousarray(np.asarray(weights[k]), dtype=dt) for k, _ in spec]
    for a, (k, shp) in zip(arrs, spec):
        assert tuple(a.shape) == tuple(shp), (k, a.shape, shp)
    x = np.ascontiguousarray(x, dtype=dt)
    n, _, h, w = x.shape
    outs = [np.empty((n, 1, h, w), dt) for _ in range(5)]
    losses = np.zeros(5, np.float64)
    grads = [np.empty_like(a) for a in arrs] if want_grads else None
    dx = np.empty_like(x) if (want_grads and want_dx) else None
    lab = None if label is None else np.ascontiguousarray(label, dtype=dt)
    PP = C.c_void_p * 52
    f = getattr(lib(), "osvos_oracle_net" + _suf(dt))
    f(PP(*[_ptr(a) for a in arrs]), _ptr(x), _ptr(lab), n, h, w, C.c_double(side_w), C.c_double(loss_scale),
      (C.c_void_p * 5)(*[_ptr(o) for o in outs]), _ptr(losses),
      PP(*[_ptr(g) for g in grads]) if grads else None, _ptr(dx))
    res = {"outs": outs, "losses": losses}
    if grads:
        res["grads"] = {k: g for (k, _), g in zip(spec, grads)}
        res["dx"] = dx
    return res


def upsample_filt(size):
    out = np.empty((size, size), np.float64)
    lib().osvos_oracle_upsample_filt(size, _ptr(out))
    return out
