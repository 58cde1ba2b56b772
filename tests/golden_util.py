"""Helpers shared by the golden-vector tests (fixtures made by tests/golden/make_golden.py)."""
import os
from collections import OrderedDict

import numpy as np

from oracle import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c16x16", "c37x53_n2", "c48x64", "c30x85"]


def load_case(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n, h, w, seed, wseed = [int(v) for v in g["meta"]]
    x = synth.make_frame(n, h, w, seed)
    m = synth.make_mask(n, h, w, seed)
    wts = synth.make_weights(wseed)
    for k in g.files:
        if k.startswith("head|"):
            wts[k[5:]] = g[k].copy()
    return g, wts, x, m


def grad_keys(g, prefix):
    keys = OrderedDict()
    for k in g.files:
        if k.startswith(prefix) and k.endswith("|sum"):
            keys[k[len(prefix):-4]] = True
    return list(keys)


def check_grad(g, prefix, key, arr, rtol_l2, what=""):
    """Compare a full gradient array with the stored (sum, l2, samples) summary."""
    a = np.asarray(arr, np.float64).ravel()
    l2 = float(g[prefix + key + "|l2"])
    idx = g[prefix + key + "|idx"]
    val = g[prefix + key + "|val"]
    scale = max(l2 / np.sqrt(a.size), 1e-30)
    assert abs(np.sqrt((a * a).sum()) - l2) <= rtol_l2 * l2 + 1e-30, (what, key, "l2", np.sqrt((a * a).sum()), l2)
    err = np.abs(a[idx] - val).max()
    assert err <= 20 * rtol_l2 * scale + rtol_l2 * np.abs(val).max(), (what, key, "samples", err, scale)


def reference_results_differ(g, mode, key, size, rtol_l2):
    """Do the REAL reference's own float32 and float64 gradients of `key` differ by more than rtol_l2 (relative l2, or the stored samples
    relative to the largest one)?  True only for a handful of stage-0..2 / input tensors of `c30x85` (an arg-max / ReLU flip at a
    near-tie: 11 of its 41 online keys, 7 of its 53 parent keys); every other golden gradient has ONE accepted answer."""
    p32, p64 = "f32|%s|grad|%s" % (mode, key), "f64|%s|grad|%s" % (mode, key)
    if p64 + "|l2" not in g.files or not np.array_equal(g[p32 + "|idx"], g[p64 + "|idx"]):
        return False
    l32, l64 = float(g[p32 + "|l2"]), float(g[p64 + "|l2"])
    v32, v64 = g[p32 + "|val"], g[p64 + "|val"]
    return bool(abs(l32 - l64) > rtol_l2 * l64 or np.abs(v32 - v64).max() > rtol_l2 * np.abs(v64).max())


def check_grad_either(g, mode, key, arr, rtol_l2, what=""):
    """A gradient must match what the REAL reference produced for it in float32 (its CPU path).  Only where the reference's own float32
    and float64 results disagree by more than the bar (reference_results_differ: an arg-max or ReLU flip at a near-tie on these
    un-trained nets; its NCHW and channels_last code paths disagree by the same amount) may it match the float64 result instead --
    an implementation whose round-off lands on the float64 side of such a flip is right, not wrong.  Everywhere else the float32
    golden value is the only accepted answer."""
    if not reference_results_differ(g, mode, key, np.asarray(arr).size, rtol_l2):
        return check_grad(g, "f32|%s|grad|" % mode, key, arr, rtol_l2, what=what)
    try:
        check_grad(g, "f32|%s|grad|" % mode, key, arr, rtol_l2, what=what)
    except AssertionError as e32:
        try:
            check_grad(g, "f64|%s|grad|" % mode, key, arr, rtol_l2, what=what + " (float64 reference)")
        except AssertionError as e64:
            raise AssertionError("matches neither the reference's float32 nor its float64 gradient: %s ; %s" % (e32, e64))
