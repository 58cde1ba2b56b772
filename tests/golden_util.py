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


def check_grad_either(g, mode, key, arr, rtol_l2, what=""):
    """A gradient must match what the REAL reference produced for it in float32 (its CPU path) OR in float64 (truth) -- both
    are stored.  On these un-trained nets the reference's own float32 result can sit several 1e-3 from its float64 result on
    the stage-0 / input gradients (one ReLU or arg-max flip at a near-tie; its NCHW and channels_last code paths disagree by
    the same amount), so an implementation whose round-off lands on the float64 side of such a flip is right, not wrong."""
    try:
        check_grad(g, "f32|%s|grad|" % mode, key, arr, rtol_l2, what=what)
    except AssertionError as e32:
        if "f64|%s|grad|%s|l2" % (mode, key) not in g.files:
            raise
        try:
            check_grad(g, "f64|%s|grad|" % mode, key, arr, rtol_l2, what=what + " (float64 reference)")
        except AssertionError as e64:
            raise AssertionError("matches neither the reference's float32 nor its float64 gradient: %s ; %s" % (e32, e64))
