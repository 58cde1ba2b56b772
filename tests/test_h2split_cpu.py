"""CPU restatement of the FP16-pair split of csrc/h2split.h (precisions 'fp32h2' / 'fp32x3h2'; round 6) -- the arithmetic the GPU kernels are
held to in tests/test_gpu_ops.py::test_f32x3_kernels_with_fp16_pairs, checked here in numpy so that the bounds the design quotes are the
bounds the formula has: block exponent from the largest magnitude (h2_exp), v 2^e = h + m + r with both pieces round-to-nearest-even fp16,
no overflow, |r| <= max(2^-22 |v 2^e|, 2^-25), and the three-product form ah bh + ah bm + am bh within 2^-21 of the exact product.
(The convolution this serves: reference vgg_osvos.py:41,136-145 and its autograd.)"""
import numpy as np
import pytest


def h2_exp(amax):
    """csrc/h2split.h: exponent field of the block's largest |value|, clamped to [15, 253]; the maximum lands in [2^14, 2^15)"""
    bits = np.float32(amax).view(np.uint32) & np.uint32(0x7fffffff)
    e = int(bits >> np.uint32(23))
    return 141 - min(max(e, 15), 253)


def split(v, e):
    s = v.astype(np.float32) * np.float32(2.0) ** e               # exact: a power of two
    h = s.astype(np.float16)                                      # RNE
    m = (s - h.astype(np.float32)).astype(np.float16)             # the subtraction is exact in fp32
    return s, h, m


@pytest.mark.parametrize("scale", [1.0, 3e-9, 2.5e6, 1e-30, 1e30])
def test_two_fp16_pieces_carry_22_bits_and_never_overflow(scale):
    rng = np.random.RandomState(3)
    v = (rng.randn(200000) * scale).astype(np.float32)
    v[:1000] *= np.exp(4 * rng.randn(1000)).astype(np.float32)                # a heavy tail inside the block
    e = h2_exp(np.abs(v).max())
    s, h, m = split(v, e)
    assert 2.0 ** 14 <= np.abs(s).max() < 2.0 ** 15 and np.isfinite(h).all() and np.isfinite(m).all()
    r = np.abs(s.astype(np.float64) - h.astype(np.float64) - m.astype(np.float64))
    assert (r <= np.maximum(2.0 ** -22 * np.abs(s.astype(np.float64)), 2.0 ** -25)).all()
    big = np.abs(s) >= 0.25                                                    # down to 2^-17 of the block maximum: full relative precision
    assert (r[big] <= 2.0 ** -22 * np.abs(s[big])).all() and np.sqrt(np.mean((r[big] / np.abs(s[big])) ** 2)) < 2.0 ** -24


def test_three_products_reproduce_the_fp32_product_to_2_pow_minus_21():
    rng = np.random.RandomState(4)
    a = rng.randn(100000).astype(np.float32) * 7.0
    b = rng.randn(100000).astype(np.float32) * 0.03
    ea, eb = h2_exp(np.abs(a).max()), h2_exp(np.abs(b).max())
    _, ah, am = split(a, ea)
    _, bh, bm = split(b, eb)
    f = lambda x: x.astype(np.float64)
    three = (f(ah) * f(bh) + f(ah) * f(bm) + f(am) * f(bh)) * 2.0 ** -(ea + eb)      # what the three MFMAs accumulate, un-scaled (exact powers of two)
    exact = f(a) * f(b)
    rel = np.abs(three - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -21 and np.sqrt(np.mean(rel ** 2)) < 2.0 ** -23
    # every fp16 x fp16 product is exact in fp32 (11 + 11 significand bits), so the matrix pipe adds nothing to this
    assert (f(ah) * f(bh) == (ah.astype(np.float32) * bh.astype(np.float32)).astype(np.float64)).all()


def test_block_exponent_edges():
    assert h2_exp(0.0) == 126 and h2_exp(np.float32(1e-45)) == 126              # zeros / fp32 subnormals: clamped, 2^126 * tiny stays finite
    assert h2_exp(np.float32(3.0e38)) == 141 - 253                              # near FLT_MAX: scaled down, 2^-e still a normal fp32
    for amax in (1.0, 0.75, 1.5, 65504.0, 1e-7):
        e = h2_exp(np.float32(amax))
        assert 2.0 ** 14 <= amax * 2.0 ** e < 2.0 ** 15
