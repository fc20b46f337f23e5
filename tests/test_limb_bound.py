"""The a-priori error bound of the default conv kernel's two-limb product (k_conv_x2.hip; DESIGN.md 3.3 "Round 6"), checked in numpy on the host: fp16 arithmetic of
numpy is IEEE round-to-nearest-even like v_cvt_f16_f32, products of fp16 numbers are exact in fp64 (and in the kernel's fp32 accumulator).  No GPU, no library call:
this pins the STATEMENT the kernel's accuracy claim rests on (models/tensor_layers.py:140-143,154-155 are plain fp32 GEMMs); the kernel itself is measured against
the fp64 oracle in tests/test_gpu_round6.py::test_two_limb_kernel_is_fp32_grade."""
import numpy as np
import pytest


def _adversarial(rng, n, binades):
    """fp32 values with every mantissa bit in play, values one ulp around powers of two and around fp16 rounding boundaries (hi ties, mid ties), mixed signs,
    spread over `binades` binades below the group's maximum"""
    m = rng.integers(1 << 23, 1 << 24, size=n).astype(np.float64)
    m[::7] = (1 << 23) + rng.integers(0, 3, size=m[::7].shape)
    m[1::7] = (1 << 24) - 1 - rng.integers(0, 3, size=m[1::7].shape)
    m[2::7] = ((rng.integers(1 << 10, 1 << 11, size=m[2::7].shape) << 13) | (1 << 12)) + rng.integers(-1, 2, size=m[2::7].shape)
    m[3::7] = (rng.integers(1 << 10, 1 << 11, size=m[3::7].shape) << 13) | ((1 << 12) + (1 << 1) - 1) | (rng.integers(0, 2, size=m[3::7].shape) << 1)
    e = rng.integers(-binades, 1, size=n)
    return (rng.choice([-1.0, 1.0], size=n) * m * np.exp2(e.astype(np.float64) - 23)).astype(np.float32)


def _range_scale(x, group):
    """the kernel's exact power-of-two scaling: the maximum of every group of `group` values lands in [2^14, 2^15)"""
    g = np.abs(x.astype(np.float64)).reshape(-1, group).max(axis=1)
    e = np.floor(np.log2(np.maximum(g, 2.0 ** -40)))
    return np.repeat(np.exp2(14 - e), group)


def _two_limbs(xs):
    hi = xs.astype(np.float32).astype(np.float16)
    mid = (xs.astype(np.float32) - hi.astype(np.float32)).astype(np.float16)          # (the subtraction is exact in fp32: Sterbenz-like, hi is x rounded to 11 bits)
    return hi.astype(np.float64), mid.astype(np.float64)


@pytest.mark.parametrize('binades', [3, 14, 30])
def test_two_limbs_carry_an_operand_to_2_to_the_minus_22(binades):
    rng = np.random.default_rng(5)
    x = _adversarial(rng, 72 * 4000, binades)
    xs = x.astype(np.float64) * _range_scale(x, 72)
    hi, mid = _two_limbs(xs)
    err = np.abs(xs - hi - mid)
    # relative 2^-22 wherever mid is a normal fp16 number (|x| >= 2^-3 after scaling), absolute 2^-25 (= 2^-39 of the group's maximum) below that
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(xs), 2.0 ** -25))
    assert np.all(np.abs(hi) < 65504) and np.all(np.abs(mid) <= 2.0 ** -11 * np.abs(xs) * (1 + 2.0 ** -10) + 2.0 ** -25)


def test_three_limb_products_stay_under_the_fp32_dot_product_bound():
    """hi.hi + hi.mid + mid.hi against the exact product: <= 3 * 2^-22 relative per product (two operand truncations + the dropped mid.mid), so a K = 72 dot
    product - plus the <= 14 roundings of its MFMA chain in the fp32 accumulator - stays under the classical a-priori bound 72 * 2^-24 * sum |a_i b_i| of an fp32 FMA
    chain of that length (what the reference's own fp32 GEMM guarantees)."""
    rng = np.random.default_rng(6)
    K, n = 72, 20000
    a = _adversarial(rng, n * K, 3)
    b = _adversarial(rng, n * K, 3)
    as_, bs_ = a.astype(np.float64) * _range_scale(a, K), b.astype(np.float64) * _range_scale(b, K)
    ah, am = _two_limbs(as_)
    bh, bm = _two_limbs(bs_)
    p3 = ah * bh + ah * bm + am * bh                         # exact in fp64: every term is a product of two 11-bit numbers
    exact = as_ * bs_
    rel = np.abs(p3 - exact) / np.abs(exact)
    assert rel.max() <= 3 * 2.0 ** -22 * (1 + 2.0 ** -9)
    assert rel.mean() < 2.0 ** -23                           # typical: a third of the bound
    dot3, dote, sabs = p3.reshape(n, K).sum(1), exact.reshape(n, K).sum(1), np.abs(exact).reshape(n, K).sum(1)
    trunc = np.abs(dot3 - dote) / sabs
    chain_roundings = 14 * 2.0 ** -24                        # one rounding per MFMA of the accumulator chain (each adds its K range exactly)
    assert trunc.max() + chain_roundings < 72 * 2.0 ** -24
    # ... and an fp32 FMA chain of the same operands, for scale: its error is of the same order (this is what `conv_kernel = 1` and the reference's CPU GEMM do)
    acc = np.zeros(n, dtype=np.float32)
    a2, b2 = as_.astype(np.float32).reshape(n, K), bs_.astype(np.float32).reshape(n, K)
    for k in range(K):
        acc = (acc.astype(np.float64) + a2[:, k].astype(np.float64) * b2[:, k].astype(np.float64)).astype(np.float32)      # fused multiply-add: one rounding per step
    chain = np.abs(acc.astype(np.float64) - dote) / sabs
    assert np.median(trunc) < 4 * np.median(chain) + 2.0 ** -26


@pytest.mark.parametrize('binades', [3, 14])
def test_emulated_kernel_arithmetic_is_at_least_as_accurate_as_an_fp32_fma_chain(binades):
    """The kernel's GEMM arithmetic restated bit for bit on the host - per K step of 16 the three MFMAs hi.mid, mid.hi, hi.hi into ONE fp32 accumulator (an MFMA adds the
    products of its K range exactly and rounds once), the packed K = 8 tail's two - against an fp32 FMA chain over the same 72 operands (one rounding per step: what
    `conv_kernel = 1` and the reference's CPU GEMM do), both against the exact dot product, on adversarial operands.  Relative to sum |a_i b_i|: the kernel's arithmetic is
    not worse than the chain at the median, the 99th percentile and the maximum (measured: median 1.27e-8 vs 1.35e-8, p99 5.5e-8 vs 8.6e-8, max 9.4e-8 vs 1.9e-7 at 3 binades)."""
    rng = np.random.default_rng(6 + binades)
    K, n = 72, 20000
    a, b = _adversarial(rng, n * K, binades), _adversarial(rng, n * K, binades)
    as_, bs_ = a.astype(np.float64) * _range_scale(a, K), b.astype(np.float64) * _range_scale(b, K)
    ah, am = [v.reshape(n, K) for v in _two_limbs(as_)]
    bh, bm = [v.reshape(n, K) for v in _two_limbs(bs_)]
    exact = (as_ * bs_).reshape(n, K)
    sabs, dote = np.abs(exact).sum(1), exact.sum(1)
    f32 = lambda v: v.astype(np.float32).astype(np.float64)
    acc = np.zeros(n)
    for s in range(4):
        sl = slice(16 * s, 16 * s + 16)
        for A, B in ((ah, bm), (am, bh), (ah, bh)):                                    # X3_STEP of k_conv_x.hip under X3_TWO_LIMBS
            acc = f32(acc + (A[:, sl] * B[:, sl]).sum(1))
    sl = slice(64, 72)
    acc = f32(acc + (ah[:, sl] * bm[:, sl]).sum(1) + (am[:, sl] * bh[:, sl]).sum(1))      # X3_TAIL3: {W_hi, W_mid} x {h_mid, h_hi}
    acc = f32(acc + (ah[:, sl] * bh[:, sl]).sum(1) + (am[:, sl] * bm[:, sl]).sum(1))      #           {W_hi, W_mid} x {h_hi, h_mid}
    kern = np.abs(acc - dote) / sabs
    c = np.zeros(n, dtype=np.float32)
    a2, b2 = as_.astype(np.float32).reshape(n, K), bs_.astype(np.float32).reshape(n, K)
    for k in range(K):
        c = (c.astype(np.float64) + a2[:, k].astype(np.float64) * b2[:, k].astype(np.float64)).astype(np.float32)
    chain = np.abs(c.astype(np.float64) - dote) / sabs
    print(f'binades {binades}: kernel arithmetic median {np.median(kern):.2e} p99 {np.quantile(kern, .99):.2e} max {kern.max():.2e} | fp32 FMA chain median {np.median(chain):.2e} '
          f'p99 {np.quantile(chain, .99):.2e} max {chain.max():.2e}')
    assert np.median(kern) <= 1.05 * np.median(chain) and np.quantile(kern, .99) <= np.quantile(chain, .99) and kern.max() <= chain.max()
    assert kern.max() < 72 * 2.0 ** -24
