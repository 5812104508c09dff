"""CPU: size-independent properties of the oracle (test infrastructure checking itself), hypothesis-driven.  The golden
fixtures pin the oracle to the reference at fixed shapes (test_oracle_golden.py); these properties pin its structure at
random ones -- the same properties the GPU suite asserts of the HIP kernels at full size."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle


@settings(max_examples=25, deadline=None)
@given(rows=st.integers(1, 300), cols=st.integers(1, 40), seed=st.integers(0, 2**31 - 1))
def test_to_blocked_is_the_documented_permutation(rows, cols, seed):
    """out[(rb*CB + cb)*512 + (r%32)*16 + ((r%128)//32)*4 + c%4] == in[r, c], zero elsewhere (SURVEY.md 8a swizzle KAT)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(1, 256, size=(rows, cols), dtype=np.uint8)          # non-zero so that padding is distinguishable
    pr, pc = oracle.padded_shape(rows, cols)
    padded = np.zeros((pr, pc), np.uint8)
    padded[:rows, :cols] = x
    out = oracle.to_blocked(padded).reshape(-1)
    assert out.size == pr * pc
    r, c = np.meshgrid(np.arange(pr), np.arange(pc), indexing="ij")
    idx = ((r // 128) * (pc // 4) + c // 4) * 512 + (r % 32) * 16 + ((r % 128) // 32) * 4 + c % 4
    assert np.array_equal(out[idx], padded)
    assert sorted(idx.reshape(-1).tolist()) == list(range(pr * pc))     # a permutation of the padded matrix


def test_e2m1_grid_round_trips_and_rounds_to_nearest_even():
    grid = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]
    for code in range(16):
        v = oracle.e2m1_decode(code)
        assert abs(v) == grid[code & 7] and (v < 0 or code < 8 or v == 0)
        assert oracle.e2m1_encode(v) & 7 == code & 7
    # ties go to the even mantissa, saturation at 6 (SURVEY.md 8a e2m1 KATs)
    for x, want in [(0.25, 0.0), (0.75, 1.0), (1.25, 1.0), (1.75, 2.0), (2.5, 2.0), (-3.5, -4.0), (5.0, 4.0), (7.5, 6.0), (100.0, 6.0), (-0.26, -0.5)]:
        assert oracle.e2m1_decode(oracle.e2m1_encode(x)) == want, x


@settings(max_examples=15, deadline=None)
@given(m=st.integers(1, 40), n8=st.integers(1, 6), kb=st.integers(1, 4), seed=st.integers(0, 2**31 - 1), kind=st.sampled_from([0, 1]))
def test_gemm_oracle_alpha_linearity_and_row_permutation(m, n8, kb, seed, kind):
    """D(alpha / 2) == D(alpha) / 2 exactly (power of two), rows of A permute rows of D, and an all-zero operand gives 0."""
    kinds = [oracle.KIND_MXFP4, oracle.KIND_NVFP4]
    knd, gs = kinds[kind], (32, 16)[kind]
    n, k = 8 * n8, 128 * kb
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(m, k // 2), dtype=np.uint8)
    b = rng.integers(0, 256, size=(n, k // 2), dtype=np.uint8)
    lo, hi = (125, 130) if knd == oracle.KIND_MXFP4 else (0x30, 0x48)
    pr_a, pc = oracle.padded_shape(m, k // gs)
    pr_b, _ = oracle.padded_shape(n, k // gs)
    sa = np.zeros((pr_a, pc), np.uint8)
    sb = np.zeros((pr_b, pc), np.uint8)
    sa[:m, : k // gs] = rng.integers(lo, hi, size=(m, k // gs), dtype=np.uint8)
    sb[:n, : k // gs] = rng.integers(lo, hi, size=(n, k // gs), dtype=np.uint8)
    f = lambda aa, ssa, alpha: oracle.bf16_bits_to_f32(oracle.gemm_blockscaled(knd, aa, b, oracle.to_blocked(ssa), oracle.to_blocked(sb), alpha, m, n, k))
    d1, dh = f(a, sa, 1.0), f(a, sa, 0.5)
    assert np.array_equal(dh, d1 * 0.5)
    perm = rng.permutation(m)
    sap = sa.copy()
    sap[:m] = sa[perm]
    assert np.array_equal(f(np.ascontiguousarray(a[perm]), sap, 1.0), d1[perm])
    assert not f(np.zeros_like(a), sa, 1.0).any()


@settings(max_examples=10, deadline=None)
@given(rows=st.integers(1, 9), seed=st.integers(0, 2**31 - 1), method=st.sampled_from([oracle.QUEST, oracle.ABS_MAX]))
def test_quantizer_oracle_scale_equivariance_and_code_range(rows, seed, method):
    """Scaling the input by 2^3 (exact in bf16) shifts every e8m0 by 3 and leaves the codes alone (up to the +1e-8 epsilon,
    irrelevant at these magnitudes); dequantised abs-max values never exceed amax * 3 / 2 in magnitude per group."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, 128)) * 8).astype(np.float32)
    bf = lambda v: (v.view(np.uint32) >> 16).astype(np.uint16)           # truncation is fine here: both inputs share it
    xb = bf(x)
    xs = bf((oracle.bf16_bits_to_f32(xb) * 8.0).astype(np.float32))
    h = np.eye(32, dtype=np.float32)
    hb = bf(h)
    q1, s1 = oracle.fused_quantize_mx(xb, hb, method)[:2]
    q2, s2 = oracle.fused_quantize_mx(xs, hb, method)[:2]
    assert np.array_equal(s2.astype(np.int32), s1.astype(np.int32) + 3)
    assert np.array_equal(q1, q2)
    assert q1.dtype == np.uint8 and q1.size == rows * 128 // 2
