"""Round-6 GPU parity tests (nothing here reads /root/reference).

  * Quest on CONSTANT groups: the sign of the fp32 variance decides the reference's arm -- MX: `if (var >= 0)` else scale 1.0 (epilogue_quant.h:531-535); NV: no
    guard at all, sqrt of the negative variance is stored as the NaN scale byte 0x7f and every code of the group becomes +-0 (epilogue_quant.h:1631-1640).  The kernel
    re-sums such groups in the reference's order (quantize.hip.h `quest_sums_in_reference_order`): with a rotation that is exact (c I) the scale bytes are compared
    EXACTLY against the oracle's sequential sums; with a Hadamard rotation only the kind of disagreement is checked.
  * `backward_qt_bf16` on a tile whose 32 rows ALL carry scale byte 0 (operands 0.0 whatever their codes, quartet_bwd_sm120.cu:369-375): amax 0 -> scale 2^-127 ... as the oracle.
  * compiled callers: `aot_eager` and `inductor` graphs of the quantize -> swizzle -> GEMM layer and of the QAT-backward data-prep ops return the eager bytes (the wrappers
    call ops whose schemas declare what they write).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)

DEV = "cuda:0"


@pytest.fixture(scope="module")
def q():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import qutlass_amd

    return qutlass_amd


def _np(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.uint16).numpy()
    if t.element_size() == 1:
        return t.view(torch.uint8).numpy()
    return t.numpy()


def _hadamard(n: int) -> torch.Tensor:
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(DEV)


def constant_groups(n: int, rot: int, seed: int) -> torch.Tensor:
    """n rotation groups of ONE repeated bf16 value each; in the second half of them a tenth of the elements have the last mantissa bit flipped (also built by the CPU half,
    tests/test_round6_cpu.py).  Rotated by c I with c = 1.7109375 every output is x c EXACTLY (one product per output, 16 significant bits), so kernel and oracle see the
    same y -- and y^2 has 32 significant bits, so every `fma(y, y, s2)` rounds: the sums depend on the order of the additions and the fp32 variance of such a group is
    rounding noise around 0, negative for about a third of them."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(n).astype(np.float32) * 8
    bits = torch.from_numpy(np.repeat(a[:, None], rot, 1)).to(torch.bfloat16).view(torch.int16).numpy().copy()
    flip = rng.random((n, rot)) < 0.1
    flip[: n // 2] = False
    return torch.from_numpy(bits ^ flip.astype(np.int16)).view(torch.bfloat16)


def scaled_identity(rot: int) -> torch.Tensor:
    return (torch.eye(rot) * 1.7109375).to(torch.bfloat16).to(DEV)


def nearly_constant_groups(n: int, rot: int, seed: int) -> torch.Tensor:
    """n rotation groups (a, a 2^-k, 0, ..., 0), k = 9 .. 14: rotated by a Hadamard matrix every output is +-(a +- a 2^-k) / sqrt(rot) -- nearly constant, with long mantissas.
    Here the rotation itself rounds (the matrix pipe's accumulation is not the oracle's, nor the reference's tensor core's): which of these groups get a negative variance
    cannot be pinned, only that the two outcomes are the ones a vanishing variance has."""
    rng = np.random.default_rng(seed)
    x = np.zeros((n, rot), np.float32)
    a = rng.standard_normal(n).astype(np.float32) * 8
    x[:, 0] = a
    x[:, 1] = a * (2.0 ** -rng.integers(9, 15, size=n))
    return torch.from_numpy(x).to(torch.bfloat16)


@pytest.mark.parametrize("rot", [16, 32, 64, 128])
def test_fused_quantize_nv_quest_constant_groups_nan_scale_byte_exact(q, rot):
    x = constant_groups(4096, rot, 3).to(DEV)
    h = scaled_identity(rot)
    gs = torch.tensor([1.0], device=DEV)
    e2m1, e4m3 = q.fusedQuantizeNv(x, h, gs, method="quest")
    rq, rs = oracle.fused_quantize_nv(_np(x), _np(h), 1.0, oracle.QUEST)
    rs = np.asarray(rs).reshape(-1)
    got_s = _np(e4m3).reshape(-1)[: rs.size]
    assert int((rs == 0x7F).sum()) > rs.size // 8          # the reference's NaN scale byte (epilogue_quant.h:1631-1640), about a quarter of these groups
    bad = np.nonzero(got_s != rs)[0]
    assert bad.size == 0, f"{bad.size} e4m3 scale bytes differ, first groups {bad[:8]}: got {got_s[bad[:8]]}, oracle {rs[bad[:8]]}"
    got_q = _np(e2m1).reshape(-1, 8)
    assert not (got_q[rs == 0x7F] & 0x77).any(), "a group with a NaN scale byte holds codes other than +-0"
    eq = oracle.codes_equal_mod_zero_sign(_np(e2m1), rq)
    assert int((~eq).sum()) <= 2, f"{int((~eq).sum())} code bytes differ"


@pytest.mark.parametrize("rot", [32, 64, 128])
@pytest.mark.parametrize("mask", [False, True])
def test_fused_quantize_mx_quest_constant_groups_exact(q, rot, mask):
    if mask and rot != 32:
        pytest.skip("the clip-mask quantizer takes rotation 32 only (fused_quantize_mx_mask.cu:107-123)")
    x = constant_groups(4096, rot, 4).to(DEV)
    h = scaled_identity(rot)
    out = q.fusedQuantizeMx(x, h, method="quest", return_mask=mask)
    rq, rs, rm = oracle.fused_quantize_mx(_np(x), _np(h), oracle.QUEST, with_mask=mask)
    rs = np.asarray(rs).reshape(-1)
    got_s = _np(out[1]).reshape(-1)[: rs.size]
    assert int((rs == 127).sum()) > rs.size // 8           # the `var < 0` arm: scale 1.0 (epilogue_quant.h:531-535)
    _, rs_lane, _ = oracle.fused_quantize_mx(_np(x), _np(h), oracle.QUEST, acc_model=2)
    assert int((np.asarray(rs_lane).reshape(-1) != rs).sum()) > rs.size // 8   # ... and the kernel's natural lane order would get a third of them wrong
    bad = np.nonzero(got_s != rs)[0]
    assert bad.size == 0, f"{bad.size} e8m0 bytes differ, first groups {bad[:8]}: got {got_s[bad[:8]]}, oracle {rs[bad[:8]]}"
    eq = oracle.codes_equal_mod_zero_sign(_np(out[0]), rq)
    assert eq.all(), f"{int((~eq).sum())} code bytes differ"
    if mask:
        assert np.array_equal(_np(out[2]).reshape(-1), np.asarray(rm).reshape(-1))


@pytest.mark.parametrize("rot", [16, 32, 64])
def test_fused_quantize_nv_quest_nearly_constant_rotated_groups(q, rot):
    """Hadamard-rotated near-constant groups: kernel and oracle may disagree on the SIGN of a vanishing variance (the rotation's own rounding), never on anything else: a
    differing scale byte is 0x7f (NaN: negative variance) on one side and the byte of sqrt(~0) * c + 1e-8 -- 0 or 1 -- on the other."""
    x = nearly_constant_groups(4096, rot, 6).to(DEV)
    h = _hadamard(rot)
    gs = torch.tensor([1.0], device=DEV)
    e2m1, e4m3 = q.fusedQuantizeNv(x, h, gs, method="quest")
    _, rs = oracle.fused_quantize_nv(_np(x), _np(h), 1.0, oracle.QUEST)
    rs = np.asarray(rs).reshape(-1)
    got_s = _np(e4m3).reshape(-1)[: rs.size]
    bad = got_s != rs
    assert set(np.unique(np.stack([got_s[bad], rs[bad]])).tolist()) <= {0, 1, 0x7F}, (got_s[bad][:8], rs[bad][:8])   # (1 = 2^-9, the smallest e4m3: a variance of a few ulps)
    if rot == 32:   # (1 / sqrt(32) is not a bf16: long mantissas -- about a fifth of these groups get a negative variance, on either side)
        assert int((got_s == 0x7F).sum()) > 500 and int((rs == 0x7F).sum()) > 500
    assert not (_np(e2m1).reshape(-1, 8)[got_s == 0x7F] & 0x77).any()


@pytest.mark.parametrize("B,N,M", [(1, 96, 64), (1, 4096 + 96, 6144 + 32), (2, 512, 1024)])   # round-3 kernel / wave-owned lines / the ring kernel (M % 128 == 0)
def test_backward_qt_bf16_whole_tiles_of_scale_byte_0(q, B, N, M):
    """[ADVICE r5] every row of some [32 n] tiles carries scale byte 0: the operands are +-0.0, the rotated group has amax 0 and quantises as the reference's 0 / 0 does
    (oracle orc_backward_qt_bf16) -- with the hardware convert's scale operand alone (0.0f reads as 2^-127 there) the codes came out +-6."""
    rng = np.random.default_rng(B + N + M)
    codes = rng.integers(0, 256, size=(B, N, M // 2), dtype=np.uint8)
    scales = rng.integers(118, 134, size=(B, N, M // 32), dtype=np.uint8)
    scales[:, 32:64, :] = 0                 # a whole band of 32 rows: every tile of it
    scales[:, 0:32, M // 64] = 0            # one tile column of the first band
    scales[:, N - 32:N, 0] = 0
    h = _hadamard(32)
    alpha = torch.tensor([0.61], device=DEV)
    e2m1, e8m0 = q.backward_qt_bf16(torch.from_numpy(codes).to(DEV), torch.from_numpy(scales).to(DEV).view(torch.float8_e8m0fnu), h, alpha)
    rq, rs = oracle.backward_qt_bf16(codes, scales, _np(h), 0.61, acc_model=1)
    got_s = _np(e8m0).reshape(rs.shape)
    assert np.array_equal(got_s, rs), f"{int((got_s != rs).sum())} scale bytes differ"
    eq = oracle.codes_equal_mod_zero_sign(_np(e2m1).reshape(rq.shape), rq)
    assert int((~eq).sum()) <= 1e-4 * eq.size, int((~eq).sum())


# ------------------------------------------------------------------------------------------------
# compiled == eager, with the backends that functionalise and eliminate dead code
# ------------------------------------------------------------------------------------------------
def _compile_backends():
    out = ["aot_eager"]
    try:   # inductor needs a working triton for the pointwise glue; the ops themselves are extern calls either way
        import triton  # noqa: F401
        out.append("inductor")
    except Exception:
        pass
    return out


def test_compiled_callers_return_the_eager_bytes(q):
    from qutlass_amd.utils import to_blocked

    def layer(x, h, wq, wsf, alpha):
        xq, xs = q.fusedQuantizeMx(x, h, method="abs_max")
        y = q.matmul_mxf4_bf16_tn(xq.view(-1, xq.size(-1)), wq, to_blocked(xs), wsf, alpha)
        return y @ y.t(), xq        # a torch op AFTER the custom ops: with undeclared writes inductor reused the quantizer's buffers for it

    def prep(g, h):
        a, b = q.backward_t_bf16(g, h)
        y, rs, cs = q.backward_bf16_square_double_mxfp8(g.view(-1, g.size(-1)))
        return a.view(torch.uint8) + 0, b.view(torch.uint8) + 0, y.view(torch.uint8) + 0, rs.view(torch.uint8) + 0, cs.view(torch.uint8) + 0

    def qt(c, s, h, alpha):
        a, b = q.backward_qt_bf16(c, s, h, alpha)
        y, ys = q.mxfp4_transpose_mxfp8(c.view(-1, c.size(-1)), s.view(-1, s.size(-1)))
        return a.view(torch.uint8) + 0, b.view(torch.uint8) + 0, y.view(torch.uint8) + 0, ys.view(torch.uint8) + 0

    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(2, 160, 512, dtype=torch.bfloat16, device=DEV, generator=g) * 5
    w = torch.randn(384, 512, dtype=torch.bfloat16, device=DEV, generator=g)
    h = _hadamard(32)
    wq, ws = q.fusedQuantizeMx(w, h, method="abs_max")
    wsf = to_blocked(ws)
    alpha = torch.tensor([0.5], device=DEV)
    grad = torch.randn(2, 256, 384, dtype=torch.bfloat16, device=DEV, generator=g)
    rng = np.random.default_rng(1)
    c = torch.from_numpy(rng.integers(0, 256, size=(1, 256, 128), dtype=np.uint8)).to(DEV)
    s = torch.from_numpy(rng.integers(120, 130, size=(1, 256, 8), dtype=np.uint8)).to(DEV).view(torch.float8_e8m0fnu)
    cases = ((layer, (x, h, wq, wsf, alpha)), (prep, (grad, h)), (qt, (c, s, h, alpha)))
    for fn, args in cases:
        eager = [t.clone() for t in fn(*args)]
        for backend in _compile_backends():
            torch._dynamo.reset()
            got = torch.compile(fn, backend=backend, fullgraph=True)(*args)
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(got, eager)):
                av = a.view(torch.uint8) if a.element_size() == 1 else a.view(torch.int16)
                bv = b.view(torch.uint8) if b.element_size() == 1 else b.view(torch.int16)
                assert torch.equal(av, bv), (fn.__name__, backend, i)


# ------------------------------------------------------------------------------------------------
# [r6] the 8-wave persistent MXFP4 kernel (csrc/lab/gemm_mx_duo.hip.h, LAB library only; qutlass/csrc/gemm.cu:174-248): same products, same K order -> the SAME bits as the 4-wave
# persistent kernel and every other schedule, on full tiles, ragged edges, K tails, odd stage counts, one tile and several tiles per workgroup
# ------------------------------------------------------------------------------------------------
import _benchlib as lab  # noqa: E402  (the LAB library: test infrastructure)


def _mx_operands(m, n, k, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=DEV, generator=g)
    b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=DEV, generator=g)
    pad = lambda r: (r + 127) // 128 * 128
    cb = (k // 32 + 3) // 4 * 4
    sa = torch.randint(121, 130, (pad(m) * cb,), dtype=torch.uint8, device=DEV, generator=g)
    sb = torch.randint(121, 130, (pad(n) * cb,), dtype=torch.uint8, device=DEV, generator=g)
    return a, b, sa, sb


DUO_SHAPES = [(256, 256, 512), (512, 768, 256), (4096, 4096, 4096), (1000, 1288, 1408), (2048, 6400, 1280), (264, 8, 3968), (8192, 8192, 768), (300, 520, 11008)]


@pytest.mark.parametrize("variant", [88, 87])
@pytest.mark.parametrize("m,n,k", DUO_SHAPES)
def test_duo_kernel_equals_the_4wave_persistent_kernel(variant, m, n, k):
    a, b, sa, sb = _mx_operands(m, n, k, m + n + k)
    alpha = torch.tensor([0.75], device=DEV)
    with lab.forced(gemm_variant=90):
        ref = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
    with lab.forced(gemm_variant=variant):
        got = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
    torch.cuda.synchronize()
    bad = (got.view(torch.int16) != ref.view(torch.int16))
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} outputs differ, first at {bad.nonzero()[0].tolist()}"


@pytest.mark.parametrize("variant", [88, 87])
def test_duo_kernel_against_the_oracle(variant):
    m, n, k = 320, 264, 1408      # edges in M and N, a K tail (5.5 stages of 256), an odd number of stages
    a, b, sa, sb = _mx_operands(m, n, k, 5)
    alpha = torch.tensor([1.0], device=DEV)
    with lab.forced(gemm_variant=variant):
        got = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a), _np(b), _np(sa), _np(sb), 1.0, m, n, k)
    assert np.array_equal(_np(got), ref), int((_np(got) != ref).sum())


# ------------------------------------------------------------------------------------------------
# [r6] the in-workgroup K-split kernel (csrc/gemm_mx_ks.hip.h; qutlass/csrc/gemm.cu:195-222, gemm_ada.cu:127-129): every tile / ring depth against the oracle on
# ragged shapes (M, N not multiples of the tile, K tails of half a stage, fewer stages than the ring is deep), and the product's rule (capi.hip ks_plan) on the
# shapes it takes -- operands in the exact regime (scale exponents within +-2: every partial sum is exact, any summation order gives the same bits)
# ------------------------------------------------------------------------------------------------
def _mx_operands_exact(m, n, k, seed):
    a, b, sa, sb = _mx_operands(m, n, k, seed)
    g = torch.Generator(device=DEV).manual_seed(seed + 1)
    sa = torch.randint(125, 129, sa.shape, dtype=torch.uint8, device=DEV, generator=g)
    sb = torch.randint(125, 129, sb.shape, dtype=torch.uint8, device=DEV, generator=g)
    return a, b, sa, sb


KS_SHAPES = [(1, 8, 128), (9, 40, 384), (33, 104, 1408), (64, 264, 4096), (100, 72, 640), (31, 4096, 4096), (64, 2048, 8192), (130, 520, 256)]


@pytest.mark.parametrize("variant", [561, 562, 563, 564, 565, 566, 567, 568, 569, 570, 571, 572, 573, 574, 575])   # (571 ... 575: the decode form on the 16x16x128 MFMA, 16 / 32 / 48 / 56 / 64 columns per workgroup)
@pytest.mark.parametrize("m,n,k", KS_SHAPES + [(5, 72, 1024), (40, 200, 2944), (64, 96, 3072), (9, 136, 4224), (33, 72, 11008), (3, 40, 14336), (64, 64, 4352),
                                               (128, 4096, 4096), (97, 136, 3328), (160, 72, 5376)])   # (568: one shot up to K = 4096, wave-owned rings beyond; 570: 64x32 tiles, one shot up to K = 3072)
def test_ks_kernel_against_the_oracle(variant, m, n, k):
    a, b, sa, sb = _mx_operands_exact(m, n, k, m * 7 + n + k)
    alpha = torch.tensor([0.5], device=DEV)
    with lab.forced(gemm_variant=variant):
        got = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a), _np(b), _np(sa), _np(sb), 0.5, m, n, k)
    bad = _np(got) != ref
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} outputs differ, first at {np.argwhere(bad)[0].tolist()}"


@pytest.mark.parametrize("m,n,k", [(1, 8, 128), (5, 72, 1024), (33, 104, 1408), (40, 200, 2944), (64, 96, 3072), (31, 264, 4096), (17, 8192, 3968), (9, 136, 4224), (33, 72, 11008), (3, 40, 14336),
                                   (128, 4096, 2048), (72, 8192, 1280)])   # (the last two: 64x32 tiles by the product rule, capi.hip os64_plan)
def test_one_shot_kernel_with_row_major_scales_against_the_oracle(q, m, n, k):
    """matmul_ada_mxf4_bf16_tn (qutlass/csrc/gemm_ada.cu; row-major scale operands) on shapes the product sends to the one-shot kernel (csrc/gemm_mx_os.hip.h, RM form):
    ragged M / N, K tails of half a stage, 1 ... 16 stages; the same operands through the blocked-scale entry (one-shot kernel, blocked form) and the 64x64 ring kernel."""
    g = torch.Generator(device=DEV).manual_seed(m * 11 + n + k)
    a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=DEV, generator=g)
    b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=DEV, generator=g)
    sa = torch.randint(125, 129, (m, k // 32), dtype=torch.uint8, device=DEV, generator=g)
    sb = torch.randint(125, 129, (n, k // 32), dtype=torch.uint8, device=DEV, generator=g)
    alpha = torch.tensor([0.5], device=DEV)
    e8 = torch.float8_e8m0fnu
    got = q.matmul_ada_mxf4_bf16_tn(a, b, sa.view(e8), sb.view(e8), alpha)
    sa_b, sb_b = oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a), _np(b), sa_b, sb_b, 0.5, m, n, k)
    bad = _np(got) != ref
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} outputs differ, first at {np.argwhere(bad)[0].tolist()}"
    tsa, tsb = torch.from_numpy(sa_b).to(DEV), torch.from_numpy(sb_b).to(DEV)
    assert np.array_equal(_np(q.matmul_mxf4_bf16_tn(a, b, tsa.view(e8), tsb.view(e8), alpha)), ref)
    with lab.forced(gemm_variant=70):   # the ring kernel with row-major scale fetch, the plan before this kernel
        assert np.array_equal(_np(lab.matmul_ada_mxf4_bf16_tn(a, b, sa, sb, alpha)), ref)
    for v in (568, 569, 570, 571, 572, 574, 575):   # ... and the kernel itself (32 / 16 columns per workgroup, 64-row tiles, the 16x16 decode form) where the product rule does not send the shape to it
        with lab.forced(gemm_variant=v):
            assert np.array_equal(_np(lab.matmul_ada_mxf4_bf16_tn(a, b, sa, sb, alpha)), ref), v


@pytest.mark.parametrize("m,n,k", [(1, 4096, 4096), (8, 8192, 4096), (48, 4096, 4096), (64, 4096, 14336), (16, 14336, 4096), (200, 1024, 2048),
                                   (16, 12288, 4096), (9, 11008, 5120), (128, 4096, 2048), (4, 16384, 4096)])   # (the last four: decode forms with 48 / 48 / - / 64 columns, 64-row tiles)
def test_product_rule_takes_the_ks_kernel_and_matches_the_oracle(q, m, n, k):
    """shapes capi.hip's ks_plan sends to the new kernel (tests/test_cabi_and_host.py pins the plan on the CPU), through the product library and the torch op"""
    a, b, sa, sb = _mx_operands_exact(m, n, k, m + n + k)
    alpha = torch.tensor([1.0], device=DEV)
    got = q.matmul_mxf4_bf16_tn(a, b, sa.view(torch.float8_e8m0fnu), sb.view(torch.float8_e8m0fnu), alpha)
    rows = np.unique(np.linspace(0, m - 1, min(m, 24)).astype(np.int64))
    cb = (k // 32 + 3) // 4 * 4
    # the oracle on a sample of rows (their scale rows gathered out of the blocked image)
    a_rows = _np(a)[rows]
    sa_img = _np(sa).reshape(-1)
    def blocked_row(img, r):   # (qutlass/utils.py:60-64) byte (r, c) of the to_blocked image
        c = np.arange(k // 32)
        return img[((r // 128) * (cb // 4) + c // 4) * 512 + (r % 32) * 16 + ((r % 128) // 32) * 4 + c % 4]
    sa_rm = np.stack([blocked_row(sa_img, int(r)) for r in rows])
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, a_rows, _np(b), oracle.to_blocked(sa_rm), _np(sb), 1.0, len(rows), n, k)
    assert np.array_equal(_np(got)[rows], ref)


# ------------------------------------------------------------------------------------------------
# [r6] an ODD number of K stages in the persistent 4-wave kernel (gemm_mx_deepp.hip.h ODD; K = 11008 is in the reference's own shape list, tests/mxfp4_test.py:194-199):
# no empty stage any more -- tiles alternate their starting LDS buffer, so the walk is forced onto FEW workgroups (lab option deepp_grid) to make every workgroup
# run tiles of both parities, and compared with the ring kernel (another schedule, same K order) and the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k,grid", [(512, 768, 768, 2), (512, 768, 1280, 1), (700, 520, 3840, 3), (1024, 1024, 11008, 5), (4096, 4096, 3840, 0), (256, 256, 11008 + 128, 1),
                                         (4096, 5120, 1280, 0), (2304, 1024, 768, 8)])   # (the last two with variant 98 too: the heterogeneous launch)
def test_persistent_kernel_with_an_odd_number_of_k_stages(m, n, k, grid):
    a, b, sa, sb = _mx_operands(m, n, k, m + n + k)
    alpha = torch.tensor([0.75], device=DEV)
    with lab.forced(gemm_variant=73):
        ref = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
    with lab.forced(gemm_variant=90, deepp_grid=grid):
        got = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
    torch.cuda.synchronize()
    bad = got.view(torch.int16) != ref.view(torch.int16)
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} outputs differ, first at {bad.nonzero()[0].tolist()}"
    if (m, n) in ((4096, 5120), (2304, 1024)):   # persistent workgroups + residual quarter tiles in one grid
        with lab.forced(gemm_variant=98, deepp_grid=grid):
            het = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
        assert torch.equal(het.view(torch.int16), ref.view(torch.int16))
    if m * n <= 1 << 20:
        a2, b2, sa2, sb2 = _mx_operands_exact(m, n, k, 3)
        with lab.forced(gemm_variant=90, deepp_grid=grid):
            got2 = lab.matmul_mxf4_bf16_tn(a2, b2, sa2, sb2, alpha)
        rows = slice(0, min(m, 40))
        cb = (k // 32 + 3) // 4 * 4
        ref2 = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a2), _np(b2), _np(sa2), _np(sb2), 0.75, m, n, k)
        assert np.array_equal(_np(got2), ref2)


# ------------------------------------------------------------------------------------------------
# [r6] the wave-owned small-batch NVFP4 kernel (csrc/gemm_nvf4_os.hip.h; qutlass/csrc/gemm.cu:250-326): 32 / 16 columns per workgroup (lab variants 46 / 47), one shot
# (K <= 4096) and wave-owned rings (longer K), ragged M / N, K tails of half a stage and of half a column tile (K % 64 == 32), against the oracle; scale bytes in the
# exact regime (e4m3 0x30 ... 0x47: every partial sum exact, any summation order gives the same bits) -- and the product rule on the shapes it sends there
# ------------------------------------------------------------------------------------------------
NVOS_SHAPES = [(1, 8, 32), (5, 72, 1024), (33, 104, 1440), (40, 200, 2944), (64, 96, 3104), (31, 264, 4096), (17, 4096, 3968), (9, 136, 4224), (33, 72, 11040), (3, 40, 14336)]


@pytest.mark.parametrize("variant", [46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 0])   # (50 ... 52: the decode form with 32 / 48 / 56 columns per workgroup; 53: 56 columns, A rows 0 ... 7 only -- M <= 8; 54 / 55: the 32x32-MFMA kernel with two / three m-tiles per workgroup)   # (48: the 16x16 decode form on v_mfma_f32_16x16x32_f16, one shot up to K = 8192; 49: two m-tiles per workgroup, up to K = 4096)
@pytest.mark.parametrize("m,n,k", NVOS_SHAPES + [(16, 4096, 4096), (7, 264, 8192), (12, 136, 8480), (8, 14336, 4096), (5, 392, 1056), (16, 616, 2080), (128, 520, 4096), (100, 264, 2080), (200, 136, 4128)])
def test_nvf4_wave_owned_kernel_against_the_oracle(variant, m, n, k):
    from qutlass_amd.utils import to_blocked

    if variant == 53 and m > 8:
        pytest.skip("the 8-row A tile: M <= 8")
    rng = np.random.default_rng(m * 13 + n + k)
    a = torch.from_numpy(rng.integers(0, 256, size=(m, k // 2), dtype=np.uint8)).to(DEV)
    b = torch.from_numpy(rng.integers(0, 256, size=(n, k // 2), dtype=np.uint8)).to(DEV)
    sa = torch.from_numpy(rng.integers(0x30, 0x48, size=(m, k // 16), dtype=np.uint8)).to(DEV)
    sb = torch.from_numpy(rng.integers(0x30, 0x48, size=(n, k // 16), dtype=np.uint8)).to(DEV)
    alpha = torch.tensor([0.5], device=DEV)
    e4 = torch.float8_e4m3fn
    with lab.forced(nvf4_variant=variant):
        got = lab.matmul_nvf4_bf16_tn(a, b, to_blocked(sa.view(e4)), to_blocked(sb.view(e4)), alpha)
    ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 0.5, m, n, k)
    bad = _np(got) != ref
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} outputs differ, first at {np.argwhere(bad)[0].tolist()}"


# ------------------------------------------------------------------------------------------------
# [r6] the wave-owned small-batch kernel on MXFP8 operands (csrc/gemm_mx_os.hip.h, EBITS = 8; qutlass/csrc/gemm.cu:328-386 on small batches) -- 32x32 / 32x16 / 64x32 tiles
# (lab variants 568 / 569 / 570), one shot (K <= 2048 / 1536) and wave-owned rings (longer K), ragged M / N, K tails of a quarter stage (K % 128 == 32), e4m3 and e5m2 A
# operands: (a) within the MXFP8 tolerance class of the fp64 oracle on quantised Gaussian operands (tests/mxfp8_test.py), (b) EXACTLY equal to the oracle on operands whose
# products and partial sums are all exact in fp32 (codes from {0, +-0.5, +-1, +-1.5, +-2, +-3, +-4}, scale exponents within +-2) -- the second pins every byte's place in
# the fragment and every scale byte's block
# ------------------------------------------------------------------------------------------------
def _mxfp8_close(got_bits, want_bits):
    got = oracle.bf16_bits_to_f32(got_bits).astype(np.float64)
    want = oracle.bf16_bits_to_f32(want_bits).astype(np.float64)
    tol = np.abs(want) / 128.0 + 2e-5 * np.abs(want).max()
    return np.abs(got - want) <= tol


MXF8_OS_SHAPES = [(1, 8, 32), (5, 72, 1024), (33, 104, 1440), (40, 200, 1952), (64, 96, 2048), (31, 264, 4096), (17, 4096, 2016), (9, 136, 4224), (33, 72, 5536), (3, 40, 7168),
                  (128, 520, 4096), (97, 136, 1536), (100, 72, 1568)]


def _exact_fp8_operands(m, n, k, seed, e5m2_a=False):
    rng = np.random.default_rng(seed)
    vals = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0], np.float32)
    af = torch.from_numpy(vals[rng.integers(0, len(vals), size=(m, k))])
    bf = torch.from_numpy(vals[rng.integers(0, len(vals), size=(n, k))])
    a = af.to(torch.float8_e5m2 if e5m2_a else torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    b = bf.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    sa = torch.from_numpy(rng.integers(125, 130, size=(m, k // 32), dtype=np.uint8)).to(DEV)
    sb = torch.from_numpy(rng.integers(125, 130, size=(n, k // 32), dtype=np.uint8)).to(DEV)
    return a, b, sa, sb


@pytest.mark.parametrize("a5", [False, True])
@pytest.mark.parametrize("variant", [568, 569, 570, 571, 572, 573, 574, 575, 0])
@pytest.mark.parametrize("m,n,k", MXF8_OS_SHAPES)
def test_mxf8_wave_owned_kernel_exact_against_the_oracle(variant, m, n, k, a5):
    from qutlass_amd.utils import to_blocked

    a, b, sa, sb = _exact_fp8_operands(m, n, k, m * 13 + n + k, a5)
    alpha = torch.tensor([0.5], device=DEV)
    e8 = torch.float8_e8m0fnu
    tsa, tsb = to_blocked(sa.view(e8)).view(torch.uint8), to_blocked(sb.view(e8)).view(torch.uint8)
    with lab.forced(gemm_variant=variant):
        got = lab.matmul_mxf8_bf16_tn_fmt(a, b, tsa, tsb, alpha, a_format=1 if a5 else 0)
    kind = oracle.KIND_MXFP8_TN_A5 if a5 else oracle.KIND_MXFP8_TN
    ref = oracle.gemm_blockscaled(kind, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 0.5, m, n, k)
    bad = _np(got) != ref
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} outputs differ, first at {np.argwhere(bad)[0].tolist()}"


@pytest.mark.parametrize("variant", [568, 569, 570, 571])
@pytest.mark.parametrize("m,n,k", [(33, 104, 1440), (31, 264, 4096), (128, 520, 4096), (9, 136, 4224)])
def test_mxf8_wave_owned_kernel_on_quantised_gaussians(variant, m, n, k):
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, dtype=torch.bfloat16) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16) * 25.0
    aq, asf = oracle.pseudoquant_mxfp8(_np(a))
    bq, bsf = oracle.pseudoquant_mxfp8(_np(b))
    e8 = torch.float8_e8m0fnu
    with lab.forced(gemm_variant=variant):
        out = lab.matmul_mxf8_bf16_tn(torch.from_numpy(aq).to(DEV), torch.from_numpy(bq).to(DEV), to_blocked(torch.from_numpy(asf).to(DEV).view(e8)),
                                      to_blocked(torch.from_numpy(bsf).to(DEV).view(e8)), torch.tensor([1.0], device=DEV))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, aq, bq, oracle.to_blocked(asf), oracle.to_blocked(bsf), 1.0, m, n, k)
    assert _mxfp8_close(_np(out), ref).all()


@pytest.mark.parametrize("m,n,k", [(1, 4096, 4096), (16, 4096, 4096), (64, 4096, 4096), (128, 4096, 4096), (32, 8192, 2048), (8, 2048, 8192), (8, 8192, 4096), (16, 11008, 4096)])
def test_product_rule_on_mxf8_small_batches_matches_the_oracle(q, m, n, k):
    """shapes the product sends to the wave-owned kernel (capi.hip os8_plan; pinned on the CPU in tests/test_cabi_and_host.py), through the product library and the torch op,
    exact-regime operands: every output element equal to the oracle's"""
    from qutlass_amd.utils import to_blocked

    a, b, sa, sb = _exact_fp8_operands(m, n, k, m + n + k)
    e4, e8 = torch.float8_e4m3fn, torch.float8_e8m0fnu
    got = q.matmul_mxf8_bf16_tn(a.view(e4), b.view(e4), to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), torch.tensor([1.0], device=DEV))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 1.0, m, n, k)
    assert np.array_equal(_np(got), ref)


# ------------------------------------------------------------------------------------------------
# [r6] decode forms through the PRODUCT rules on ragged wide weights: M = 1 ... 16 (17 ... 32 for the NVFP4 32x16 form), N anywhere in 4104 ... 16384 (a multiple of 8, so the last
# 32 / 48 / 56 / 64-column workgroup is part-filled and its B rows / scale rows fall off the operand), K inside and outside the one-shot range -- exact-regime operands, every
# output element against the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", ["mxf4", "ada", "mxf8", "nvf4"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_decode_forms_on_ragged_wide_weights(q, fmt, seed):
    from qutlass_amd.utils import to_blocked

    rng = np.random.default_rng(1000 * seed + len(fmt))
    e8, e4 = torch.float8_e8m0fnu, torch.float8_e4m3fn
    for _ in range(3):
        m = int(rng.integers(1, 33 if fmt == "nvf4" else 17))
        n = int(rng.integers(513, 2049)) * 8
        k = int(rng.choice([1024, 2048, 3072, 4096, 5120, 8192])) if fmt != "mxf8" else int(rng.choice([1024, 2048, 4096, 4128]))
        alpha = torch.tensor([0.5], device=DEV)
        if fmt in ("mxf4", "ada"):
            a = torch.from_numpy(rng.integers(0, 256, size=(m, k // 2), dtype=np.uint8)).to(DEV)
            b = torch.from_numpy(rng.integers(0, 256, size=(n, k // 2), dtype=np.uint8)).to(DEV)
            sa = torch.from_numpy(rng.integers(125, 129, size=(m, k // 32), dtype=np.uint8)).to(DEV)
            sb = torch.from_numpy(rng.integers(125, 129, size=(n, k // 32), dtype=np.uint8)).to(DEV)
            if fmt == "ada":
                got = q.matmul_ada_mxf4_bf16_tn(a, b, sa.view(e8), sb.view(e8), alpha)
            else:
                got = q.matmul_mxf4_bf16_tn(a, b, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), alpha)
            ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 0.5, m, n, k)
        elif fmt == "mxf8":
            a, b, sa, sb = _exact_fp8_operands(m, n, k, int(rng.integers(0, 1 << 30)))
            got = q.matmul_mxf8_bf16_tn(a.view(e4), b.view(e4), to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), alpha)
            ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 0.5, m, n, k)
        else:
            a = torch.from_numpy(rng.integers(0, 256, size=(m, k // 2), dtype=np.uint8)).to(DEV)
            b = torch.from_numpy(rng.integers(0, 256, size=(n, k // 2), dtype=np.uint8)).to(DEV)
            sa = torch.from_numpy(rng.integers(0x30, 0x48, size=(m, k // 16), dtype=np.uint8)).to(DEV)
            sb = torch.from_numpy(rng.integers(0x30, 0x48, size=(n, k // 16), dtype=np.uint8)).to(DEV)
            got = q.matmul_nvf4_bf16_tn(a, b, to_blocked(sa.view(e4)), to_blocked(sb.view(e4)), alpha)
            ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 0.5, m, n, k)
        bad = _np(got) != ref
        assert not bad.any(), f"{fmt} {m}x{n}x{k}: {int(bad.sum())} of {bad.size} outputs differ, first at {np.argwhere(bad)[0].tolist()}"


@pytest.mark.parametrize("m,n,k", [(300, 2048, 2048), (384, 2056, 2048), (24, 28672, 4096), (160, 4096, 8192), (96, 6144, 4096), (128, 4104, 4128), (192, 4096, 4096), (72, 8192, 2080)])
def test_nvf4_priced_forms_through_the_product_rule(q, m, n, k):
    """[r6] shapes nvf4_plan sends to the wave-owned kernel in several rounds of 32x32 tiles or to its 64x32 / 96x32 forms (gemm_nvf4.hip.h: the priced candidates; the CPU half pins
    the plan), through the product library and the torch op: exact-regime scale bytes, sampled rows against the oracle and every row against the lab library's 128x128 tiles"""
    from qutlass_amd.utils import to_blocked

    rng = np.random.default_rng(m + n + k)
    a = torch.from_numpy(rng.integers(0, 256, size=(m, k // 2), dtype=np.uint8)).to(DEV)
    b = torch.from_numpy(rng.integers(0, 256, size=(n, k // 2), dtype=np.uint8)).to(DEV)
    sa = torch.from_numpy(rng.integers(0x38, 0x48, size=(m, k // 16), dtype=np.uint8)).to(DEV)
    sb = torch.from_numpy(rng.integers(0x38, 0x48, size=(n, k // 16), dtype=np.uint8)).to(DEV)
    e4 = torch.float8_e4m3fn
    alpha = torch.tensor([0.5], device=DEV)
    sa_b, sb_b = to_blocked(sa.view(e4)), to_blocked(sb.view(e4))
    got = q.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, alpha)
    with lab.forced(nvf4_variant=5):   # 128x128 tiles, one pass
        single = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, alpha)
    assert torch.equal(got.view(torch.int16), single.view(torch.int16))
    rows = sorted({0, 31, 32, 63, 64, 95, 96, m // 2, m - 1} & set(range(m)))
    ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, np.ascontiguousarray(_np(a)[rows]), _np(b), oracle.to_blocked(np.ascontiguousarray(_np(sa)[rows])), oracle.to_blocked(_np(sb)), 0.5, len(rows), n, k)
    assert np.array_equal(_np(got)[rows], ref)
