"""Round-5 GPU parity tests: the semantic EDGES of the path, against the pinned CPU oracle (nothing here reads /root/reference).

  * non-finite / special ACTIVATIONS through the fused quantizers (NaN, +-inf, -0, bf16 denormals, the largest finite bf16): the reference defines them through
    `cvt.rn.satfinite.e2m1x2.f32` (NaN -> 0x7, +-inf -> +-6) and its scale rule (epilogue_quant.h:77-97, :520-571; oracle/qutlass_oracle.c:63-76, :230-295).
  * one decode of an e4m3 scale byte: 0x7f / 0xff are NaN (OCP e4m3fn, what `scales.float()` gives in tests/nvfp4_test.py:196-203) in EVERY NVFP4 kernel the
    dispatch can pick, so the same call cannot change meaning with its shape (qutlass/csrc/gemm.cu:250-326 has one semantics).
  * e8m0 bytes 0 and 255 in the operands of the QAT-backward data-prep ops (quartet_bwd_sm120.cu:369-375, :503-509, :580-586).
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


def _bf16_from_bits(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.astype(np.uint16)).view(torch.bfloat16)


def _hadamard(n: int) -> torch.Tensor:
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(DEV)


# bf16 bit patterns: NaN, +inf, -inf, -0, smallest / largest denormal (both signs), largest finite (both signs)
SPECIALS = {"nan": 0x7FC0, "nan_payload": 0xFFC1, "+inf": 0x7F80, "-inf": 0xFF80, "-0": 0x8000, "den_min": 0x0001, "-den_max": 0x807F, "max": 0x7F7F, "-max": 0xFF7F}


def _special_activations(rows, cols, rot, seed):
    """random bf16 rows with ONE special value per chosen rotation group (so that each special is seen alone), a group holding +inf AND -inf, one holding NaN and inf,
    an all-denormal group, an all -0 group; the untouched groups are the control"""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, cols)).astype(np.float32) * 3.0)
    bits = _np(torch.from_numpy(x).to(torch.bfloat16)).copy().reshape(-1)
    ngroups = bits.size // rot
    names = list(SPECIALS)
    for i, nm in enumerate(names):
        g = 2 * i + 1
        bits[g * rot + (5 * i + 3) % rot] = SPECIALS[nm]
    g = 2 * len(names) + 1
    bits[g * rot + 1], bits[g * rot + 17 % rot] = SPECIALS["+inf"], SPECIALS["-inf"]
    bits[(g + 2) * rot + 2], bits[(g + 2) * rot + 9] = SPECIALS["nan"], SPECIALS["+inf"]
    bits[(g + 4) * rot:(g + 5) * rot] = rng.integers(1, 0x80, rot) | (rng.integers(0, 2, rot) << 15)
    bits[(g + 6) * rot:(g + 7) * rot] = SPECIALS["-0"]
    assert g + 7 <= ngroups
    return _bf16_from_bits(bits.reshape(rows, cols)).to(DEV)


@pytest.mark.parametrize("rot", [32, 64, 128])
@pytest.mark.parametrize("method", ["abs_max", "quest"])
@pytest.mark.parametrize("ident", [True, False])
def test_fused_quantize_mx_special_activations(q, rot, method, ident):
    """identity rotation: each special value stays in its element; Hadamard: a NaN / inf spreads over its whole rotation group (inf - inf = NaN) -- both must give the
    oracle's scale bytes, codes and clip mask."""
    x = _special_activations(8, 1024, rot, 11 * rot + (method == "quest"))
    h = torch.eye(rot, dtype=torch.bfloat16, device=DEV) if ident else _hadamard(rot)
    mask = rot == 32 and method == "quest"
    out = q.fusedQuantizeMx(x, h, method=method, return_mask=mask)
    rq, rs, rm = oracle.fused_quantize_mx(_np(x), _np(h), oracle.QUEST if method == "quest" else oracle.ABS_MAX, with_mask=mask)
    n = x.numel()
    got_s = _np(out[1]).reshape(-1)[: n // 32]
    bad = np.nonzero(got_s != rs)[0]
    assert bad.size == 0, f"e8m0 differs in groups {bad[:8]}: got {got_s[bad[:8]]}, oracle {rs[bad[:8]]}"
    eq = oracle.codes_equal_mod_zero_sign(_np(out[0]), rq)
    assert eq.all(), f"{int((~eq).sum())} code bytes differ, first at byte {int(np.nonzero(~eq.reshape(-1))[0][0])}"
    if mask:
        assert np.array_equal(_np(out[2]).reshape(-1), rm)


@pytest.mark.parametrize("rot", [16, 32, 128])
@pytest.mark.parametrize("method", ["abs_max", "quest"])
def test_fused_quantize_nv_special_activations(q, rot, method):
    x = _special_activations(8, 1024, rot, 7 * rot + (method == "quest"))
    h = torch.eye(rot, dtype=torch.bfloat16, device=DEV)
    gs = torch.tensor([1.5], device=DEV)
    e2m1, e4m3 = q.fusedQuantizeNv(x, h, gs, method=method)
    rq, rs = oracle.fused_quantize_nv(_np(x), _np(h), 1.5, oracle.QUEST if method == "quest" else oracle.ABS_MAX)
    n = x.numel()
    got_s = _np(e4m3).reshape(-1)[: n // 16]
    bad = np.nonzero(got_s != rs)[0]
    assert bad.size == 0, f"e4m3 scale differs in groups {bad[:8]}: got {got_s[bad[:8]]}, oracle {rs[bad[:8]]}"
    eq = oracle.codes_equal_mod_zero_sign(_np(e2m1), rq)
    assert int((~eq).sum()) <= 2, f"{int((~eq).sum())} code bytes differ"   # (the reference's rcp.approx vs the oracle's exact reciprocal: a tie may fall either way)


# ------------------------------------------------------------------------------------------------
# one decode of an e4m3 scale byte for every NVFP4 kernel class
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k", [(4, 512, 512), (32, 4096, 1024), (96, 256, 256), (128, 512, 512), (300, 1024, 768), (1024, 2048, 512), (4096, 4096, 512)])
def test_matmul_nvf4_nan_scale_bytes_decode_the_same_in_every_kernel(q, m, n, k):
    """Scale bytes 0x7f and 0xff are NaN in OCP e4m3fn.  Rows r % 5 == 1 of A carry 0x7f in one group, rows r % 5 == 3 carry 0xff; column c % 7 == 2 of B carries
    0x7f: exactly those outputs are NaN, every other output equals the oracle's -- whatever kernel the shape is dispatched to (skinny, small tiles, split-K, persistent)."""
    from qutlass_amd.utils import to_blocked

    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, generator=g)
    b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, generator=g)
    sa = torch.randint(0x38, 0x48, (-(-m // 128) * 128, k // 16), dtype=torch.uint8, generator=g)
    sb = torch.randint(0x38, 0x48, (-(-n // 128) * 128, k // 16), dtype=torch.uint8, generator=g)
    ra = torch.arange(sa.shape[0])
    sa[ra % 5 == 1, 3] = 0x7F
    sa[ra % 5 == 3, (k // 16) - 1] = 0xFF
    rb = torch.arange(sb.shape[0])
    sb[rb % 7 == 2, 1] = 0x7F
    out = q.matmul_nvf4_bf16_tn(a.to(DEV), b.to(DEV), to_blocked(sa.to(DEV).view(torch.float8_e4m3fn)), to_blocked(sb.to(DEV).view(torch.float8_e4m3fn)), torch.tensor([1.0], device=DEV))
    got = out.float().cpu().numpy()
    nan_expected = np.zeros((m, n), bool)
    nan_expected[(np.arange(m) % 5 == 1) | (np.arange(m) % 5 == 3), :] = True
    nan_expected[:, np.arange(n) % 7 == 2] = True
    assert np.array_equal(np.isnan(got), nan_expected), f"NaN pattern differs in {int((np.isnan(got) != nan_expected).sum())} outputs"
    rows = list(range(min(m, 64)))
    pad = np.zeros((128 - len(rows), k // 16), np.uint8)
    ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, np.ascontiguousarray(a.numpy()[rows]), b.numpy(), oracle.to_blocked(np.concatenate([sa.numpy()[rows], pad])),
                                  oracle.to_blocked(sb.numpy()), 1.0, len(rows), n, k)
    gu, ru = _np(out)[rows].view(np.uint16), ref.view(np.uint16)
    ok = nan_expected[rows]
    assert np.array_equal(gu[~ok], ru[~ok])


# ------------------------------------------------------------------------------------------------
# e8m0 bytes 0 and 255 in the operands of the backward data-prep ops
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,M", [(1, 96, 64), (1, 4096 + 96, 6144 + 32)])     # the round-3 kernel / the wave-owned-lines kernel (product rule)
def test_backward_qt_bf16_with_e8m0_bytes_0_and_255(q, B, N, M):
    """scale byte 0 is the bf16 with bits 0x0000 (e8m0 << 7: the operand is 0 whatever its code), byte 255 is +inf (operand +-inf, or NaN for code 0):
    quartet_bwd_sm120.cu:369-375 multiplies the decoded code with that bf16; the oracle restates it (oracle/qutlass_oracle.c orc_backward_qt_bf16)."""
    rng = np.random.default_rng(N + M)
    codes = rng.integers(0, 256, size=(B, N, M // 2), dtype=np.uint8)
    scales = rng.integers(118, 134, size=(B, N, M // 32), dtype=np.uint8)
    scales[:, 5::64, 0] = 0
    scales[:, 40::64, 1] = 255
    scales[:, 7::96, -1] = 0
    h = _hadamard(32)
    alpha = torch.tensor([0.61], device=DEV)
    e2m1, e8m0 = q.backward_qt_bf16(torch.from_numpy(codes).to(DEV), torch.from_numpy(scales).to(DEV).view(torch.float8_e8m0fnu), h, alpha)
    rq, rs = oracle.backward_qt_bf16(codes, scales, _np(h), 0.61, acc_model=1)
    got_s = _np(e8m0).reshape(rs.shape)
    assert np.array_equal(got_s, rs), f"{int((got_s != rs).sum())} scale bytes differ"
    eq = oracle.codes_equal_mod_zero_sign(_np(e2m1).reshape(rq.shape), rq)
    assert int((~eq).sum()) <= 1e-4 * eq.size, int((~eq).sum())


@pytest.mark.parametrize("m,n", [(128, 256), (256, 512), (2048, 2304)])
def test_mxfp4_transpose_mxfp8_with_e8m0_bytes_0_and_255(q, m, n):
    """byte 0 -> 2^-127 (bf16 bits 0x0040, `__nv_cvt_e8m0_to_bf16raw`), byte 255 -> bits 0x7f80 = +inf (operand +-inf, NaN for code 0): quartet_bwd_sm120.cu:628-712;
    oracle orc_mxfp4_transpose_mxfp8.  (m is a multiple of 256 or padded inside the kernel with zero codes / unit scales: the oracle gets the padded operand.)"""
    rng = np.random.default_rng(m + n)
    codes = rng.integers(0, 256, size=(m, n // 2), dtype=np.uint8)
    scales = rng.integers(117, 137, size=(m, n // 32), dtype=np.uint8)
    scales[3::32, 0] = 0
    scales[17::64, 2] = 0
    scales[9::64, 1] = 255
    y, sf = q.mxfp4_transpose_mxfp8(torch.from_numpy(codes).to(DEV), torch.from_numpy(scales).to(DEV).view(torch.float8_e8m0fnu))
    m_pad = -(-m // 256) * 256
    pc = np.zeros((m_pad, n // 2), np.uint8); pc[:m] = codes
    ps = np.full((m_pad, n // 32), 127, np.uint8); ps[:m] = scales
    ry, rs = oracle.mxfp4_transpose_mxfp8(pc, ps)
    assert np.array_equal(_np(sf).reshape(-1), np.asarray(rs).reshape(-1)), int((_np(sf).reshape(-1) != np.asarray(rs).reshape(-1)).sum())
    assert np.array_equal(_np(y).reshape(-1), np.asarray(ry).reshape(-1)), int((_np(y).reshape(-1) != np.asarray(ry).reshape(-1)).sum())


# ------------------------------------------------------------------------------------------------
# fake kernels: a compiled caller (AOT autograd over the fake kernels of qutlass_amd/ops.py, eager execution of the graph) returns the eager bytes
# ------------------------------------------------------------------------------------------------
def test_compiled_quantize_swizzle_gemm_equals_eager(q):
    from qutlass_amd.utils import to_blocked

    def layer(x, h, wq, wsf, alpha):
        xq, xs = q.fusedQuantizeMx(x, h, method="abs_max")
        return q.matmul_mxf4_bf16_tn(xq.view(-1, xq.size(-1)), wq, to_blocked(xs), wsf, alpha)

    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(2, 160, 512, dtype=torch.bfloat16, device=DEV, generator=g) * 5
    w = torch.randn(384, 512, dtype=torch.bfloat16, device=DEV, generator=g)
    h = _hadamard(32)
    wq, ws = q.fusedQuantizeMx(w, h, method="abs_max")
    wsf = to_blocked(ws)
    alpha = torch.tensor([0.5], device=DEV)
    eager = layer(x, h, wq, wsf, alpha)
    for backend in ("eager", "aot_eager"):
        out = torch.compile(layer, backend=backend, fullgraph=True)(x, h, wq, wsf, alpha)
        assert torch.equal(out.view(torch.int16), eager.view(torch.int16)), backend
    sb = torch.compile(lambda t: to_blocked(t), backend="aot_eager", fullgraph=True)(ws)
    assert torch.equal(sb.view(torch.uint8), wsf.view(torch.uint8))


# ------------------------------------------------------------------------------------------------
# [r5] backward_qt_bf16 with whole-line input: bwd_qt_ring_kernel (product: 12 units per CU and more with M % 128 == 0; lab variants 5-8) and the lab-only panel
# kernel (variant 4) -- quartet_bwd_sm120.cu:327-430.  Forced through the LAB build on ragged / batched shapes against the oracle, and the product rule at a size that
# takes the ring kernel against the round-3 kernel byte for byte.
# ------------------------------------------------------------------------------------------------
import _benchlib as lab  # noqa: E402  (the LAB library: test infrastructure)


@pytest.mark.parametrize("variant", [4, 5, 6, 7, 8])
@pytest.mark.parametrize("B,N,M", [(1, 256, 256), (2, 288, 640), (1, 32, 128), (3, 1056, 1152), (1, 2080, 384)])
def test_backward_qt_whole_line_kernels_equal_the_oracle(q, variant, B, N, M):
    """group counts that are not multiples of 4 / 8, M % 256 in {0, 128}, fewer rows than a workgroup takes, batches; input scale bytes 0 and 255 included"""
    rng = np.random.default_rng(B * 1000 + N + M + variant)
    h = _hadamard(32)
    codes = rng.integers(0, 256, size=(B, N, M // 2), dtype=np.uint8)
    scales = rng.integers(110, 140, size=(B, N, M // 32), dtype=np.uint8)
    codes[:, -32:, : M // 4] = 0                  # all-zero groups: the reference's 0 * inf = NaN -> code 7 path
    scales[:, 0, 0] = 0
    scales[:, N // 2, -1] = 255
    with lab.forced(bwd_variant=variant):
        e2m1, e8m0 = lab.backward_qt_bf16(torch.from_numpy(codes).to(DEV), torch.from_numpy(scales).to(DEV), h, torch.tensor([3.0], device=DEV))
    rq, rs = oracle.backward_qt_bf16(codes, scales, _np(h), 3.0, acc_model=1)
    assert np.array_equal(_np(e8m0), rs), int((_np(e8m0) != rs).sum())
    eq = oracle.codes_equal_mod_zero_sign(_np(e2m1).reshape(rq.shape), rq)
    assert int((~eq).sum()) <= 1e-4 * eq.size, int((~eq).sum())


def test_backward_qt_product_rule_takes_the_ring_kernel_and_equals_the_round3_kernel(q):
    """8192 x 4224 x 2 batches = 16.5 units per CU with M % 128 == 0: the PRODUCT library launches bwd_qt_ring_kernel (test_backward_op_kernel_rules pins the rule on
    the CPU); its bytes equal the round-3 kernel's and the wave-owned kernel's, general rotation matrix, 131 + 1 groups (not a multiple of 4)."""
    g = torch.Generator(device=DEV).manual_seed(11)
    h = (torch.randn(32, 32, device=DEV, generator=g) * 0.2).to(torch.bfloat16)
    B, N, M = 2, 4096 + 128, 8192 + 128
    xq = torch.randint(0, 256, (B, N, M // 2), dtype=torch.uint8, device=DEV, generator=g)
    xs = torch.randint(116, 136, (B, N, M // 32), dtype=torch.uint8, device=DEV, generator=g)
    alpha = torch.tensor([0.61], device=DEV)
    lib = q._lib.load()
    import ctypes
    f = lib.qutlass_amd_debug_stream_plan
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    assert f(1, B, N, M) == 5
    pq, ps = q.backward_qt_bf16(xq, xs.view(torch.float8_e8m0fnu), h, alpha)
    for v in (1, 2):
        with lab.forced(bwd_variant=v):
            oq, osf = lab.backward_qt_bf16(xq, xs, h, alpha)
        assert torch.equal(pq.view(torch.uint8).reshape(oq.shape), oq) and torch.equal(ps.view(torch.uint8).reshape(osf.shape), osf), v


def test_backward_qt_scale_tensor_at_an_odd_address_takes_the_other_kernels(q):
    """the ring kernel fetches the scale bytes as dword LDS-DMA pieces: an e8m0 operand that is a view at an odd byte offset must not take it (capi.hip checks the
    alignment) and must give the same bytes as the aligned copy"""
    g = torch.Generator(device=DEV).manual_seed(13)
    h = _hadamard(32)
    B, N, M = 1, 8192, 8192
    xq = torch.randint(0, 256, (B, N, M // 2), dtype=torch.uint8, device=DEV, generator=g)
    xs = torch.randint(116, 136, (B, N, M // 32), dtype=torch.uint8, device=DEV, generator=g)
    buf = torch.empty(xs.numel() + 16, dtype=torch.uint8, device=DEV)
    odd = buf[1:1 + xs.numel()].view(xs.shape)
    odd.copy_(xs)
    assert odd.data_ptr() % 4 == 1
    alpha = torch.tensor([1.25], device=DEV)
    want = q.backward_qt_bf16(xq, xs.view(torch.float8_e8m0fnu), h, alpha)
    got = q.backward_qt_bf16(xq, odd.view(torch.float8_e8m0fnu), h, alpha)
    for a_, b_ in zip(got, want):
        assert torch.equal(a_.view(torch.uint8), b_.view(torch.uint8))
