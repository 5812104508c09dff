"""GPU: the reference's own end-to-end shape grids (tests/mxfp4_test.py:272-299 `test_llama_shapes`, tests/nvfp4_test.py:227-262,
tests/mxfp8_test.py:98-130): LLaMA 7B / 13B / 33B / 70B layer shapes x batch {1, 16} x every rotation size, quantise both
operands with the fused quantizer, swizzle the scales, multiply, and require what the reference requires --
`out.equal(out_ref)` for MXFP4 / NVFP4 against the fp64 dequantise-matmul of the SAME packed operands, assert_close
(1e-1) for MXFP8 TN and NN.  Two references per case: (i) the fp64 dequantise-matmul in plain torch on the GPU over the WHOLE
output (a test-side restatement, pinned to nothing by itself) and (ii) the pinned CPU oracle (oracle/, golden-vector checked)
on 192 sampled output columns -- so every case of the grid carries the pinned-oracle guarantee."""
import numpy as np
import pytest
import torch

import oracle  # the checker

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)

# (k, n) per linear layer, as listed in the reference's tests (data, not code)
LLAMA_MODELS = {
    "7B": [(4096, 3 * 4096), (4096, 4096), (4096, 2 * 10752), (10752, 4096)],
    "13B": [(5120, 3 * 5120), (5120, 5120), (5120, 2 * 13568), (13568, 5120)],
    "33B": [(6656, 3 * 6656), (6656, 6656), (6656, 2 * 17664), (17664, 6656)],
    "70B": [(8192, 3 * 8192), (8192, 8192), (8192, 2 * 21760), (21760, 8192)],
}
_E2M1 = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0]


@pytest.fixture(scope="module")
def q():
    import qutlass_amd

    return qutlass_amd


def _hadamard(n):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(DEV)


def _dequant_fp64(codes, scales, rows, group):
    """packed e2m1 (rows, k/2) + per-group scales (>= rows, k/group) -> (rows, k) float64; element 2j = low nibble of byte j."""
    lut = torch.tensor(_E2M1, dtype=torch.float64, device=codes.device)
    c = codes.view(torch.uint8)
    vals = torch.stack([lut[(c & 0xF).long()], lut[(c >> 4).long()]], dim=-1).reshape(rows, -1)
    s = scales[:rows]
    if s.dtype == torch.float8_e8m0fnu:
        s = torch.pow(2.0, s.view(torch.uint8).to(torch.float64) - 127.0)
    else:
        s = s.to(torch.float32).to(torch.float64)
    return (vals.reshape(rows, -1, group) * s[:, : vals.shape[1] // group, None]).reshape(rows, -1)


def _np(t):
    t = t.detach().cpu().contiguous()
    return t.view(torch.uint16).numpy() if t.dtype == torch.bfloat16 else (t.view(torch.uint8).numpy() if t.element_size() == 1 else t.numpy())


def _oracle_columns(kind, a_q, a_s, b_q, b_s, out, m, n, k, group, seed):
    """oracle.gemm_blockscaled on all rows of A x 192 sampled rows of B (= output columns), bit-exact against `out`."""
    cols = sorted({0, 1, n // 2, n - 2, n - 1} | set(np.random.default_rng(seed).integers(0, n, 187).tolist()))
    ci = torch.tensor(cols, device=out.device)
    rm = lambda s_, rows: _np(s_).reshape(-1)[: rows * (k // group)].reshape(rows, k // group)
    ref = oracle.gemm_blockscaled(kind, _np(a_q), _np(b_q[ci]), oracle.to_blocked(rm(a_s, m)), oracle.to_blocked(np.ascontiguousarray(rm(b_s, n)[cols])),
                                  1.0, m, len(cols), k)
    got = _np(out[:, ci])
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} sampled outputs differ from the pinned oracle"


_weights = {}


def _weight(key, n, k, make):
    """quantised weight of a layer, shared by the batch sizes of the same (format, rotation) case"""
    if key not in _weights:
        _weights.clear()   # one resident weight at a time (up to 43520 x 8192)
        _weights[key] = make()
    return _weights[key]


@pytest.mark.parametrize("had_size", [32, 64, 128])
@pytest.mark.parametrize("layer_idx", [0, 1, 2, 3])
@pytest.mark.parametrize("model", list(LLAMA_MODELS))
def test_mxfp4_llama_shapes_exact(q, model, layer_idx, had_size):
    from qutlass_amd.utils import to_blocked

    k, n = LLAMA_MODELS[model][layer_idx]
    h = _hadamard(had_size)
    alpha = torch.tensor([1.0], device=DEV)
    torch.manual_seed(0)
    b = torch.rand(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b_q, b_s = q.fusedQuantizeMx(b, h, method="quest")
    b_dq = _dequant_fp64(b_q, b_s, n, 32)
    b_sf = to_blocked(b_s)
    for m in (1, 16):
        a = torch.rand(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
        a_q, a_s = q.fusedQuantizeMx(a, h, method="quest")
        ref = (_dequant_fp64(a_q, a_s, m, 32) @ b_dq.T).to(torch.bfloat16)
        out = q.matmul_mxf4_bf16_tn(a_q, b_q, to_blocked(a_s), b_sf, alpha)
        assert out.equal(ref), (model, layer_idx, m, had_size, int((out != ref).sum()))
        _oracle_columns(oracle.KIND_MXFP4, a_q, a_s, b_q, b_s, out, m, n, k, 32, layer_idx * 7 + m)
        out2 = q.matmul_ada_mxf4_bf16_tn(a_q, b_q, a_s[:m].contiguous(), b_s[:n].contiguous(), alpha)
        assert out2.equal(ref), ("ada", model, layer_idx, m, had_size)


@pytest.mark.parametrize("rot_size", [16, 32, 64, 128])
@pytest.mark.parametrize("layer_idx", [0, 1, 2, 3])
@pytest.mark.parametrize("model", list(LLAMA_MODELS))
def test_nvfp4_llama_shapes_exact(q, model, layer_idx, rot_size):
    from qutlass_amd.utils import to_blocked

    k, n = LLAMA_MODELS[model][layer_idx]
    h = _hadamard(rot_size)
    alpha = torch.tensor([1.0], device=DEV)
    gs = torch.tensor([1.0], device=DEV)
    torch.manual_seed(0)
    b = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b_q, b_s = q.fusedQuantizeNv(b, h, gs)
    b_dq = _dequant_fp64(b_q, b_s, n, 16)
    b_sf = to_blocked(b_s)
    for m in (1, 16):
        a = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
        a_q, a_s = q.fusedQuantizeNv(a, h, gs)
        ref = (_dequant_fp64(a_q, a_s, m, 16) @ b_dq.T).to(torch.bfloat16)
        out = q.matmul_nvf4_bf16_tn(a_q, b_q, to_blocked(a_s), b_sf, alpha)
        assert out.equal(ref), (model, layer_idx, m, rot_size, int((out != ref).sum()))
        _oracle_columns(oracle.KIND_NVFP4, a_q, a_s, b_q, b_s, out, m, n, k, 16, layer_idx * 7 + m)


def _pseudoquant_mxfp8(x):
    """e8m0 per 32 = floor(log2 amax) - 8 + 1 ... restated from the reference's producer semantics: scale = 2^(floor(log2(amax)) - 8),
    values rounded to e4m3fn.  Only used to produce operands; the check itself dequantises the produced bytes."""
    xr = x.float().reshape(x.shape[0], -1, 32)
    amax = xr.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -120)
    e = torch.floor(torch.log2(amax)) - 8.0
    scale = torch.pow(2.0, e)
    q8 = (xr / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    e8 = (e + 127.0).clamp(0, 254).to(torch.uint8).reshape(x.shape[0], -1).view(torch.float8_e8m0fnu)
    return q8.reshape(x.shape), e8


@pytest.mark.parametrize("layer_idx", [0, 1, 2, 3])
@pytest.mark.parametrize("model", list(LLAMA_MODELS))
def test_mxfp8_llama_shapes_tn_nn_close(q, model, layer_idx):
    from qutlass_amd.utils import to_blocked

    k, n = LLAMA_MODELS[model][layer_idx]
    m = 16
    torch.manual_seed(0)
    a = torch.rand(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b = torch.rand(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    a8, a_s = _pseudoquant_mxfp8(a)
    b8, b_s = _pseudoquant_mxfp8(b)
    deq = lambda v, s: (v.to(torch.float32).to(torch.float64).reshape(v.shape[0], -1, 32)
                        * torch.pow(2.0, s.view(torch.uint8).to(torch.float64) - 127.0)[:, :, None]).reshape(v.shape[0], -1)
    ref = (deq(a8, a_s) @ deq(b8, b_s).T).to(torch.bfloat16)
    alpha = torch.tensor([1.0], device=DEV)
    pad = lambda s, rows: torch.cat([s.view(torch.uint8), torch.zeros(-(-rows // 128) * 128 - rows, s.shape[1], dtype=torch.uint8, device=DEV)]).view(torch.float8_e8m0fnu)
    a_sf, b_sf = to_blocked(pad(a_s, m)), to_blocked(pad(b_s, n))
    out = q.matmul_mxf8_bf16_tn(a8, b8, a_sf, b_sf, alpha)
    torch.testing.assert_close(out, ref, atol=1e-1, rtol=1e-1)
    a8t = a8.view(torch.uint8).T.contiguous().view(torch.float8_e4m3fn)
    out_nn = q.matmul_mxf8_bf16_nn(a8t, b8, a_sf, b_sf, alpha)
    torch.testing.assert_close(out_nn, ref, atol=1e-1, rtol=1e-1)
    # (TN may take the split-K path for these small outputs, NN never does: same values up to fp32 summation order)
    torch.testing.assert_close(out_nn, out, atol=0.0, rtol=2.0 ** -7)
    # pinned oracle on 192 sampled columns, 1 bf16 ulp + 2e-5 max|ref| (fp32 accumulation of 8-bit-significand products)
    cols = sorted({0, n - 1} | set(np.random.default_rng(layer_idx).integers(0, n, 190).tolist()))
    ci = torch.tensor(cols, device=DEV)
    oref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, _np(a8), _np(b8[ci]), oracle.to_blocked(_np(a_s)), oracle.to_blocked(_np(b_s[ci])), 1.0, m, len(cols), k)
    want = oracle.bf16_bits_to_f32(oref).astype(np.float64)
    for o in (out, out_nn):
        got = oracle.bf16_bits_to_f32(_np(o[:, ci])).astype(np.float64)
        assert (np.abs(got - want) <= np.abs(want) / 128.0 + 2e-5 * np.abs(want).max()).all()


def test_quartet_forward_gemm_exact(q):
    """tests/quartet_test.py:241-258: 12288 x 8192 x 4096, both operands Quest-quantised WITH the clip mask, exact equality."""
    from qutlass_amd.utils import to_blocked

    m, n, k = 4096 * 3, 4096 * 2, 4096
    h = _hadamard(32)
    torch.manual_seed(0)
    a = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    a_q, a_s, a_mask = q.fusedQuantizeMx(a, h, method="quest", return_mask=True)
    b_q, b_s, _ = q.fusedQuantizeMx(b, h, method="quest", return_mask=True)
    assert a_mask.shape == (m, k // 8) and a_mask.dtype == torch.uint8
    ref = (_dequant_fp64(a_q, a_s, m, 32) @ _dequant_fp64(b_q, b_s, n, 32).T).to(torch.bfloat16)
    out = q.matmul_mxf4_bf16_tn(a_q, b_q, to_blocked(a_s), to_blocked(b_s), torch.tensor([1.0], device=DEV))
    assert out.equal(ref), int((out != ref).sum())


def test_quartet_fp8_requant_chain(q):
    """tests/quartet_test.py `_fp8_requant_test` (m, n = 2694, 256; rows are padded by the wrappers as in the reference):
    backward_bf16_square_double_mxfp8 and mxfp4_transpose_mxfp8 bit-exact against the CPU oracle, then their outputs feed
    matmul_mxf8_bf16_nn, whose result matches the fp64 product of the dequantised operands within the reference's 1e-1."""
    import numpy as np

    import oracle
    from qutlass_amd.utils import to_blocked

    def npu8(t):
        return t.detach().cpu().contiguous().view(torch.uint8).numpy()

    m, n = 2694, 256
    bf16 = torch.arange(0, n, dtype=torch.bfloat16, device=DEV)[None, :].repeat(m, 1)
    a8, a_rs, a_cs = q.backward_bf16_square_double_mxfp8(bf16)
    mp = -(-m // 128) * 128
    assert a8.shape == (mp, n) and a_rs.shape == (mp, n // 32) and a_cs.shape == (n, mp // 32)
    xpad = torch.cat([bf16, torch.zeros(mp - m, n, dtype=torch.bfloat16, device=DEV)])
    oy, ors, ocs = oracle.backward_bf16_square_double_mxfp8(xpad.cpu().view(torch.uint16).numpy())
    assert np.array_equal(npu8(a8), oy) and np.array_equal(npu8(a_rs), ors) and np.array_equal(npu8(a_cs), ocs)

    fp4, scales = q.fusedQuantizeMx(bf16, torch.eye(32, dtype=torch.bfloat16, device=DEV), method="abs_max")
    sc = scales.clone()
    b8, b_e = q.mxfp4_transpose_mxfp8(fp4, sc)
    mp2 = -(-m // 256) * 256
    assert b8.shape == (n, mp2) and b_e.shape == (n, mp2 // 32) and mp2 == mp
    fp4_pad = torch.cat([fp4.view(torch.uint8), torch.zeros(mp2 - m, n // 2, dtype=torch.uint8, device=DEV)])
    sc_pad = npu8(sc)[:mp2, : n // 32].copy()
    sc_pad[m:] = 127   # the padded rows carry unit scales (the reference's wrapper writes 1.0 into the caller's tensor; here the kernel assumes it)
    oy2, oe2 = oracle.mxfp4_transpose_mxfp8(npu8(fp4_pad), sc_pad)
    assert np.array_equal(npu8(b8), oy2) and np.array_equal(npu8(b_e), oe2)
    # [r3] padding lives in the kernel: the caller's scale tensor is untouched, and a scale tensor of exactly m rows is enough
    assert torch.equal(sc.view(torch.uint8), scales.view(torch.uint8))
    sc_exact = scales.view(torch.uint8).reshape(-1)[: m * n // 32].reshape(m, n // 32).clone().view(torch.float8_e8m0fnu)
    b8x, b_ex = q.mxfp4_transpose_mxfp8(fp4, sc_exact)
    assert torch.equal(b8x.view(torch.uint8), b8.view(torch.uint8)) and torch.equal(b_ex.view(torch.uint8), b_e.view(torch.uint8))

    # the chain: A stored (K, M) = a8 (mp, n), scales (M, K/32) = column scales; B (N, K) = b8, scales (N, K/32) = b_e
    al = torch.tensor([1.0], device=DEV)
    out = q.matmul_mxf8_bf16_nn(a8, b8, to_blocked(a_cs), to_blocked(b_e), al)
    deq = lambda v, s: (v.to(torch.float32).to(torch.float64).reshape(v.shape[0], -1, 32)
                        * torch.pow(2.0, s.view(torch.uint8).to(torch.float64) - 127.0)[:, :, None]).reshape(v.shape[0], -1)
    a_dq = deq(a8.view(torch.uint8).T.contiguous().view(torch.float8_e4m3fn), a_cs)      # (M = n, K = mp)
    ref = (a_dq @ deq(b8, b_e).T).to(torch.bfloat16)
    assert out.shape == (n, n)
    torch.testing.assert_close(out, ref, atol=1e-1, rtol=1e-1)


def _rtne_e2m1(v):
    """fp64 values -> nearest e2m1 grid value, ties to the even mantissa, saturating at 6 (SURVEY.md 8a KATs)."""
    a = v.abs()
    q = torch.full_like(a, 6.0)
    for bound, val, tie_low in ((5.0, 4.0, True), (3.5, 3.0, False), (2.5, 2.0, True), (1.75, 1.5, False), (1.25, 1.0, True), (0.75, 0.5, False), (0.25, 0.0, True)):
        q = torch.where((a <= bound) if tie_low else (a < bound), torch.full_like(a, val), q)
    return torch.copysign(q, v)


@pytest.mark.parametrize("rot_size", [32, 64, 128])
@pytest.mark.parametrize("method", ["abs_max", "quest"])
def test_mx_quantizer_full_size_within_the_references_bound(q, rot_size, method):
    """tests/mxfp4_test.py:208-222 / :240-254 at the reference's size (2, 4096, 4096): dequantised GPU output vs an fp64
    restatement of the quantiser; the reference's own bar is <= 1e-4 mismatching elements (fp32 vs fp64 rotation ties)."""
    torch.manual_seed(0)
    h = _hadamard(rot_size)
    x = torch.randn(2, 4096, 4096, dtype=torch.bfloat16, device=DEV) * 25.0
    xq, xs = q.fusedQuantizeMx(x, h, method=method)
    rows = 2 * 4096
    alpha = 3.0 if method == "abs_max" else 1.0
    got = _dequant_fp64(xq.reshape(rows, -1), xs.view(torch.uint8).reshape(-1)[: rows * 128].reshape(rows, 128).view(torch.float8_e8m0fnu), rows, 32) / alpha
    bad = 0
    for r0 in range(0, rows, 2048):   # fp64 in slabs of 2048 rows
        xh = (x.reshape(rows, -1)[r0 : r0 + 2048].to(torch.float64).reshape(-1, rot_size) @ h.to(torch.float64)).reshape(-1, 32)
        if method == "abs_max":
            s = xh.abs().amax(dim=-1, keepdim=True) + 1e-8
        else:
            mean = xh.mean(dim=-1, keepdim=True)
            var = (xh * xh).mean(dim=-1, keepdim=True) - mean * mean
            s = torch.sqrt(var.clamp_min(0.0)) * (2.92247856 / 6.0) + 1e-8
        scale = torch.pow(2.0, torch.floor(torch.log2(s)))
        ref = _rtne_e2m1(xh / scale * alpha) * scale / alpha
        bad += int((got[r0 : r0 + 2048].reshape(-1, 32) != ref).sum())
    assert bad / x.numel() <= 1e-4, bad


@pytest.mark.parametrize("rot_size", [16, 32, 64, 128])
def test_nv_quantizer_full_size_and_gemm_exact(q, rot_size):
    """tests/nvfp4_test.py:190-224: (2, 4096, 4096) with global_scale 6 -> dequantised output vs an fp64 restatement within
    the reference's bound (<= 1e-1 mismatching elements), then 504 x 8192 x 4096 quantise -> swizzle -> GEMM, exact."""
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(0)
    h = _hadamard(rot_size)
    gs = torch.tensor([6.0], device=DEV)
    x = torch.randn(2, 4096, 4096, dtype=torch.bfloat16, device=DEV) * 25.0
    xq, xs = q.fusedQuantizeNv(x, h, gs)
    rows = 2 * 4096
    s2d = xs.view(torch.uint8).reshape(-1)[: rows * 256].reshape(rows, 256).view(torch.float8_e4m3fn)
    got = _dequant_fp64(xq.reshape(rows, -1), s2d, rows, 16) / 6.0
    bad = 0
    for r0 in range(0, rows, 2048):
        xh = (x.reshape(rows, -1)[r0 : r0 + 2048].to(torch.float64).reshape(-1, rot_size) @ h.to(torch.float64)).reshape(-1, 16)
        sf = (xh.abs().amax(dim=-1, keepdim=True) * (6.0 / 6.0)).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float32).to(torch.float64)
        step = sf / 6.0
        ref = torch.where(sf > 0, _rtne_e2m1(xh / step.clamp_min(1e-300)) * step, torch.zeros_like(xh))
        bad += int((got[r0 : r0 + 2048].reshape(-1, 16) != ref).sum())
    assert bad / x.numel() <= 1e-1, bad

    m, n, k = 504, 4096 * 2, 4096
    a = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    a_q, a_s = q.fusedQuantizeNv(a, h, gs)
    b_q, b_s = q.fusedQuantizeNv(b, h, gs)
    ref = (_dequant_fp64(a_q, a_s, m, 16) @ _dequant_fp64(b_q, b_s, n, 16).T).to(torch.bfloat16)
    out = q.matmul_nvf4_bf16_tn(a_q, b_q, to_blocked(a_s), to_blocked(b_s), torch.tensor([1.0], device=DEV))
    assert out.equal(ref), int((out != ref).sum())
