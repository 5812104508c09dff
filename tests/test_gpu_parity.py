"""GPU parity tests (pytest -m gpu, run on the MI355X box).

Every test goes through the product path -- qutlass_amd's Python operator surface, which calls the C ABI
of libqutlass_amd.so (hand-written HIP) -- and checks the result against
  * the committed golden fixtures produced by the reference's own Python oracles (tests/golden/), and
  * the CPU oracle (oracle/) on seeded inputs,
with the pass rules of SURVEY.md section 8c:
  bytes / indices (to_blocked, e8m0, clip mask, packed layout) bit-exact; MXFP4 / NVFP4 GEMM outputs
  bit-equal in bf16 (every product and partial sum is exact); e2m1 codes equal modulo the sign of zero with
  a mismatch fraction <= 1e-6 (reference's own bound: 1e-4) on random data and 0 on exactly-representable
  data; MXFP8 GEMM within 1 bf16 ulp + 2e-5 * max|ref| (reference tests: rtol = atol = 1e-1).
Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)
import _benchlib as lab  # noqa: E402  (the LAB build of the library: forced tiles / schedules, see tests/_benchlib.py)


@pytest.fixture(scope="module")
def q():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import qutlass_amd

    return qutlass_amd


DEV = "cuda:0"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _bf16(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.astype(np.int16)).view(torch.bfloat16).to(DEV)


def _hadamard(n):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(DEV)


def _np(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.uint16).numpy()
    if t.element_size() == 1:
        return t.view(torch.uint8).numpy()
    return t.numpy()


def _loaded_native_lib():
    # the parity claims are void if the HIP library is not the thing that ran
    with open("/proc/self/maps") as f:
        return any("libqutlass_amd.so" in line for line in f)


def test_native_library_is_loaded(q):
    assert _loaded_native_lib()
    assert b"gfx950" in q._lib.load().qutlass_amd_version()


# ------------------------------------------------------------------------------------------------
# to_blocked
# ------------------------------------------------------------------------------------------------
def test_to_blocked_golden_and_ragged(q, golden_dir):
    from qutlass_amd.utils import to_blocked

    g = _load(golden_dir, "to_blocked.npz")
    for i in range(4):
        x = torch.from_numpy(g[f"in{i}"]).to(DEV)
        assert np.array_equal(_np(to_blocked(x)), g[f"out{i}"]), i
        assert np.array_equal(_np(to_blocked(x.view(torch.float8_e8m0fnu), use_triton_kernel=True)), g[f"out{i}"])
    rng = np.random.default_rng(0)
    for rows, cols in [(1, 4), (16, 128), (130, 5), (504, 64), (4096, 128), (300, 131), (8192, 512)]:
        a = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        got = _np(to_blocked(torch.from_numpy(a).to(DEV)))
        assert np.array_equal(got, oracle.to_blocked(a)), (rows, cols)
    out = to_blocked(torch.zeros(128, 4, dtype=torch.float8_e4m3fn, device=DEV))
    assert out.dtype == torch.float8_e4m3fn and out.dim() == 1  # same dtype, flat (utils.py:193)


# ------------------------------------------------------------------------------------------------
# fusedQuantizeMx
# ------------------------------------------------------------------------------------------------
def _check_mx(q, x, h, method, mask, exact):
    out = q.fusedQuantizeMx(x, h, method=method, return_mask=mask)
    e2m1, e8m0 = out[0], out[1]
    assert e2m1.shape == (*x.shape[:-1], x.shape[-1] // 2) and e2m1.dtype == torch.uint8
    assert e8m0.dtype == torch.float8_e8m0fnu
    n = x.numel()
    rq, rs, rm = oracle.fused_quantize_mx(_np(x), _np(h), oracle.QUEST if method == "quest" else oracle.ABS_MAX, with_mask=mask)
    got_s = _np(e8m0).reshape(-1)[: n // 32]
    assert np.array_equal(got_s, rs), f"e8m0 mismatches: {(got_s != rs).sum()}"
    eq = oracle.codes_equal_mod_zero_sign(_np(e2m1), rq)
    bad = int((~eq).sum())
    assert bad == 0 if exact else bad <= max(1, int(1e-6 * n)), f"{bad} code mismatches of {n}"
    if mask:
        assert out[2].shape == (*x.shape[:-1], x.shape[-1] // 8)
        assert np.array_equal(_np(out[2]).reshape(-1), rm)


@pytest.mark.parametrize("hw", [0, 1])
def test_fused_quantize_mx_golden(q, golden_dir, hw):
    """hw = 1: the product (torch op -> libqutlass_amd.so, hardware e2m1 convert -- the one encoder it ships).  hw = 0: the software encoder, kept in the
    LAB build only ([r4]: the product library has no options any more), through the lab library's C ABI: both must reproduce the golden bytes."""
    if hw:
        quant = q.fusedQuantizeMx
    else:
        quant = lambda x, h, method, return_mask=False: lab.fused_quantize_mx(x, h, method, return_mask)
        assert lab.set_option("hw_fp4_cvt", 0) == 1
    try:
        g = _load(golden_dir, "quantize_mx.npz")
        for c in range(int(g["ncases"])):
            R, quest = g[f"meta{c}"]
            x, h = _bf16(g[f"x{c}"]), _bf16(g[f"h{c}"])
            e2m1, e8m0 = quant(x, h, method="quest" if quest else "abs_max")
            n = x.numel()
            assert np.array_equal(_np(e8m0).reshape(-1)[: n // 32], g[f"e8m0_{c}"].reshape(-1)), c
            eq = oracle.codes_equal_mod_zero_sign(_np(e2m1), g[f"e2m1_{c}"])
            assert eq.all(), (c, int((~eq).sum()))
            if quest and R == 32:
                _, _, m = quant(x, h, method="quest", return_mask=True)
                assert np.array_equal(_np(m).reshape(-1), g[f"mask{c}"].reshape(-1)), c
    finally:
        lab.set_option("hw_fp4_cvt", 1)
    assert q._lib.set_option("hw_fp4_cvt", 0) == -1   # the product library knows no key


@pytest.mark.parametrize("rot", [32, 64, 128])
@pytest.mark.parametrize("method", ["quest", "abs_max"])
def test_fused_quantize_mx_random_vs_oracle(q, rot, method):
    torch.manual_seed(0)
    x = torch.randn(2, 512, 4096, dtype=torch.bfloat16, device=DEV) * 25.0
    _check_mx(q, x, _hadamard(rot), method, mask=(rot == 32 and method == "quest"), exact=False)


@pytest.mark.parametrize("shape", [(1, 32), (3, 96), (33, 7, 64), (1, 4096), (5, 1056)])
def test_fused_quantize_mx_ragged_shapes_and_exact_inputs(q, shape):
    torch.manual_seed(1)
    # small integers x (+-0.25) Hadamard: every fp32 operation is exact -> bit-exact regardless of order
    x = torch.randint(-8, 9, shape, device=DEV).to(torch.bfloat16)
    h = (_hadamard(32).float().sign() * 0.25).to(torch.bfloat16)
    for method in ("quest", "abs_max"):
        _check_mx(q, x, h, method, mask=(method == "quest"), exact=True)
    # identity rotation (quartet_test.py:380 passes torch.eye(32)); all-zero group -> e8m0 of 1e-8
    xz = torch.zeros(4, 64, dtype=torch.bfloat16, device=DEV)
    _check_mx(q, xz, torch.eye(32, dtype=torch.bfloat16, device=DEV), "abs_max", False, True)


def test_fused_quantize_mx_leaves_scale_padding_untouched(q):
    # reference contract: only the first numel/32 bytes of the (padded_rows, padded_cols) buffer are written
    x = torch.randn(3, 96, dtype=torch.bfloat16, device=DEV)
    _, s = q.fusedQuantizeMx(x, _hadamard(32), method="abs_max")
    assert s.shape == (128, 4)


# ------------------------------------------------------------------------------------------------
# MXFP4 GEMM
# ------------------------------------------------------------------------------------------------
def _gemm_golden(q, g, c, fn, sf_dtype, kind):
    from qutlass_amd.utils import to_blocked

    m, n, k = (int(v) for v in g[f"meta{c}"])
    a, b = torch.from_numpy(g[f"a{c}"]).to(DEV), torch.from_numpy(g[f"b{c}"]).to(DEV)
    asf = to_blocked(torch.from_numpy(g[f"asf{c}"]).to(DEV).view(sf_dtype))
    bsf = to_blocked(torch.from_numpy(g[f"bsf{c}"]).to(DEV).view(sf_dtype))
    alpha = torch.tensor([float(g[f"alpha{c}"])] if f"alpha{c}" in g else [1.0], device=DEV)
    if kind == oracle.KIND_MXFP8_TN:
        a, b = a.view(torch.float8_e4m3fn), b.view(torch.float8_e4m3fn)
    out = fn(a, b, asf, bsf, alpha)
    assert out.shape == (m, n) and out.dtype == torch.bfloat16
    return _np(out), g[f"out{c}"]


@pytest.mark.parametrize("variant", [0, 1, 2, 5, 6, 7, 20, 24, 27, 28, 29, 30, 40, 60, 70, 71, 72, 73, 74, 75, 77, 90])
def test_matmul_mxf4_golden_bit_exact(q, golden_dir, variant):
    g = _load(golden_dir, "gemm_mxfp4.npz")
    impl = q if variant == 0 else lab   # 0: the product library's own dispatch; otherwise the lab build with that variant forced
    with lab.forced(gemm_variant=variant):
        for c in range(int(g["ncases"])):
            got, want = _gemm_golden(q, g, c, impl.matmul_mxf4_bf16_tn, torch.float8_e8m0fnu, oracle.KIND_MXFP4)
            assert np.array_equal(got, want), (variant, c, int((got != want).sum()))


def _pipeline(q, m, n, k, method, rot=32, seed=0):
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(seed)
    h = _hadamard(rot)
    a = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    a_q, a_s = q.fusedQuantizeMx(a, h, method=method)
    b_q, b_s = q.fusedQuantizeMx(b, h, method=method)
    alpha = torch.tensor([1.0], device=DEV)
    out = q.matmul_mxf4_bf16_tn(a_q, b_q, to_blocked(a_s), to_blocked(b_s), alpha)
    return a_q, a_s, b_q, b_s, out


@pytest.mark.parametrize("m,n,k,method", [(256, 256, 512, "abs_max"), (1, 504, 4096, "abs_max"), (504, 504, 2048, "quest"),
                                           (16, 4096, 4096, "quest"), (300, 1032, 1152, "abs_max")])
def test_pipeline_quantize_swizzle_gemm_vs_oracle(q, m, n, k, method):
    """The reference's own end-to-end test shape list (tests/mxfp4_test.py:224-237, 257-269): quantise both
    operands on the GPU, swizzle, multiply; compare with the CPU oracle's dequantise-matmul on the SAME packed
    operands -> exact bf16 equality."""
    a_q, a_s, b_q, b_s, out = _pipeline(q, m, n, k, method)
    sfa = oracle.to_blocked(_np(a_s)[:, : k // 32] if a_s.shape[1] == k // 32 else _np(a_s))
    sfb = oracle.to_blocked(_np(b_s)[:, : k // 32] if b_s.shape[1] == k // 32 else _np(b_s))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a_q), _np(b_q), sfa, sfb, 1.0, m, n, k)
    got = _np(out)
    assert np.array_equal(got, ref), int((got != ref).sum())


def test_matmul_mxf4_full_size_properties(q):
    """BASELINE.json configs[1] (4096^3) at full size: (i) a 96-row slab against the CPU oracle, bit-exact;
    (ii) linearity in alpha (exact: power of two); (iii) row-permutation equivariance; (iv) every tile
    schedule produces the identical matrix."""
    from qutlass_amd.utils import to_blocked

    m = n = k = 4096
    a_q, a_s, b_q, b_s, out = _pipeline(q, m, n, k, "abs_max")
    rows = torch.randperm(m)[:96].sort().values
    sfa_rm = _np(a_s)[rows.numpy()]
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a_q[rows.to(DEV)]), _np(b_q), oracle.to_blocked(sfa_rm),
                                  oracle.to_blocked(_np(b_s)), 1.0, len(rows), n, k)
    assert np.array_equal(_np(out[rows.to(DEV)]), ref)
    asf, bsf = to_blocked(a_s), to_blocked(b_s)
    half = q.matmul_mxf4_bf16_tn(a_q, b_q, asf, bsf, torch.tensor([0.5], device=DEV))
    assert torch.equal(half, out * 0.5)
    perm = torch.randperm(m, device=DEV)
    outp = q.matmul_mxf4_bf16_tn(a_q[perm].contiguous(), b_q, to_blocked(a_s[perm].contiguous()), bsf, torch.tensor([1.0], device=DEV))
    assert torch.equal(outp, out[perm])
    for variant in (1, 5, 6, 3, 4, 8, 9, 20, 27, 28, 29, 30, 40, 70, 73, 90):
        with lab.forced(gemm_variant=variant):
            assert torch.equal(lab.matmul_mxf4_bf16_tn(a_q, b_q, asf, bsf, torch.tensor([1.0], device=DEV)), out), variant


@pytest.mark.parametrize("m,n,k", [(64, 4096, 14336), (16, 512, 8192), (200, 264, 7168), (128, 4096, 6144), (40, 1032, 14464)])
def test_split_k_small_output_long_k(q, m, n, k):
    """Small outputs with a long K run the ring kernel with K split over grid.y and a second pass that sums the fp32
    partials (qutlass_amd_matmul_mxf4_bf16_tn_ws; the torch op takes the scratch from the caching allocator).  On
    quantised test data every partial sum is exact, so the result must equal the single-pass kernel (split-K off,
    "pp_flags" bit 7), the 2-stage simple schedule, and the CPU oracle, bit for bit."""
    from qutlass_amd.utils import to_blocked

    a_q, a_s, b_q, b_s, out = _pipeline(q, m, n, k, "abs_max", seed=m + k)
    expect_split = q._lib.load().qutlass_amd_gemm_splitk_workspace_bytes(4, m, n, k)
    t32, kt = -(-m // 32) * -(-n // 32), -(-k // 256)
    takes_ks = t32 <= 256 and (kt <= 24 or (2 * t32 > 256 and kt <= 64))   # [r6] capi.hip ks_plan: the in-workgroup K-split kernel (no scratch) takes the shape
    assert (expect_split > 0) == (k >= 32 * 256 and -(-m // 64) * -(-n // 64) <= 128 and not takes_ks)
    asf, bsf, al = to_blocked(a_s), to_blocked(b_s), torch.tensor([1.0], device=DEV)
    with lab.forced(pp_flags=1 | 128):
        single = lab.matmul_mxf4_bf16_tn(a_q, b_q, asf, bsf, al)
    with lab.forced(gemm_variant=29):
        simple = lab.matmul_mxf4_bf16_tn(a_q, b_q, asf, bsf, al)
    assert torch.equal(out, single) and torch.equal(out, simple)
    rows = sorted({0, m // 3, m - 1})
    sfa = oracle.to_blocked(np.ascontiguousarray(np.concatenate([_np(a_s)[rows], np.zeros((128 - len(rows), k // 32), np.uint8)])))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, np.ascontiguousarray(_np(a_q)[rows]), _np(b_q), sfa, oracle.to_blocked(_np(b_s)), 1.0, len(rows), n, k)
    assert np.array_equal(_np(out)[rows], ref)


@pytest.mark.parametrize("m,n,k", [(96, 5120, 25600), (128, 8192, 28672), (100, 2056, 57344), (256, 5120, 25600)])
def test_split_k_model_corrected_plans(q, m, n, k):
    """[r3] Where the tile-count rule would leave a long K to unsplit (or twice-split) 64x64 tiles, capi.hip's fitted model picks 128x128 / 64x128 ring tiles with
    4 - 8 K ranges (reference counterpart: the tile / split heuristics inside CUTLASS' kernel selection, gemm.cu:195-222).  Same contract as every split: on
    quantised data all partial sums are exact, so the result equals the single-pass launch (split-K off, "pp_flags" bit 7) and the oracle, bit for bit."""
    from qutlass_amd.utils import to_blocked

    a_q, a_s, b_q, b_s, out = _pipeline(q, m, n, k, "abs_max", seed=m + k)
    t64 = -(-m // 64) * -(-n // 64)
    ws = q._lib.load().qutlass_amd_gemm_splitk_workspace_bytes(4, m, n, k)
    assert ws % (m * n * 4) == 0 and (ws // (m * n * 4) >= 4 or (t64 > 256 and ws // (m * n * 4) == 2)) and (t64 >= 128 or ws // (m * n * 4) == 8)   # more ranges than 256 / (64x64 tiles) allows
    asf, bsf, al = to_blocked(a_s), to_blocked(b_s), torch.tensor([1.0], device=DEV)
    with lab.forced(pp_flags=1 | 128):
        single = lab.matmul_mxf4_bf16_tn(a_q, b_q, asf, bsf, al)
    assert torch.equal(out, single)
    rows = sorted({0, m // 3, m - 1})
    sfa = oracle.to_blocked(np.ascontiguousarray(np.concatenate([_np(a_s)[rows], np.zeros((128 - len(rows), k // 32), np.uint8)])))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, np.ascontiguousarray(_np(a_q)[rows]), _np(b_q), sfa, oracle.to_blocked(_np(b_s)), 1.0, len(rows), n, k)
    assert np.array_equal(_np(out)[rows], ref)


def test_matmul_mxf4_largest_sweep_shape_row_samples(q):
    """M = 65536 is the top of the reference's benchmark sweep (benchmarks/bench_mxfp4_sm100.py:176-194): tile offsets
    approach 2^31 bytes.  Random codes, scales within 3 binades (exact regime) -> sampled rows must match the oracle bit
    for bit, including the last row."""
    from qutlass_amd.utils import to_blocked

    m, n, k = 65536, 4096, 4096
    g = torch.Generator(device="cpu").manual_seed(7)
    a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, generator=g).to(DEV)
    b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, generator=g).to(DEV)
    sa = torch.randint(126, 129, (m, k // 32), dtype=torch.uint8, generator=g)
    sb = torch.randint(126, 129, (n, k // 32), dtype=torch.uint8, generator=g)
    e8 = torch.float8_e8m0fnu
    out = q.matmul_mxf4_bf16_tn(a, b, to_blocked(sa.to(DEV).view(e8)), to_blocked(sb.to(DEV).view(e8)), torch.tensor([1.0], device=DEV))
    rows = [0, 1, 255, 256, 32767, 32768, 40001, 65279, 65280, 65534, 65535]
    a_s = np.ascontiguousarray(_np(a[rows]))
    sa_s = np.concatenate([sa.numpy()[rows], np.zeros((128 - len(rows), k // 32), np.uint8)])
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, a_s, _np(b), oracle.to_blocked(sa_s), oracle.to_blocked(sb.numpy()), 1.0, len(rows), n, k)
    assert np.array_equal(_np(out[rows]), ref)


def test_matmul_empty_batch_returns_empty_output(q):
    """An empty batch (M = 0) is a valid call at the op level: (0, N) bf16, nothing launched."""
    e8 = torch.float8_e8m0fnu
    b = torch.zeros(64, 64, dtype=torch.uint8, device=DEV)
    bsf = torch.zeros(128 * 4, dtype=torch.uint8, device=DEV).view(e8)
    a = torch.zeros(0, 64, dtype=torch.uint8, device=DEV)
    asf = torch.zeros(0, dtype=torch.uint8, device=DEV).view(e8)
    al = torch.tensor([1.0], device=DEV)
    for fn in (q.matmul_mxf4_bf16_tn, q.matmul_ada_mxf4_bf16_tn):
        out = fn(a, b, asf, bsf, al)
        assert out.shape == (0, 64) and out.dtype == torch.bfloat16
    e4 = torch.float8_e4m3fn
    bsf_nv = torch.zeros(128 * 8, dtype=torch.uint8, device=DEV).view(e4)   # K = 128: 8 groups of 16 per row
    out = q.matmul_nvf4_bf16_tn(a, b, asf.view(torch.uint8).view(e4), bsf_nv, al)
    assert out.shape == (0, 64)
    # [r3, ADVICE r2] a malformed scale / alpha tensor is rejected whatever the batch size (the checks used to sit behind the
    # empty-shape return)
    with pytest.raises(RuntimeError, match="scale layout of B needs 1024"):
        q.matmul_nvf4_bf16_tn(a, b, asf.view(torch.uint8).view(e4), bsf.view(torch.uint8).view(e4), al)
    with pytest.raises(RuntimeError, match="alpha must be a float32"):
        q.matmul_mxf4_bf16_tn(a, b, asf, bsf, al.double())


def test_matmul_mxf4_errors(q):
    u8 = torch.zeros(128, 64, dtype=torch.uint8, device=DEV)
    sf = torch.zeros(128 * 4, dtype=torch.float8_e8m0fnu, device=DEV)
    al = torch.ones(1, device=DEV)
    with pytest.raises(RuntimeError, match="A_sf must be float8_e8m0fnu"):
        q.matmul_mxf4_bf16_tn(u8, u8, sf.view(torch.uint8), sf, al)
    with pytest.raises(RuntimeError, match="Inner dimensions must match"):
        q.matmul_mxf4_bf16_tn(u8, u8[:, :32].contiguous(), sf, sf, al)
    with pytest.raises(RuntimeError, match="K-dim must be >= 32"):
        q.matmul_mxf4_bf16_tn(u8[:, :8].contiguous(), u8[:, :8].contiguous(), sf, sf, al)
    with pytest.raises(RuntimeError, match="multiple of 128"):   # CUTLASS alignment in the reference (gemm.cu:187)
        q.matmul_mxf4_bf16_tn(u8[:, :48].contiguous(), u8[:, :48].contiguous(), sf, sf, al)


def test_op_layer_validation_messages_follow_the_reference(q):
    """bindings.cpp:38-57 / bindings_utils.h:67-136: contiguity, device and dtype checks with the reference's wording,
    raised by the C++ extension (the message carries its source location) before anything is launched."""
    u8 = torch.zeros(4, 64, dtype=torch.uint8, device=DEV)
    sf = torch.zeros(128, 4, dtype=torch.float8_e8m0fnu, device=DEV)
    al = torch.ones(1, device=DEV)
    C = torch.ops._qutlass_C
    with pytest.raises(RuntimeError, match="Expected tensor to have cuda DeviceType, but got tensor with cpu DeviceType"):
        C.matmul_mxf4_bf16_tn(u8, u8.cpu(), sf, sf, al)
    nc = torch.zeros(64, 8, dtype=torch.uint8, device=DEV).t()
    with pytest.raises(RuntimeError, match=r"Expected contiguous tensor, but got non-contiguous tensor for argument #0 'A' \(while checking arguments for matmul_mxf4_bf16_tn\)"):
        C.matmul_mxf4_bf16_tn(nc, u8, sf, sf, al)
    with pytest.raises(RuntimeError, match=r"torch_ext\.cpp"):
        C.matmul_mxf4_bf16_tn(u8, u8, sf.view(torch.uint8), sf, al)
    with pytest.raises(RuntimeError, match="A_sf has 4 elements"):      # scale tensor smaller than the blocked layout the kernel reads
        C.matmul_mxf4_bf16_tn(u8, u8, sf.reshape(-1)[:4].contiguous(), sf, al)
    with pytest.raises(RuntimeError, match="alpha must be a float32 tensor"):
        C.matmul_mxf4_bf16_tn(u8, u8, sf, sf, al.to(torch.float64))
    with pytest.raises(RuntimeError, match="to_blocked expects a 2-D matrix"):
        torch.ops.qutlass_amd.to_blocked(torch.zeros(8, dtype=torch.uint8, device=DEV))
    with pytest.raises(NotImplementedError):                               # CUDA dispatch key only, as the reference registers
        C.matmul_mxf4_bf16_tn(u8.cpu(), u8.cpu(), sf.cpu(), sf.cpu(), al.cpu())


# ------------------------------------------------------------------------------------------------
# NVFP4
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nv_variant", [0, 1, 2, 3, 4, 5, 6, 7])   # 5 / 6 / 7 = 128x128 / 128x64 / 64x64 tiles, 1 / 4 = per-wave dequant (8 / 4 waves), 2 = dequantise once into f16 LDS tiles, 3 = small-batch split-K
def test_matmul_nvf4_golden_bit_exact(q, golden_dir, nv_variant):
    g = _load(golden_dir, "gemm_nvfp4.npz")
    impl = q if nv_variant == 0 else lab
    with lab.forced(nvf4_variant=nv_variant):
        for c in range(int(g["ncases"])):
            got, want = _gemm_golden(q, g, c, impl.matmul_nvf4_bf16_tn, torch.float8_e4m3fn, oracle.KIND_NVFP4)
            assert np.array_equal(got, want), (nv_variant, c, int((got != want).sum()))


@pytest.mark.parametrize("rot", [16, 32, 64, 128])
def test_fused_quantize_nv_and_gemm_vs_oracle(q, rot):
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(2)
    m, n, k = 504, 1024, 2048
    h = _hadamard(rot)
    gs = torch.tensor([6.0], device=DEV)
    a = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    a_q, a_s = q.fusedQuantizeNv(a, h, gs)
    b_q, b_s = q.fusedQuantizeNv(b, h, gs)
    assert a_s.dtype == torch.float8_e4m3fn and a_s.shape == (512, k // 16)
    rq, rs = oracle.fused_quantize_nv(_np(a), _np(h), 6.0, oracle.ABS_MAX, acc_model=1)
    got_s = _np(a_s).reshape(-1)[: rs.size]
    sbad = int((got_s != rs).sum())
    assert sbad <= 2e-4 * rs.size, sbad   # rcp / MFMA-order differences only (reference bound: 1e-1)
    same = (got_s == rs).repeat(16)
    eq = oracle.codes_equal_mod_zero_sign(_np(a_q), rq)
    assert int((~eq & same).sum()) <= 2e-4 * eq.size
    out = q.matmul_nvf4_bf16_tn(a_q, b_q, to_blocked(a_s), to_blocked(b_s), torch.tensor([1.0], device=DEV))
    ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, _np(a_q), _np(b_q), oracle.to_blocked(_np(a_s)[:, : k // 16]),
                                  oracle.to_blocked(_np(b_s)[:, : k // 16]), 1.0, m, n, k)
    assert np.array_equal(_np(out), ref)   # exact: the reference asserts out.equal(out_ref) (nvfp4_test.py:224)


@pytest.mark.parametrize("m,n,k", [(256, 4096, 1024), (512, 4096, 1024), (128, 14336, 512), (96, 4096, 1024), (1000, 2056, 544), (2048, 4096, 1024), (1536, 4096, 512)])
def test_matmul_nvf4_occupancy_tile_choice_is_bit_identical(q, m, n, k):
    """The auto rule picks 64x64 / 128x64 / split-K / 128x128 tiles -- [r3] and, on the last two (half-chip) shapes, the 256x128 tile on
    four waves -- on these shapes; every configuration accumulates K in the same order, so the result must equal the forced 128x128
    launch bit for bit, and the oracle on sampled rows."""
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, generator=g).to(DEV)
    b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, generator=g).to(DEV)
    # e4m3 scales in [1, 4), all mantissas: every partial sum stays exact in fp32, so the oracle comparison is bit-exact
    sa = torch.randint(0x38, 0x48, (-(-m // 128) * 128, k // 16), dtype=torch.uint8, generator=g)
    sb = torch.randint(0x38, 0x48, (-(-n // 128) * 128, k // 16), dtype=torch.uint8, generator=g)
    from qutlass_amd.utils import to_blocked

    sa_b = to_blocked(sa.to(DEV).view(torch.float8_e4m3fn))
    sb_b = to_blocked(sb.to(DEV).view(torch.float8_e4m3fn))
    al = torch.tensor([0.25], device=DEV)
    outs = {0: q.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al).view(torch.int16).cpu()}
    for v in (5, 6, 7, 40):   # 40: the 256x128 tile forced (lab)
        with lab.forced(nvf4_variant=v):
            outs[v] = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al).view(torch.int16).cpu()
    for v in (0, 6, 7, 40):
        assert torch.equal(outs[v], outs[5]), v
    rows = sorted({0, m // 2, m - 1})
    a_s = np.ascontiguousarray(_np(a)[rows])
    sa_s = np.ascontiguousarray(sa.numpy()[rows])
    pad = np.zeros((128 - len(rows), k // 16), np.uint8)
    ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, a_s, _np(b), oracle.to_blocked(np.concatenate([sa_s, pad])),
                                  oracle.to_blocked(sb.numpy()), 0.25, len(rows), n, k)
    assert np.array_equal(outs[0].numpy()[rows].view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("m,n,k", [(256, 4096, 14336), (128, 4096, 14336), (200, 4104, 14368), (64, 8192, 28672), (768, 4096, 14336), (40, 1032, 6144), (100, 2056, 6144), (256, 2048, 2048), (1024, 5120, 25600),
                                   (128, 2048, 28672)])
def test_matmul_nvf4_split_k_equals_single_pass_and_oracle(q, m, n, k):
    """[r3] Outputs of a few dozen tiles with a long K: matmul_nvf4_bf16_tn splits K into ranges of an even number of 256-element stages over scratch from the
    caching allocator (qutlass_amd_matmul_nvf4_bf16_tn_ws; reference: the CUTLASS workspace of gemm.cu:290-300) and a second kernel sums the fp32 partials
    in fixed order.  e4m3 scales in [1, 4) keep every partial sum exact in fp32, so the split result must equal the forced single pass on 128x128 tiles bit
    for bit -- ragged M / N, a K tail (14368 = 56 stages + 32 elements), ranges that do not divide the stages (7 of 8, 6 of 8) -- and the oracle on sampled rows.
    256 x 2048 x 2048 (8 stages), 40 x 1032 x 6144 and [r6] 128 x 4096 x 14336 / 100 x 2056 x 6144 (the wave-owned small-batch kernels' in-workgroup split wins: 64x32 tiles in one
    round) do not split; the plan is checked against the workspace query."""
    from qutlass_amd.utils import to_blocked

    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, generator=g).to(DEV)
    b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, generator=g).to(DEV)
    sa = torch.randint(0x38, 0x48, (-(-m // 128) * 128, k // 16), dtype=torch.uint8, generator=g)
    sb = torch.randint(0x38, 0x48, (-(-n // 128) * 128, k // 16), dtype=torch.uint8, generator=g)
    sa_b = to_blocked(sa.to(DEV).view(torch.float8_e4m3fn))
    sb_b = to_blocked(sb.to(DEV).view(torch.float8_e4m3fn))
    al = torch.tensor([0.5], device=DEV)
    ws = q._lib.load().qutlass_amd_nvf4_splitk_workspace_bytes(m, n, k)
    assert (ws > 0) == ((m, n, k) not in ((256, 2048, 2048), (40, 1032, 6144), (128, 4096, 14336), (100, 2056, 6144))) and ws % (m * n * 4) == 0 and 0 <= ws // (m * n * 4) <= 8
    out = q.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)                       # torch op: workspace from the allocator -> the split path
    with lab.forced(nvf4_variant=5):
        single = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)              # 128x128 tiles, one pass
    assert torch.equal(out.view(torch.int16), single.view(torch.int16))
    # the plain C entry (no workspace) runs one pass and agrees as well
    plain = torch.empty_like(out)
    lib = q._lib.load()
    assert lib.qutlass_amd_matmul_nvf4_bf16_tn(a.data_ptr(), b.data_ptr(), sa_b.data_ptr(), sb_b.data_ptr(), al.data_ptr(), plain.data_ptr(), m, n, k,
                                               torch.cuda.current_stream().cuda_stream) == 0
    assert torch.equal(out.view(torch.int16), plain.view(torch.int16))
    rows = sorted({0, m // 2, m - 1})
    pad = np.zeros((128 - len(rows), k // 16), np.uint8)
    ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, np.ascontiguousarray(_np(a)[rows]), _np(b), oracle.to_blocked(np.concatenate([np.ascontiguousarray(sa.numpy()[rows]), pad])),
                                  oracle.to_blocked(sb.numpy()), 0.5, len(rows), n, k)
    assert np.array_equal(_np(out)[rows].view(np.uint16), ref.view(np.uint16))


# ------------------------------------------------------------------------------------------------
# MXFP8
# ------------------------------------------------------------------------------------------------
def _mxfp8_close(got_bits, want_bits):
    got = oracle.bf16_bits_to_f32(got_bits).astype(np.float64)
    want = oracle.bf16_bits_to_f32(want_bits).astype(np.float64)
    tol = np.abs(want) / 128.0 + 2e-5 * np.abs(want).max()
    return np.abs(got - want) <= tol


@pytest.mark.parametrize("m,n,k", [(192, 4096, 14336), (384, 4096, 14336), (96, 5120, 12800), (64, 1024, 8192)])   # ([r6] the last was 64 x 4096 x 4096: the wave-owned kernel's now)
def test_matmul_mxf8_split_k_plans_vs_single_pass_and_oracle(q, m, n, k):
    """[r3] MXFP8 outputs whose plan splits K -- the round-2 rule (64x64 ring tiles, last shape) and the round-3 corrections (128x128 ring tiles in 2 - 4 ranges): the
    split result against the forced single pass (an fp32 sum in another order: the oracle's tolerance between them, equal on almost every element) and against the
    oracle on sampled rows (tests/mxfp8_test.py tolerance class)."""
    from qutlass_amd.utils import to_blocked

    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) * 4).to(torch.float8_e4m3fn).to(DEV)
    y = (torch.randn(n, k, generator=g) * 4).to(torch.float8_e4m3fn).to(DEV)
    sa = torch.randint(120, 131, (m, k // 32), dtype=torch.uint8, generator=g).to(DEV)
    sb = torch.randint(120, 131, (n, k // 32), dtype=torch.uint8, generator=g).to(DEV)
    e8, al = torch.float8_e8m0fnu, torch.tensor([1.0], device=DEV)
    ws = q._lib.load().qutlass_amd_gemm_splitk_workspace_bytes(8, m, n, k)
    assert ws >= 2 * m * n * 4
    out = q.matmul_mxf8_bf16_tn(x, y, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), al)
    with lab.forced(pp_flags=1 | 128):
        single = lab.matmul_mxf8_bf16_tn(x, y, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), al)
    assert _mxfp8_close(_np(out), _np(single)).all()
    assert float((out.view(torch.int16) == single.view(torch.int16)).float().mean()) > 0.98
    rows = sorted({0, m // 2, m - 1})
    sfa = oracle.to_blocked(np.ascontiguousarray(np.concatenate([_np(sa)[rows], np.zeros((128 - len(rows), k // 32), np.uint8)])))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, np.ascontiguousarray(_np(x)[rows]), _np(y), sfa, oracle.to_blocked(_np(sb)), 1.0, len(rows), n, k)
    assert _mxfp8_close(_np(out)[rows], ref).all()


@pytest.mark.parametrize("variant", [0, 20, 30, 70, 73, 90])   # auto, 8-wave simple, 4-wave deep (per tile), ring 64x64 / 128x128, persistent deep
def test_matmul_mxf8_large_tiles_vs_oracle(q, variant):
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(6)
    m, n, k = 520, 776, 1056
    a = torch.randn(m, k, dtype=torch.bfloat16) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16) * 25.0
    aq, asf = oracle.pseudoquant_mxfp8(_np(a))
    bq, bsf = oracle.pseudoquant_mxfp8(_np(b))
    e4, e8 = torch.float8_e4m3fn, torch.float8_e8m0fnu
    impl = q if variant == 0 else lab
    with lab.forced(gemm_variant=variant):
        out = impl.matmul_mxf8_bf16_tn(torch.from_numpy(aq).to(DEV).view(e4), torch.from_numpy(bq).to(DEV).view(e4),
                                       to_blocked(torch.from_numpy(asf).to(DEV).view(e8)), to_blocked(torch.from_numpy(bsf).to(DEV).view(e8)),
                                       torch.tensor([1.0], device=DEV))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, aq, bq, oracle.to_blocked(asf), oracle.to_blocked(bsf), 1.0, m, n, k)
    assert _mxfp8_close(_np(out), ref).all()


def test_matmul_mxf8_tn_golden_and_random(q, golden_dir):
    from qutlass_amd.utils import to_blocked

    g = _load(golden_dir, "gemm_mxfp8.npz")
    for c in range(int(g["ncases"])):
        got, want = _gemm_golden(q, g, c, q.matmul_mxf8_bf16_tn, torch.float8_e8m0fnu, oracle.KIND_MXFP8_TN)
        assert _mxfp8_close(got, want).all(), c
    torch.manual_seed(3)
    m, n, k = 16, 4096, 4096   # reference test shape family (batch 16 x Llama layers, mxfp8_test.py:126-130)
    a = torch.rand(m, k, dtype=torch.bfloat16) * 25.0
    b = torch.rand(n, k, dtype=torch.bfloat16) * 25.0
    aq, asf = oracle.pseudoquant_mxfp8(_np(a))
    bq, bsf = oracle.pseudoquant_mxfp8(_np(b))
    out = q.matmul_mxf8_bf16_tn(torch.from_numpy(aq).to(DEV).view(torch.float8_e4m3fn), torch.from_numpy(bq).to(DEV).view(torch.float8_e4m3fn),
                                to_blocked(torch.from_numpy(asf).to(DEV).view(torch.float8_e8m0fnu)),
                                to_blocked(torch.from_numpy(bsf).to(DEV).view(torch.float8_e8m0fnu)), torch.tensor([1.0], device=DEV))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, aq, bq, oracle.to_blocked(asf), oracle.to_blocked(bsf), 1.0, m, n, k)
    assert _mxfp8_close(_np(out), ref).all()


def test_matmul_mxf8_e5m2_operand_golden(q, golden_dir):
    """Extension: A may be float8_e5m2 (gradient operand, BASELINE.json configs[4]); fixtures from torch's own e5m2 cast
    (tests/golden/make_golden_e5m2.py).  TN and NN through the Python surface, within the MXFP8 tolerance."""
    from qutlass_amd.utils import to_blocked

    g = _load(golden_dir, "gemm_mxfp8_e5m2.npz")
    e5, e4, e8 = torch.float8_e5m2, torch.float8_e4m3fn, torch.float8_e8m0fnu
    for c in range(int(g["ncases"])):
        m, n, k = (int(v) for v in g[f"meta{c}"])
        a = torch.from_numpy(g[f"a{c}"]).to(DEV).view(e5)
        b = torch.from_numpy(g[f"b{c}"]).to(DEV).view(e4)
        asf = to_blocked(torch.from_numpy(g[f"asf{c}"]).to(DEV).view(e8))
        bsf = to_blocked(torch.from_numpy(g[f"bsf{c}"]).to(DEV).view(e8))
        alpha = torch.tensor([1.0], device=DEV)
        out = q.matmul_mxf8_bf16_tn(a, b, asf, bsf, alpha)
        assert out.shape == (m, n) and _mxfp8_close(_np(out), g[f"out{c}"]).all(), c
        if m % 16 == 0:
            a_km = a.view(torch.uint8).T.contiguous().view(e5)
            out_nn = q.matmul_mxf8_bf16_nn(a_km, b, asf, bsf, alpha)
            assert torch.equal(out_nn.view(torch.int16), out.view(torch.int16)), c
    with pytest.raises(RuntimeError, match="B must be float8_e4m3fn"):      # only A may be e5m2
        q.matmul_mxf8_bf16_tn(a, b.view(e5), asf, bsf, alpha)


@pytest.mark.parametrize("m,n,k", [(16, 4096, 1024), (200, 264, 7168), (520, 776, 1056), (1024, 4096, 512), (2048, 4096, 256), (4096, 4096, 128)])
def test_matmul_mxf8_e5m2_operand_every_tile_vs_oracle(q, m, n, k):
    """The shapes walk the auto dispatch through ring 64x64 (+ split-K), ring 128x128, simple 128x128 and the 256x256 deep
    kernel, each in its e5m2-A instantiation, against oracle.gemm_blockscaled(KIND_MXFP8_TN_A5)."""
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, dtype=torch.bfloat16) * 25.0 * torch.exp2(torch.randint(-6, 7, (m, 1)).float()).to(torch.bfloat16)
    b = torch.randn(n, k, dtype=torch.bfloat16) * 25.0
    aq, asf = oracle.pseudoquant_mxfp8(_np(a), e5m2=True)
    bq, bsf = oracle.pseudoquant_mxfp8(_np(b))
    e8 = torch.float8_e8m0fnu
    out = q.matmul_mxf8_bf16_tn(torch.from_numpy(aq).to(DEV).view(torch.float8_e5m2), torch.from_numpy(bq).to(DEV).view(torch.float8_e4m3fn),
                                to_blocked(torch.from_numpy(asf).to(DEV).view(e8)), to_blocked(torch.from_numpy(bsf).to(DEV).view(e8)),
                                torch.tensor([0.5], device=DEV))
    rows = sorted({0, 1, m // 2, m - 1} | set(np.random.default_rng(m).integers(0, m, 28).tolist()))
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN_A5, np.ascontiguousarray(aq[rows]), bq, oracle.to_blocked(np.ascontiguousarray(asf[rows])),
                                  oracle.to_blocked(bsf), 0.5, len(rows), n, k)
    assert _mxfp8_close(_np(out)[rows], ref).all()


@pytest.mark.parametrize("m,n,k", [(16, 4096, 4096), (272, 520, 1056), (4096, 4096, 4096)])
def test_matmul_mxf8_nn_equals_tn_and_oracle(q, golden_dir, m, n, k):
    """matmul_mxf8_bf16_nn takes A stored (K, M) (mxfp8_test.py:77-96: a_e4m3.T.contiguous().view((k, m))):
    the result must be bit-identical to the TN op on the same operands (same kernel after the re-layout)
    and agree with the oracle's NN path within the fp8 tolerance."""
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(5)
    a = torch.randn(m, k, dtype=torch.bfloat16) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16) * 25.0
    aq, asf = oracle.pseudoquant_mxfp8(_np(a))
    bq, bsf = oracle.pseudoquant_mxfp8(_np(b))
    a_t = torch.from_numpy(aq).to(DEV).view(torch.float8_e4m3fn)
    b_t = torch.from_numpy(bq).to(DEV).view(torch.float8_e4m3fn)
    sa = to_blocked(torch.from_numpy(asf).to(DEV).view(torch.float8_e8m0fnu))
    sb = to_blocked(torch.from_numpy(bsf).to(DEV).view(torch.float8_e8m0fnu))
    alpha = torch.tensor([1.0], device=DEV)
    a_km = a_t.view(torch.uint8).T.contiguous().view(torch.float8_e4m3fn)
    assert a_km.shape == (k, m)
    out_nn = q.matmul_mxf8_bf16_nn(a_km, b_t, sa, sb, alpha)
    out_tn = q.matmul_mxf8_bf16_tn(a_t, b_t, sa, sb, alpha)
    assert out_nn.shape == (m, n) and out_nn.dtype == torch.bfloat16
    assert torch.equal(out_nn.view(torch.int16), out_tn.view(torch.int16))
    # every operand path explicitly: 63 = persistent kernel on the (K, M) operand (ds_read_b64_tr_b8 fragment reads; the
    # product's choice for large problems), 61 = per-tile fused kernel of round 1 (v_perm byte transposes; lab only),
    # 62 = byte-transpose pre-pass + TN
    for path in (63, 61, 62):
        with lab.forced(gemm_variant=path):
            o = lab.matmul_mxf8_bf16_nn(a_km, b_t, sa, sb, alpha)
        assert torch.equal(o.view(torch.int16), out_tn.view(torch.int16)), path
    if m * n * k <= 1 << 28:
        ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_NN, _np(a_km), bq, oracle.to_blocked(asf), oracle.to_blocked(bsf), 1.0, m, n, k)
        assert _mxfp8_close(_np(out_nn), ref).all()
    with pytest.raises(RuntimeError, match="Inner dimensions must match for A.T @ B.T"):
        q.matmul_mxf8_bf16_nn(a_km, b_t[:, : k - 32].contiguous(), sa, sb, alpha)


def test_matmul_mxf8_nn_k_tail_does_not_read_past_the_operand(q):
    """(K, M) operand on the persistent kernel with K % 128 != 0: the last K stage holds k-rows that do not exist.  The operand is
    a view into a larger buffer whose tail is filled with fp8 NaN bytes -- a kernel that fetched those rows (instead of zeros)
    would turn whole output columns into NaN.  Bit-equal to the TN op on the same operands."""
    from qutlass_amd.utils import to_blocked

    m, n, k = 4096, 4096, 1056                      # 256 tiles of 256x256 -> the in-place path; 1056 = 8 * 128 + 32
    g = torch.Generator(device=DEV).manual_seed(21)
    a = torch.randint(0, 0x78, (m, k), dtype=torch.uint8, device=DEV, generator=g)      # finite e4m3 codes
    b = torch.randint(0, 0x78, (n, k), dtype=torch.uint8, device=DEV, generator=g)
    a_s = torch.randint(124, 131, (m, k // 32), dtype=torch.uint8, device=DEV, generator=g)
    b_s = torch.randint(124, 131, (n, k // 32), dtype=torch.uint8, device=DEV, generator=g)
    e8, f8 = torch.float8_e8m0fnu, torch.float8_e4m3fn
    sa, sb = to_blocked(a_s.view(e8)), to_blocked(b_s.view(e8))
    alpha = torch.tensor([1.0], device=DEV)
    buf = torch.full((k * m + 256 * m,), 0x7F, dtype=torch.uint8, device=DEV)           # 0x7F = NaN in e4m3fn
    buf[: k * m] = a.T.contiguous().reshape(-1)
    a_km = buf[: k * m].view(k, m).view(f8)
    out_nn = q.matmul_mxf8_bf16_nn(a_km, b.view(f8), sa, sb, alpha)
    out_tn = q.matmul_mxf8_bf16_tn(a.view(f8), b.view(f8), sa, sb, alpha)
    assert not torch.isnan(out_nn.float()).any()
    assert torch.equal(out_nn.view(torch.int16), out_tn.view(torch.int16))


# ------------------------------------------------------------------------------------------------
# QAT-backward data-prep ops (reference tests/quartet_test.py:239-260, 368-384)
# ------------------------------------------------------------------------------------------------
def _codes_close(got, want, frac=1e-4):   # observed: 0 mismatching codes against the fp64 goldens and against the fp32 oracle
    eq = oracle.codes_equal_mod_zero_sign(got.reshape(want.shape), want)
    return int((~eq).sum()) <= frac * eq.size, int((~eq).sum())


def test_backward_t_bf16_golden_and_oracle(q, golden_dir):
    g = _load(golden_dir, "quartet_bwd.npz")
    h = torch.from_numpy(g["h"]).view(torch.bfloat16).to(DEV)
    for c in range(int(g["t_ncases"])):
        x = torch.from_numpy(g[f"t_x{c}"]).view(torch.bfloat16).to(DEV)
        e2m1, e8m0 = q.backward_t_bf16(x, h)
        assert e2m1.dtype == torch.float4_e2m1fn_x2 and e8m0.dtype == torch.float8_e8m0fnu
        assert e2m1.shape == (*x.shape[:-2], x.size(-1), x.size(-2) // 2)
        assert np.array_equal(_np(e8m0), g[f"t_e8m0_{c}"]), c                    # reference: xh_e8m0.equal(ref)
        ok, bad = _codes_close(_np(e2m1), g[f"t_e2m1_{c}"])
        assert ok, (c, bad)
    torch.manual_seed(11)
    x = torch.randn(2, 512, 1000, dtype=torch.bfloat16, device=DEV) * 25.0     # ragged M (multiple of 8, not of 64)
    e2m1, e8m0 = q.backward_t_bf16(x, h)
    rq, rs = oracle.backward_t_bf16(_np(x), _np(h), acc_model=1)
    assert np.array_equal(_np(e8m0), rs)
    ok, bad = _codes_close(_np(e2m1), rq, 1e-4)
    assert ok, bad
    # equivalence with the forward quantiser on the explicit transpose, up to the epsilon the forward op adds (none matters here)
    fq, fs = q.fusedQuantizeMx(x.transpose(-2, -1).contiguous(), h, method="abs_max")
    assert np.array_equal(_np(fs).reshape(-1)[: rs.size], rs.reshape(-1))
    assert np.array_equal(_np(fq), _np(e2m1).reshape(_np(fq).shape))


def test_backward_qt_bf16_golden_and_oracle(q, golden_dir):
    g = _load(golden_dir, "quartet_bwd.npz")
    h = torch.from_numpy(g["h"]).view(torch.bfloat16).to(DEV)
    alpha = torch.tensor([3.0], device=DEV)
    for c in range(int(g["qt_ncases"])):
        xq = torch.from_numpy(g[f"qt_xq{c}"]).to(DEV)
        xs = torch.from_numpy(g[f"qt_xs{c}"]).to(DEV).view(torch.float8_e8m0fnu)
        e2m1, e8m0 = q.backward_qt_bf16(xq, xs, h, alpha)
        assert np.array_equal(_np(e8m0), g[f"qt_e8m0_{c}"]), c
        ok, bad = _codes_close(_np(e2m1), g[f"qt_e2m1_{c}"])
        assert ok, (c, bad)
    torch.manual_seed(12)
    x = torch.randn(1, 256, 1024, dtype=torch.bfloat16, device=DEV) * 25.0
    xq, xs = q.fusedQuantizeMx(x, h, method="abs_max")
    xs = xs.view(torch.uint8).reshape(-1)[: x.numel() // 32].reshape(1, 256, 32).view(torch.float8_e8m0fnu)
    e2m1, e8m0 = q.backward_qt_bf16(xq, xs, h, alpha)
    rq, rs = oracle.backward_qt_bf16(_np(xq), _np(xs), _np(h), 3.0, acc_model=1)
    assert np.array_equal(_np(e8m0), rs)
    ok, bad = _codes_close(_np(e2m1), rq, 1e-4)
    assert ok, bad


def test_backward_bf16_square_double_mxfp8_bit_exact(q, golden_dir):
    g = _load(golden_dir, "quartet_bwd.npz")
    for c in range(int(g["sq_ncases"])):
        x = torch.from_numpy(g[f"sq_x{c}"]).view(torch.bfloat16).to(DEV)
        y, rs, cs = q.backward_bf16_square_double_mxfp8(x)          # pads rows to 128 like the reference
        assert y.dtype == torch.float8_e4m3fn and rs.dtype == torch.float8_e8m0fnu
        assert np.array_equal(_np(y), g[f"sq_y{c}"]), (c, int((_np(y) != g[f"sq_y{c}"]).sum()))
        assert np.array_equal(_np(rs), g[f"sq_rs{c}"]) and np.array_equal(_np(cs), g[f"sq_cs{c}"]), c
    torch.manual_seed(13)
    x = torch.randn(1024, 2048, dtype=torch.bfloat16, device=DEV) * torch.logspace(-3, 3, 2048, device=DEV).to(torch.bfloat16)
    x[128:160, 256:288] = 0
    y, rs, cs = q.backward_bf16_square_double_mxfp8(x)
    ry, rrs, rcs = oracle.backward_bf16_square_double_mxfp8(_np(x))
    assert np.array_equal(_np(rs), rrs) and np.array_equal(_np(cs), rcs)
    assert np.array_equal(_np(y), ry), int((_np(y) != ry).sum())


def test_mxfp4_transpose_mxfp8_bit_exact(q, golden_dir):
    g = _load(golden_dir, "quartet_bwd.npz")
    for c in range(int(g["tr_ncases"])):
        xq = torch.from_numpy(g[f"tr_xq{c}"]).to(DEV)
        xs = torch.from_numpy(g[f"tr_xs{c}"]).to(DEV).view(torch.float8_e8m0fnu)
        if xq.shape[1] * 2 % 256:      # the kernel needs n % 256 == 0 (the reference launches n/256 blocks)
            continue
        y, e = q.mxfp4_transpose_mxfp8(xq, xs)
        assert np.array_equal(_np(e), g[f"tr_e{c}"]) and np.array_equal(_np(y), g[f"tr_y{c}"]), c
    torch.manual_seed(14)
    x = torch.randn(1024, 768, dtype=torch.bfloat16, device=DEV) * 25.0
    xq, xs = q.fusedQuantizeMx(x, torch.eye(32, dtype=torch.bfloat16, device=DEV), method="abs_max")
    xs = xs.view(torch.uint8).reshape(-1)[: x.numel() // 32].reshape(1024, 24).contiguous().view(torch.float8_e8m0fnu)
    y, e = q.mxfp4_transpose_mxfp8(xq, xs)
    ry, re = oracle.mxfp4_transpose_mxfp8(_np(xq), _np(xs))
    assert y.shape == (768, 1024) and e.shape == (768, 32)
    assert np.array_equal(_np(e), re)
    assert np.array_equal(_np(y), ry), int((_np(y) != ry).sum())


# ------------------------------------------------------------------------------------------------
# small-batch path: matmul_ada_mxf4_bf16_tn (un-swizzled scales) and the skinny kernel behind matmul_mxf4_bf16_tn
# ------------------------------------------------------------------------------------------------
def test_matmul_ada_mxf4_golden_bit_exact(q, golden_dir):
    g = _load(golden_dir, "gemm_mxfp4.npz")
    for c in range(int(g["ncases"])):
        m, n, k = (int(v) for v in g[f"meta{c}"])
        a, b = torch.from_numpy(g[f"a{c}"]).to(DEV), torch.from_numpy(g[f"b{c}"]).to(DEV)
        asf = torch.from_numpy(g[f"asf{c}"]).to(DEV).view(torch.float8_e8m0fnu)      # row-major (m, k/32): NOT to_blocked
        bsf = torch.from_numpy(g[f"bsf{c}"]).to(DEV).view(torch.float8_e8m0fnu)
        alpha = torch.tensor([float(g[f"alpha{c}"])], device=DEV)
        out = q.matmul_ada_mxf4_bf16_tn(a, b, asf, bsf, alpha)
        assert out.shape == (m, n) and out.dtype == torch.bfloat16
        assert np.array_equal(_np(out), g[f"out{c}"]), (c, int((_np(out) != g[f"out{c}"]).sum()))


@pytest.mark.parametrize("m,n,k", [(1, 4096, 4096), (16, 4096, 4096), (32, 14336, 4096), (7, 504, 1152), (40, 1032, 2048)])
def test_small_batch_paths_agree_with_tiled_kernel_and_oracle(q, m, n, k):
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(21)
    h = _hadamard(32)
    a = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    a_q, a_s = q.fusedQuantizeMx(a, h, method="abs_max")
    b_q, b_s = q.fusedQuantizeMx(b, h, method="abs_max")
    alpha = torch.tensor([1.0 / 9.0], device=DEV)
    rm = lambda s, rows: s.view(torch.uint8).reshape(-1)[: rows * k // 32].reshape(rows, k // 32).contiguous().view(torch.float8_e8m0fnu)
    out_ada = q.matmul_ada_mxf4_bf16_tn(a_q, b_q, rm(a_s, m), rm(b_s, n), alpha)
    out_auto = q.matmul_mxf4_bf16_tn(a_q, b_q, to_blocked(a_s), to_blocked(b_s), alpha)     # M <= 32 -> skinny kernel
    with lab.forced(gemm_variant=24):
        out_tiled = lab.matmul_mxf4_bf16_tn(a_q, b_q, to_blocked(a_s), to_blocked(b_s), alpha)
    assert torch.equal(out_ada.view(torch.int16), out_tiled.view(torch.int16))
    assert torch.equal(out_auto.view(torch.int16), out_tiled.view(torch.int16))
    if m * n * k <= 1 << 29:
        ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a_q), _np(b_q), oracle.to_blocked(_np(rm(a_s, m))), oracle.to_blocked(_np(rm(b_s, n))),
                                      1.0 / 9.0, m, n, k)
        assert np.array_equal(_np(out_ada), ref)



@pytest.mark.parametrize("m,n,k", [(512, 768, 1024), (48, 512, 14336)])   # second shape: split-K with scratch from the caching allocator
def test_ops_are_hip_graph_capturable(q, m, n, k):
    """The reference's benchmarks time the ops under CUDA graphs (benchmarks/bench_mxfp4_sm100.py:216); the whole
    quantize -> swizzle -> GEMM chain (incl. the allocations inside the ops) must capture and replay."""
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(31)
    h = _hadamard(32)
    a = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    alpha = torch.tensor([1.0], device=DEV)
    b_q, b_s = q.fusedQuantizeMx(b, h, method="abs_max")
    b_sf = to_blocked(b_s)

    def chain():
        a_q, a_s = q.fusedQuantizeMx(a, h, method="abs_max")
        return q.matmul_mxf4_bf16_tn(a_q, b_q, to_blocked(a_s), b_sf, alpha)

    want = chain()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            chain()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        got = chain()
    a.mul_(-1.0)            # new input in the captured buffer: the replay must recompute
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, -want)      # value equality: an exact 0 keeps its sign bit under negation of the input


def test_tail_split_launch_matches_single_launch(q):
    """320 tiles of 256x256 on 256 CUs = 1.25 rounds: auto runs the heterogeneous launch ([r3]; rounds 1-2: a second launch of smaller
    tiles over the last 4 tile columns) -- 256 persistent workgroups + 256 quarter tiles; forcing the per-tile kernel runs ONE schedule."""
    from qutlass_amd.utils import to_blocked

    torch.manual_seed(41)
    m, n, k = 4096, 5120, 512
    h = _hadamard(32)
    a = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25.0
    a_q, a_s = q.fusedQuantizeMx(a, h, method="abs_max")
    b_q, b_s = q.fusedQuantizeMx(b, h, method="quest")
    asf, bsf = to_blocked(a_s), to_blocked(b_s)
    alpha = torch.tensor([0.5], device=DEV)
    out_split = q.matmul_mxf4_bf16_tn(a_q, b_q, asf, bsf, alpha)
    with lab.forced(gemm_variant=30):   # the per-tile deep schedule over the whole output, one launch
        out_one = lab.matmul_mxf4_bf16_tn(a_q, b_q, asf, bsf, alpha)
    assert torch.equal(out_split.view(torch.int16), out_one.view(torch.int16))
    rows = [0, 255, 256, 2047, 4095]
    sub = torch.tensor(rows, device=DEV)
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a_q[sub]), _np(b_q), oracle.to_blocked(_np(a_s).reshape(-1)[: m * k // 32].reshape(m, k // 32)[rows]),
                                  oracle.to_blocked(_np(b_s).reshape(-1)[: n * k // 32].reshape(n, k // 32)), 0.5, len(rows), n, k)
    assert np.array_equal(_np(out_split[sub]), ref)
