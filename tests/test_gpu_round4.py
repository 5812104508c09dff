"""Round-4 GPU parity tests (Python surface -> torch ops -> C ABI -> HIP kernels), against the pinned CPU oracle.

  * the persistent NVFP4 kernel (gemm_nvf4_pk.hip.h; reference: matmul_host_nvf4_bf16_tn + the CUTLASS tile scheduler,
    qutlass/csrc/gemm.cu:250-326, :73-75): whole-tile rounds, balanced rounds, and stream-K over a part-filled last round
    (tiles cut along K, the fp32 partial parked in scratch and added by the owner of the rest of the tile) must return the bytes
    of the per-tile kernels and of the oracle -- ragged M / N, two-stage K (every boundary snaps to a tile), long K, alpha != 1.
  * every finite non-negative e4m3 scale byte through the hardware fp8 -> f16 convert the persistent kernel uses
    (tests/nvfp4_test.py:196-203 dequantises with `scales.float()`; fused_quantize_nv never emits a sign bit or a NaN).
  * stream-K scratch hygiene: a captured graph replayed (same launch tag, same scratch) returns the same bytes every time.
Nothing here reads /root/reference.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)
import _benchlib as lab  # noqa: E402  (LAB build: forced schedules)

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


def _nv_operands(m, n, k, seed, lo=0x38, hi=0x48):
    """random e2m1 codes; e4m3 scales in [lo, hi) (default [1, 4), all mantissas: every partial sum stays exact in fp32, so any summation
    order and the fp64 oracle agree bit for bit)"""
    from qutlass_amd.utils import to_blocked

    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, generator=g)
    b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, generator=g)
    sa = torch.randint(lo, hi, (-(-m // 128) * 128, k // 16), dtype=torch.uint8, generator=g)
    sb = torch.randint(lo, hi, (-(-n // 128) * 128, k // 16), dtype=torch.uint8, generator=g)
    sa_b = to_blocked(sa.to(DEV).view(torch.float8_e4m3fn))
    sb_b = to_blocked(sb.to(DEV).view(torch.float8_e4m3fn))
    return a.to(DEV), b.to(DEV), sa, sb, sa_b, sb_b


def _oracle_rows(a, b, sa, sb, alpha, rows, n, k):
    pad = np.zeros((128 - len(rows), k // 16), np.uint8)
    return oracle.gemm_blockscaled(oracle.KIND_NVFP4, np.ascontiguousarray(_np(a)[rows]), _np(b), oracle.to_blocked(np.concatenate([np.ascontiguousarray(sa.numpy()[rows]), pad])),
                                   oracle.to_blocked(sb.numpy()), alpha, len(rows), n, k)


# (M, N, K): tiles of 256x256 on 256 CUs
#   4096 x 4096            256 tiles = one exact round (whole tiles, one per workgroup)
#   6144 x 4096            384 = 1.5 rounds: stream-K, every workgroup walks 1.5 tiles
#   4096 x 5120, K = 512   320 tiles of TWO stages: every range boundary snaps to a tile (stream-K region without a cut)
#   4360 x 5128 (ragged)   18 x 21 = 378 tiles, partial edge tiles in both dimensions, K = 1536 (6 stages)
#   8192 x 4608            576 = 2.25 rounds: one data-parallel round + a stream-K region of 320 tiles
#   2560 x 4096            160 tiles < one round: balanced (160 workgroups), no scratch needed
#   2800 x 6152            11 x 25 tiles: the last group of the raster is THREE tile rows tall (raster_decode's division by 3, common.hip.h)
PK_SHAPES = [(4096, 4096, 1024), (6144, 4096, 1024), (4096, 5120, 512), (4360, 5128, 1536), (8192, 4608, 768), (2560, 4096, 1024), (6144, 4096, 4096), (2800, 6152, 512)]


@pytest.mark.parametrize("m,n,k", PK_SHAPES)
def test_matmul_nvf4_persistent_kernel_equals_per_tile_kernels_and_oracle(q, m, n, k):
    a, b, sa, sb, sa_b, sb_b = _nv_operands(m, n, k, m + n + k)
    al = torch.tensor([0.5], device=DEV)
    lib = q._lib.load()
    out = q.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)                        # product: the auto rule (persistent kernel in balanced whole-tile rounds where it picks 256x256 tiles)
    outs = {}
    for v in (41, 42, 43, 5):   # per-tile 256x256 kernel (round 3) / persistent, whole tiles / persistent + stream-K / 128x128 tiles
        with lab.forced(nvf4_variant=v):
            outs[v] = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
    for v, o in outs.items():
        assert torch.equal(out.view(torch.int16), o.view(torch.int16)), (v, int((out.view(torch.int16) != o.view(torch.int16)).sum()))
    # the plain C entry (no scratch): balanced whole-tile rounds
    plain = torch.empty_like(out)
    assert lib.qutlass_amd_matmul_nvf4_bf16_tn(a.data_ptr(), b.data_ptr(), sa_b.data_ptr(), sb_b.data_ptr(), al.data_ptr(), plain.data_ptr(), m, n, k,
                                               torch.cuda.current_stream().cuda_stream) == 0
    assert torch.equal(out.view(torch.int16), plain.view(torch.int16))
    rows = sorted({0, 255, 256, m // 2, m - 257 if m > 600 else 1, m - 1})
    ref = _oracle_rows(a, b, sa, sb, 0.5, rows, n, k)
    assert np.array_equal(_np(out)[rows].view(np.uint16), ref.view(np.uint16))


def test_matmul_nvf4_persistent_kernel_every_e4m3_scale_byte(q):
    """Row r of A carries scale byte r % 127 (0x00 .. 0x7e: zero, the subnormals, every normal up to 448) in ALL its groups, B's scales are 1.0:
    each output is (scale) x (an exactly representable sum), so the kernel's hardware e4m3 -> f16 convert is compared with the oracle's decode
    bit for bit for every byte the quantizer can emit."""
    from qutlass_amd.utils import to_blocked

    m, n, k = 4096, 4096, 512
    g = torch.Generator(device="cpu").manual_seed(7)
    a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, generator=g).to(DEV)
    b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, generator=g).to(DEV)
    sa = (torch.arange(m, dtype=torch.int32) % 127).to(torch.uint8).view(m, 1).repeat(1, k // 16).contiguous()
    sb = torch.full((n, k // 16), 0x38, dtype=torch.uint8)
    sa_b = to_blocked(sa.to(DEV).view(torch.float8_e4m3fn))
    sb_b = to_blocked(sb.to(DEV).view(torch.float8_e4m3fn))
    al = torch.tensor([1.0], device=DEV)
    with lab.forced(nvf4_variant=42):
        out = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
    with lab.forced(nvf4_variant=5):
        old = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
    assert torch.equal(out.view(torch.int16), old.view(torch.int16))
    rows = list(range(0, 127)) + [4095]
    ref = _oracle_rows(a, b, sa, sb, 1.0, rows, n, k)
    assert np.array_equal(_np(out)[rows].view(np.uint16), ref.view(np.uint16))


def test_matmul_nvf4_stream_k_graph_replay_and_general_data(q):
    """(1) The arrival flags of the stream-K scratch carry a per-launch tag and are reset by their consumer: a captured graph (same tag, same scratch
    on every replay) must return the same bytes each time.  (2) Scales over six binades (sums no longer exact in fp32): the persistent kernel stays
    deterministic and within the north-star tolerance (1e-2 relative) of the oracle; the cut tiles' summation order (own part + parked part) differs
    from the single pass by at most an fp32 rounding."""
    m, n, k = 6144, 4096, 2048
    a, b, sa, sb, sa_b, sb_b = _nv_operands(m, n, k, 99, lo=0x20, hi=0x50)
    al = torch.tensor([1.0], device=DEV)
    with lab.forced(nvf4_variant=43):   # the stream-K form (lab build: the product walks whole tiles only)
        ref_out = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            o1 = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
            o2 = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
        for _ in range(3):
            o1.zero_(); o2.zero_()
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(o1.view(torch.int16), ref_out.view(torch.int16)) and torch.equal(o2.view(torch.int16), ref_out.view(torch.int16))
    prod = q.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
    assert torch.equal(prod.view(torch.int16), q.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al).view(torch.int16))   # the product path: deterministic
    rows = [0, 300, 3071, 6143]
    got = ref_out[rows].float().cpu().numpy()
    ref = torch.from_numpy(np.ascontiguousarray(_oracle_rows(a, b, sa, sb, 1.0, rows, n, k)).view(np.int16)).view(torch.bfloat16).float().numpy()
    denom = np.maximum(np.abs(ref), np.abs(ref).mean())
    assert float(np.max(np.abs(got - ref) / denom)) <= 1e-2
    with lab.forced(nvf4_variant=41):
        per_tile = lab.matmul_nvf4_bf16_tn(a, b, sa_b, sb_b, al)
    d = (ref_out.float() - per_tile.float()).abs() / per_tile.float().abs().clamp_min(float(per_tile.float().abs().mean()))
    assert float(d.max()) <= 2.0 ** -7      # at most one bf16 ulp where a cut tile's fp32 rounding moved a tie


@pytest.mark.parametrize("dev_index", [0, 1])
def test_pipeline_on_a_second_device_and_a_side_stream(q, dev_index):
    """quantize -> swizzle -> GEMM on cuda:<dev_index> and a non-default stream while the CURRENT device is cuda:0: exercises the DeviceGuard of every
    op (csrc/torch_ext.cpp; reference: include/common.h:40-45) and the per-device CU cache (capi.hip chip_cus).  MXFP4, NVFP4 and MXFP8 against the
    oracle.  Skips the second parametrisation on a single-GPU box (the driver's 8-GPU replica run is the other user of these paths)."""
    if dev_index >= torch.cuda.device_count():
        pytest.skip("needs a second GPU")
    from qutlass_amd.utils import to_blocked

    dev = torch.device(f"cuda:{dev_index}")
    torch.cuda.set_device(0)
    m, n, k = 384, 512, 1024
    g = torch.Generator(device="cpu").manual_seed(5 + dev_index)
    x = (torch.randn(m, k, generator=g) * 25).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 25).to(torch.bfloat16).to(dev)
    h = torch.ones(1, 1)
    while h.shape[0] < 32:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    h = (h * 32 ** -0.5).to(torch.bfloat16).to(dev)
    alpha = torch.tensor([1.0], device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        xq, xs = q.fusedQuantizeMx(x, h, method="abs_max")
        wq, ws = q.fusedQuantizeMx(w, h, method="abs_max")
        out = q.matmul_mxf4_bf16_tn(xq, wq, to_blocked(xs), to_blocked(ws), alpha)
        gs = torch.tensor([1.0], device=dev)
        h16 = torch.eye(16, dtype=torch.bfloat16, device=dev)
        xq4, xs4 = q.fusedQuantizeNv(x, h16, gs)
        wq4, ws4 = q.fusedQuantizeNv(w, h16, gs)
        out4 = q.matmul_nvf4_bf16_tn(xq4, wq4, to_blocked(xs4), to_blocked(ws4), alpha)
    side.synchronize()
    assert out.device == dev and out4.device == dev and torch.cuda.current_device() == 0
    hb = _np(h)
    rq, rs, _ = oracle.fused_quantize_mx(_np(x), hb, oracle.ABS_MAX)
    assert np.array_equal(_np(xs).reshape(-1)[: rs.size], rs)
    ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(xq), _np(wq), _np(to_blocked(xs)), _np(to_blocked(ws)), 1.0, m, n, k)
    assert np.array_equal(_np(out).view(np.uint16), ref.view(np.uint16))
    ref4 = oracle.gemm_blockscaled(oracle.KIND_NVFP4, _np(xq4), _np(wq4), _np(to_blocked(xs4)), _np(to_blocked(ws4)), 1.0, m, n, k)
    assert np.array_equal(_np(out4).view(np.uint16), ref4.view(np.uint16))


# ------------------------------------------------------------------------------------------------
# stream-K form of the MX persistent kernels (gemm_mx_deepp.hip.h, streamk.hip.h; LAB variant 89: correct, deterministic, measured slower -- see capi.hip)
# ------------------------------------------------------------------------------------------------
def _rand_mx(m, n, k, seed, fp8=False):
    """random operands in the exact regime (block exponents within +-3): any K order gives the same fp32 sums for fp4"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if fp8:
        a = torch.randint(0, 256, (m, k), dtype=torch.uint8, generator=g)
        b = torch.randint(0, 256, (n, k), dtype=torch.uint8, generator=g)
        a = torch.where((a & 0x7f) == 0x7f, a & 0x80, a)
        b = torch.where((b & 0x7f) == 0x7f, b & 0x80, b)
    else:
        a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, generator=g)
        b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, generator=g)
    sa = torch.randint(124, 131, (m, k // 32), dtype=torch.uint8, generator=g)
    sb = torch.randint(124, 131, (n, k // 32), dtype=torch.uint8, generator=g)
    return a, b, sa, sb


def _mx_rows(kind, a, b, sa, sb, alpha, rows, n, k):
    return oracle.gemm_blockscaled(kind, a[rows].numpy(), b.numpy(), oracle.to_blocked(sa[rows].numpy()), oracle.to_blocked(sb.numpy()), alpha, len(rows), n, k)


# tiles of 256x256 on 256 CUs: 384 = 1.5 rounds (every tile of the stream is cut or whole, 24 / 48 stages per workgroup), 320 = 1.25, ragged 378 tiles,
# 576 = 2.25 rounds (one whole-tile round first), K = 2048 .. 8192 (8 .. 32 stages of fp4, 16 .. 64 of fp8)
@pytest.mark.parametrize("m,n,k", [(6144, 4096, 4096), (4096, 5120, 2048), (4360, 5128, 2048), (8192, 4608, 2048), (6144, 4096, 8192)])
def test_stream_k_mxfp4_equals_persistent_kernel_and_oracle(q, m, n, k):
    from qutlass_amd.utils import to_blocked

    a, b, sa, sb = _rand_mx(m, n, k, seed=m + n + k)
    ad, bd = a.to(DEV), b.to(DEV)
    asf = to_blocked(sa.to(DEV).view(torch.float8_e8m0fnu))
    bsf = to_blocked(sb.to(DEV).view(torch.float8_e8m0fnu))
    alpha = torch.tensor([0.5], device=DEV)
    with lab.forced(gemm_variant=89):
        sk = lab.matmul_mxf4_bf16_tn(ad, bd, asf, bsf, alpha)
        sk2 = lab.matmul_mxf4_bf16_tn(ad, bd, asf, bsf, alpha)
    with lab.forced(gemm_variant=90, pp_flags=1 | 64):   # ONE persistent launch over whole tiles (balanced rounds)
        one = lab.matmul_mxf4_bf16_tn(ad, bd, asf, bsf, alpha)
    assert torch.equal(sk.view(torch.int16), sk2.view(torch.int16))
    assert torch.equal(sk.view(torch.int16), one.view(torch.int16)), int((sk.view(torch.int16) != one.view(torch.int16)).sum())
    got = q.matmul_mxf4_bf16_tn(ad, bd, asf, bsf, alpha)  # the product's own choice (balanced rounds or the heterogeneous launch)
    assert torch.equal(got.view(torch.int16), one.view(torch.int16))
    rows = sorted({0, 255, 256, m // 2, m - 257, m - 1})
    ref = _mx_rows(oracle.KIND_MXFP4, a, b, sa, sb, 0.5, rows, n, k)
    assert np.array_equal(_np(sk)[rows].view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("m,n,k", [(6144, 4096, 2048), (4096, 5120, 4096), (4360, 5128, 1024)])
def test_stream_k_mxfp8_equals_persistent_kernel_and_oracle(q, m, n, k):
    from qutlass_amd.utils import to_blocked

    a, b, sa, sb = _rand_mx(m, n, k, seed=m + n + k + 1, fp8=True)
    ad, bd = a.to(DEV).view(torch.float8_e4m3fn), b.to(DEV).view(torch.float8_e4m3fn)
    asf = to_blocked(sa.to(DEV).view(torch.float8_e8m0fnu))
    bsf = to_blocked(sb.to(DEV).view(torch.float8_e8m0fnu))
    alpha = torch.tensor([1.0], device=DEV)
    with lab.forced(gemm_variant=89):
        sk = lab.matmul_mxf8_bf16_tn(ad, bd, asf, bsf, alpha)
        sk2 = lab.matmul_mxf8_bf16_tn(ad, bd, asf, bsf, alpha)
    with lab.forced(gemm_variant=90, pp_flags=1 | 64):
        one = lab.matmul_mxf8_bf16_tn(ad, bd, asf, bsf, alpha)
    assert torch.equal(sk.view(torch.int16), sk2.view(torch.int16))                      # deterministic
    # e4m3 x e4m3 products are not all exact in fp32 sums: the cut tiles' order (parked last part + first part) may move a bf16 tie
    d = (sk.float() - one.float()).abs()
    assert float((d / one.float().abs().clamp_min(float(one.float().abs().mean()))).max()) <= 2.0 ** -7
    assert float((sk.view(torch.int16) != one.view(torch.int16)).float().mean()) <= 2e-3
    rows = sorted({0, 255, 256, m // 2, m - 257, m - 1})
    ref = oracle.bf16_bits_to_f32(_mx_rows(oracle.KIND_MXFP8_TN, a, b, sa, sb, 1.0, rows, n, k)).astype(np.float64)
    g = oracle.bf16_bits_to_f32(_np(sk)[rows]).astype(np.float64)
    assert (np.abs(g - ref) <= np.abs(ref) / 128.0 + 1e-4 * np.abs(ref).max()).all()


def test_stream_k_mx_graph_replay(q):
    """the arrival flags carry a per-launch tag and are reset by their consumer: a captured graph (same tag, same scratch) replays to the same bytes"""
    from qutlass_amd.utils import to_blocked

    m, n, k = 6144, 4096, 4096
    a, b, sa, sb = _rand_mx(m, n, k, seed=3)
    ad, bd = a.to(DEV), b.to(DEV)
    asf = to_blocked(sa.to(DEV).view(torch.float8_e8m0fnu))
    bsf = to_blocked(sb.to(DEV).view(torch.float8_e8m0fnu))
    alpha = torch.tensor([1.0], device=DEV)
    with lab.forced(gemm_variant=89):
        ref = lab.matmul_mxf4_bf16_tn(ad, bd, asf, bsf, alpha)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                lab.matmul_mxf4_bf16_tn(ad, bd, asf, bsf, alpha)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            o1 = lab.matmul_mxf4_bf16_tn(ad, bd, asf, bsf, alpha)
            o2 = lab.matmul_mxf4_bf16_tn(ad, bd, asf, bsf, alpha)
        for _ in range(3):
            o1.zero_(); o2.zero_()
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(o1.view(torch.int16), ref.view(torch.int16)) and torch.equal(o2.view(torch.int16), ref.view(torch.int16))


# ------------------------------------------------------------------------------------------------
# [r4] backward_t_bf16 / backward_qt_bf16: wave-owned output lines (bwd_quant_tw_kernel, quartet_bwd.hip.h).  Three kernels sit behind one entry
# point (capi.hip: bwd_kernel_choice): 1 = the round-3 kernel (small tensors), 2 = units of 4 scale groups (QT), 3 = units of 8 (T).  Each is
# forced through the LAB build on ragged shapes against the oracle (quartet_bwd_sm120.cu:304-323 / :407-426), and the product rule is checked
# against the round-3 kernel, byte for byte, on tensors large enough to take the new kernels.
# ------------------------------------------------------------------------------------------------
def _hadamard32():
    h = torch.ones(1, 1)
    while h.shape[0] < 32:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * 32 ** -0.5).to(torch.bfloat16).to(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("B,N,M", [(2, 512, 320), (1, 96, 64), (3, 32, 1056), (1, 2080, 160), (1, 416, 8 * 37)])
def test_backward_wave_owned_kernels_equal_the_oracle_on_ragged_shapes(q, variant, B, N, M):
    """group counts that are not multiples of 4 / 8, m-tile counts that are not multiples of 4, a last m-tile of 8 .. 56 rows, batches: every kernel
    behind backward_t_bf16 / backward_qt_bf16 returns the oracle's scale bytes exactly and its codes up to the reference's own tolerance."""
    rng = np.random.default_rng(B * 1000 + N + M + variant)
    h = _hadamard32()
    with lab.forced(bwd_variant=variant):
        x = torch.from_numpy(rng.standard_normal((B, N, M)).astype(np.float32) * 20.0).to(torch.bfloat16).to(DEV)
        e2m1, e8m0 = lab.backward_t_bf16(x, h)
        rq, rs = oracle.backward_t_bf16(_np(x), _np(h), acc_model=1)
        assert np.array_equal(_np(e8m0), rs), int((_np(e8m0) != rs).sum())
        eq = oracle.codes_equal_mod_zero_sign(_np(e2m1).reshape(rq.shape), rq)
        assert int((~eq).sum()) <= 1e-4 * eq.size, int((~eq).sum())
        if M % 32 == 0:
            codes = rng.integers(0, 256, size=(B, N, M // 2), dtype=np.uint8)
            scales = rng.integers(110, 140, size=(B, N, M // 32), dtype=np.uint8)
            codes[:, -32:, : M // 4] = 0              # all-zero groups: the reference's 0 * inf = NaN -> code 7 path
            e2m1, e8m0 = lab.backward_qt_bf16(torch.from_numpy(codes).to(DEV), torch.from_numpy(scales).to(DEV), h, torch.tensor([3.0], device=DEV))
            rq, rs = oracle.backward_qt_bf16(codes, scales, _np(h), 3.0, acc_model=1)
            assert np.array_equal(_np(e8m0), rs), int((_np(e8m0) != rs).sum())
            eq = oracle.codes_equal_mod_zero_sign(_np(e2m1).reshape(rq.shape), rq)
            assert int((~eq).sum()) <= 1e-4 * eq.size, int((~eq).sum())


@pytest.mark.gpu
def test_backward_product_rule_takes_the_wave_owned_kernels_and_equals_the_round3_kernel(q):
    """tensors above the thresholds of bwd_kernel_choice (QT: 3 units per CU, T: 6), ragged in M and in the group count: the PRODUCT library's bytes
    (whatever kernel its rule picks) equal the round-3 kernel's, forced through the lab build -- and equal the lab build's own rule."""
    g = torch.Generator(device=DEV).manual_seed(5)
    h = _hadamard32()
    B, N, M = 1, 4096 + 96, 6152                       # T: 131 groups (not a multiple of 8), 97 m-tiles, the last one of 8 rows
    x = torch.randn(B, N, M, dtype=torch.bfloat16, device=DEV, generator=g) * 7.0
    pq, ps = q.backward_t_bf16(x, h)
    with lab.forced(bwd_variant=1):
        oq, osf = lab.backward_t_bf16(x, h)
    aq, asf = lab.backward_t_bf16(x, h)
    assert torch.equal(pq.view(torch.uint8).reshape(oq.shape), oq) and torch.equal(ps.view(torch.uint8).reshape(osf.shape), osf)
    assert torch.equal(aq, oq) and torch.equal(asf, osf)
    M = 6144 + 32                                      # QT: M a multiple of 32
    xq = torch.randint(0, 256, (B, N, M // 2), dtype=torch.uint8, device=DEV, generator=g)
    xs = torch.randint(118, 134, (B, N, M // 32), dtype=torch.uint8, device=DEV, generator=g)
    alpha = torch.tensor([0.61], device=DEV)
    pq, ps = q.backward_qt_bf16(xq, xs.view(torch.float8_e8m0fnu), h, alpha)
    with lab.forced(bwd_variant=1):
        oq, osf = lab.backward_qt_bf16(xq, xs, h, alpha)
    aq, asf = lab.backward_qt_bf16(xq, xs, h, alpha)
    assert torch.equal(pq.view(torch.uint8).reshape(oq.shape), oq) and torch.equal(ps.view(torch.uint8).reshape(osf.shape), osf)
    assert torch.equal(aq, oq) and torch.equal(asf, osf)


# ------------------------------------------------------------------------------------------------
# [r4] mxfp4_transpose_mxfp8 with wave-owned output lines (mxfp4_transpose_mxfp8_tw_kernel): both unit shapes against the oracle
# (quartet_bwd_sm120.cu mxfp4_transpose_mxfp8 kernel; oracle.mxfp4_transpose_mxfp8) and against the one-shot kernel, incl. padded rows.
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("variant", [4, 2, 128, 3, 0])     # wave-owned lines / segments (lab), 4-wave one-shot, 8-wave one-shot (256 rows), the product rule
@pytest.mark.parametrize("m,m_pad,n", [(128, 128, 256), (384, 384, 768), (200, 256, 512), (1, 128, 256), (129, 256, 1280), (2048, 2048, 2304), (3000, 3072, 5632)])
def test_transposer_wave_owned_kernels_equal_the_oracle(q, variant, m, m_pad, n):
    rng = np.random.default_rng(m * 7 + n + variant)
    codes = rng.integers(0, 256, size=(m, n // 2), dtype=np.uint8)
    scales = rng.integers(117, 137, size=(m, n // 32), dtype=np.uint8)
    if m >= 64:
        codes[32:64, : n // 4] = 0                      # an all-zero block along m for a quarter of the columns (amax = 0 -> e8m0 127)
    with lab.forced(transpose_nc=variant):
        y, sf = lab.mxfp4_transpose_mxfp8_rows(torch.from_numpy(codes).to(DEV), torch.from_numpy(scales).to(DEV), m, m_pad, n)
    pc = np.zeros((m_pad, n // 2), np.uint8); pc[:m] = codes
    ps = np.full((m_pad, n // 32), 127, np.uint8); ps[:m] = scales
    ry, rs = oracle.mxfp4_transpose_mxfp8(pc, ps)
    assert np.array_equal(_np(sf), np.asarray(rs).reshape(_np(sf).shape)), int((_np(sf) != np.asarray(rs).reshape(_np(sf).shape)).sum())
    assert np.array_equal(_np(y), np.asarray(ry).reshape(_np(y).shape)), int((_np(y) != np.asarray(ry).reshape(_np(y).shape)).sum())


# ------------------------------------------------------------------------------------------------
# [r4] backward_bf16_square_double_mxfp8 with 512 columns per workgroup (16 waves: 16-byte row-scale pieces) -- and the lab's 1024-column form --
# against the oracle (quartet_bwd_sm120.cu:511-621) and the 4-wave form, incl. tensors large enough for the product rule to take it.
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 4, 8])
@pytest.mark.parametrize("m,n", [(128, 1024), (384, 2048), (256, 3072), (1024, 5120)])
def test_square_double_wide_workgroups_equal_the_oracle(q, variant, m, n):
    if variant == 8 and n % 1024:
        pytest.skip("1024-column workgroups need n % 1024 == 0")
    rng = np.random.default_rng(m + n + variant)
    x = torch.from_numpy(rng.standard_normal((m, n)).astype(np.float32) * float(rng.choice([1e-3, 1.0, 300.0]))).to(torch.bfloat16).to(DEV)
    x[32:64, 128:256] = 0          # an all-zero 32 x 32 block row: shared exponent 127
    with lab.forced(transpose_nc=variant):
        y, rs, cs = lab.backward_bf16_square_double_mxfp8(x)
    ry, rrs, rcs = oracle.backward_bf16_square_double_mxfp8(_np(x))
    assert np.array_equal(_np(y), ry) and np.array_equal(_np(rs), rrs) and np.array_equal(_np(cs), rcs)


@pytest.mark.gpu
def test_square_double_product_rule_takes_the_wide_form_and_equals_the_four_wave_form(q):
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(4000, 4608, dtype=torch.bfloat16, device=DEV, generator=g) * 11.0      # 32 row blocks (the last one padded in the kernel) x 9 wide tiles = 288 workgroups
    py, prs, pcs = q.backward_bf16_square_double_mxfp8(x)                                    # product library, its own rule, in-kernel row padding
    xp = torch.zeros(4096, 4608, dtype=torch.bfloat16, device=DEV)
    xp[:4000] = x
    with lab.forced(transpose_nc=1):
        oy, ors, ocs = lab.backward_bf16_square_double_mxfp8(xp)
    assert torch.equal(py.view(torch.uint8), oy) and torch.equal(prs.view(torch.uint8), ors) and torch.equal(pcs.view(torch.uint8), ocs)
