"""GPU: randomised shape sweep of every op against the CPU oracle (seeded, small sizes so the oracle stays fast).
Complements test_gpu_parity.py, which pins the golden fixtures and the reference's own test shapes: here ragged
M / N, K tails, every auto-selected tile configuration and leading batch dimensions are drawn at random."""
import numpy as np
import pytest
import torch

import oracle
import _benchlib as lab  # the LAB build of the library (forced operand paths), tests/_benchlib.py

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
import os

SEED = int(os.environ.get("QAMD_FUZZ_SEED", "0"))   # extra sweeps: QAMD_FUZZ_SEED=1,2,... python -m pytest tests/test_gpu_fuzz.py -m gpu


@pytest.fixture(scope="module")
def q():
    import qutlass_amd

    return qutlass_amd


def _np(t):
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.uint16).numpy()
    if t.element_size() == 1:
        return t.view(torch.uint8).numpy()
    return t.numpy()


def _hadamard(n):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(DEV)


def _rand_codes(rng, rows, kbytes):
    return torch.from_numpy(rng.integers(0, 256, size=(rows, kbytes), dtype=np.uint8)).to(DEV)


def test_fuzz_matmul_mxf4(q):
    from qutlass_amd.utils import to_blocked

    rng = np.random.default_rng(101 + 1000 * SEED)
    ms = [1, 2, 7, 16, 31, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256, 300, 520]
    for it in range(40):
        m = int(rng.choice(ms))
        n = int(rng.integers(1, 90)) * 8
        k = int(rng.integers(1, 12)) * 128
        a, b = _rand_codes(rng, m, k // 2), _rand_codes(rng, n, k // 2)
        # scale exponents within 3 binades: every fp32 partial sum is exact (K * 144 * 2^(2*3) < 2^24 for K <= 1408), so
        # any summation order must reproduce the fp64 oracle bit for bit; every 4th case draws 16 binades, where fp32
        # accumulation may round and the bar is the north-star tolerance (<= 1e-2 relative to max|ref|)
        wide = it % 4 == 3
        lo, hi = (118, 134) if wide else (126, 129)
        sa = torch.from_numpy(rng.integers(lo, hi, size=(m, k // 32), dtype=np.uint8)).to(DEV)
        sb = torch.from_numpy(rng.integers(lo, hi, size=(n, k // 32), dtype=np.uint8)).to(DEV)
        alpha = torch.tensor([float(rng.choice([1.0, 0.5, 0.25]))], device=DEV)
        e8 = torch.float8_e8m0fnu
        out = q.matmul_mxf4_bf16_tn(a, b, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), alpha)
        out2 = q.matmul_ada_mxf4_bf16_tn(a, b, sa.view(e8), sb.view(e8), alpha)
        ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)),
                                      float(alpha.item()), m, n, k)
        if wide:
            want = oracle.bf16_bits_to_f32(ref).astype(np.float64)
            for o in (out, out2):
                got = oracle.bf16_bits_to_f32(_np(o)).astype(np.float64)
                assert np.abs(got - want).max() <= 1e-2 * np.abs(want).max(), (it, m, n, k)
        else:
            assert np.array_equal(_np(out), ref), (it, m, n, k, int((_np(out) != ref).sum()))
            assert np.array_equal(_np(out2), ref), ("ada", it, m, n, k)


def test_fuzz_matmul_mxf4_long_k_split(q):
    """Long K with a small output: the ring kernel splits K over grid.y (scratch from the caching allocator) and a
    second kernel sums the partials.  Scale exponents within 2 binades keep every fp32 partial sum exact for K <= 12288
    (K * 144 * 2^2 < 2^24), so the split result must equal the fp64 oracle bit for bit."""
    from qutlass_amd.utils import to_blocked

    rng = np.random.default_rng(105 + 1000 * SEED)
    e8 = torch.float8_e8m0fnu
    nsplit = 0
    for it in range(8):
        m = int(rng.choice([1, 16, 33, 40, 64, 100, 128]))
        n = int(rng.integers(1, 64)) * 8
        k = int(rng.choice([8192, 8320, 9216, 11008 // 128 * 128, 12288]))
        a, b = _rand_codes(rng, m, k // 2), _rand_codes(rng, n, k // 2)
        sa = torch.from_numpy(rng.integers(127, 129, size=(m, k // 32), dtype=np.uint8)).to(DEV)
        sb = torch.from_numpy(rng.integers(127, 129, size=(n, k // 32), dtype=np.uint8)).to(DEV)
        alpha = torch.tensor([float(rng.choice([1.0, 0.5]))], device=DEV)
        nsplit += q._lib.load().qutlass_amd_gemm_splitk_workspace_bytes(4, m, n, k) > 0
        out = q.matmul_mxf4_bf16_tn(a, b, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), alpha)
        ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)),
                                      float(alpha.item()), m, n, k)
        assert np.array_equal(_np(out), ref), (it, m, n, k, int((_np(out) != ref).sum()))
    assert nsplit >= 6   # these shapes are in the split-K regime (<= 128 tiles, >= 32 stages)


def test_fuzz_matmul_nvf4_and_mxf8(q):
    from qutlass_amd.utils import to_blocked

    rng = np.random.default_rng(102 + 1000 * SEED)
    for it in range(16):
        m, n = int(rng.choice([1, 16, 40, 128, 136, 300])), int(rng.integers(1, 50)) * 8
        k = int(rng.integers(1, 20)) * 32
        a, b = _rand_codes(rng, m, k // 2), _rand_codes(rng, n, k // 2)
        sa = torch.from_numpy(rng.integers(0x28, 0x48, size=(m, k // 16), dtype=np.uint8)).to(DEV)
        sb = torch.from_numpy(rng.integers(0x28, 0x48, size=(n, k // 16), dtype=np.uint8)).to(DEV)
        alpha = torch.tensor([1.0], device=DEV)
        e4 = torch.float8_e4m3fn
        out = q.matmul_nvf4_bf16_tn(a, b, to_blocked(sa.view(e4)), to_blocked(sb.view(e4)), alpha)
        ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 1.0, m, n, k)
        assert np.array_equal(_np(out), ref), (it, m, n, k, int((_np(out) != ref).sum()))
    # [r3] long K against small outputs: the NVFP4 split-K path (ranges of an even number of 256-element stages + the reduce pass), K tails included -- the plan's own
    # choice (which at these widths is mostly the small-batch kernel) AND a forced tile x split of the lab build, both against the full oracle
    for it in range(8):
        m, n = int(rng.choice([40, 100, 136, 200])), int(rng.integers(8, 48)) * 8
        k = int(rng.integers(96, 224)) * 32
        a, b = _rand_codes(rng, m, k // 2), _rand_codes(rng, n, k // 2)
        sa = torch.from_numpy(rng.integers(0x30, 0x48, size=(m, k // 16), dtype=np.uint8)).to(DEV)
        sb = torch.from_numpy(rng.integers(0x30, 0x48, size=(n, k // 16), dtype=np.uint8)).to(DEV)
        alpha = torch.tensor([0.25], device=DEV)
        e4 = torch.float8_e4m3fn
        ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, _np(a), _np(b), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 0.25, m, n, k)
        out = q.matmul_nvf4_bf16_tn(a, b, to_blocked(sa.view(e4)), to_blocked(sb.view(e4)), alpha)
        assert np.array_equal(_np(out), ref), ("nvf4 long K", it, m, n, k, int((_np(out) != ref).sum()))
        forced = int(rng.choice([112, 114, 118, 122, 124, 132, 134]))   # 128x128 / 128x64 / 64x64 tiles x 2 / 4 / 8 K ranges
        with lab.forced(nvf4_variant=forced):
            out = lab.matmul_nvf4_bf16_tn(a, b, to_blocked(sa.view(e4)), to_blocked(sb.view(e4)), alpha)
        assert np.array_equal(_np(out), ref), ("nvf4 split", forced, it, m, n, k, int((_np(out) != ref).sum()))
    for it in range(16):
        m, n = int(rng.choice([16, 48, 128, 144, 272])), int(rng.integers(1, 50)) * 8
        k = int(rng.integers(1, 20)) * 32
        x = (torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32)) * 4).to(torch.float8_e4m3fn).to(DEV)
        y = (torch.from_numpy(rng.standard_normal((n, k)).astype(np.float32)) * 4).to(torch.float8_e4m3fn).to(DEV)
        sa = torch.from_numpy(rng.integers(120, 131, size=(m, k // 32), dtype=np.uint8)).to(DEV)
        sb = torch.from_numpy(rng.integers(120, 131, size=(n, k // 32), dtype=np.uint8)).to(DEV)
        alpha = torch.tensor([1.0], device=DEV)
        e8 = torch.float8_e8m0fnu
        out = q.matmul_mxf8_bf16_tn(x, y, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), alpha)
        ref = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, _np(x), _np(y), oracle.to_blocked(_np(sa)), oracle.to_blocked(_np(sb)), 1.0, m, n, k)
        got = oracle.bf16_bits_to_f32(_np(out)).astype(np.float64)
        want = oracle.bf16_bits_to_f32(ref).astype(np.float64)
        # fp32 accumulation of 8-bit-significand products: 1 bf16 ulp + 1e-4 * max|ref| (cancelling outputs next to 2e4-sized
        # partial sums were seen 3e-5 * max off, identically in every tile configuration / schedule)
        assert (np.abs(got - want) <= np.abs(want) / 128.0 + 1e-4 * np.abs(want).max()).all(), (it, m, n, k)
        x_km = x.view(torch.uint8).T.contiguous().view(torch.float8_e4m3fn)
        for path in (0, 63, 61, 62):
            impl = q if path == 0 else lab
            with lab.forced(gemm_variant=path):
                out_nn = impl.matmul_mxf8_bf16_nn(x_km, y, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), alpha)
            if path in (61, 63):   # different tile configuration than the auto TN kernel: compare with the oracle tolerance
                gnn = oracle.bf16_bits_to_f32(_np(out_nn)).astype(np.float64)
                assert (np.abs(gnn - want) <= np.abs(want) / 128.0 + 1e-4 * np.abs(want).max()).all(), ("nn fused", it, m, n, k)
            else:
                assert torch.equal(out_nn.view(torch.int16), out.view(torch.int16)), ("nn", path, it, m, n, k)


def test_fuzz_quantizers_and_swizzle(q):
    rng = np.random.default_rng(103 + 1000 * SEED)
    for it in range(24):
        R = int(rng.choice([32, 64, 128]))
        lead = tuple(int(v) for v in rng.integers(1, 5, size=int(rng.integers(0, 3))))
        rows, cols = int(rng.integers(1, 40)), int(rng.integers(1, 6)) * R
        x = torch.from_numpy(rng.standard_normal(lead + (rows, cols)).astype(np.float32) * float(rng.choice([0.01, 1.0, 25.0, 3000.0]))).to(torch.bfloat16).to(DEV)
        h = _hadamard(R) if rng.random() < 0.7 else (torch.randn(R, R) * 0.2).to(torch.bfloat16).to(DEV)
        method = str(rng.choice(["quest", "abs_max"]))
        mask = method == "quest" and R == 32 and rng.random() < 0.5
        res = q.fusedQuantizeMx(x, h, method=method, return_mask=mask)
        rq, rs, rm = oracle.fused_quantize_mx(_np(x), _np(h), oracle.QUEST if method == "quest" else oracle.ABS_MAX, with_mask=mask, acc_model=1)
        got_s = _np(res[1]).reshape(-1)[: rs.size]
        sbad = int((got_s != rs).sum())
        assert sbad <= max(1, 2e-3 * rs.size), (it, R, method, sbad)        # MFMA vs exact-sum order: rare binade flips only
        same_grp = got_s == rs                                                # groups whose scale agrees
        eq = oracle.codes_equal_mod_zero_sign(_np(res[0]).reshape(-1), rq)     # one flag per code
        assert int((~eq & same_grp.repeat(32)).sum()) <= max(2, 2e-3 * eq.size), (it, R, method)
        if mask:
            mm = _np(res[2]).reshape(-1)                                       # 4 mask bytes per group
            assert int(((mm != rm) & same_grp.repeat(4)).sum()) <= max(1, 1e-3 * rm.size)
    for it in range(12):
        r, c = int(rng.integers(1, 700)), int(rng.integers(1, 70))
        a = torch.from_numpy(rng.integers(0, 256, size=(r, c), dtype=np.uint8)).to(DEV)
        from qutlass_amd.utils import to_blocked

        assert np.array_equal(_np(to_blocked(a)), oracle.to_blocked(_np(a))), (it, r, c)


def test_fuzz_backward_ops(q):
    rng = np.random.default_rng(104 + 1000 * SEED)
    h = _hadamard(32)
    for it in range(10):
        B, N, M = int(rng.integers(1, 4)), int(rng.integers(1, 9)) * 32, int(rng.integers(1, 40)) * 8
        x = torch.from_numpy(rng.standard_normal((B, N, M)).astype(np.float32) * 25.0).to(torch.bfloat16).to(DEV)
        e2m1, e8m0 = q.backward_t_bf16(x, h)
        rq, rs = oracle.backward_t_bf16(_np(x), _np(h), acc_model=1)
        assert np.array_equal(_np(e8m0), rs), (it, B, N, M)
        eq = oracle.codes_equal_mod_zero_sign(_np(e2m1).reshape(rq.shape), rq)
        assert int((~eq).sum()) <= max(2, 1e-3 * eq.size), (it, B, N, M)
    for it in range(8):   # [r3] backward_qt_bf16: ragged unit counts (m-tiles not a multiple of 4, groups not a multiple of 8), random alpha
        B, N, M = int(rng.integers(1, 3)), int(rng.integers(1, 20)) * 32, int(rng.integers(1, 24)) * 32
        alpha = float(np.float32(rng.uniform(0.05, 20.0)))
        xq = torch.from_numpy(rng.integers(0, 256, size=(B, N, M // 2), dtype=np.uint8)).to(DEV)
        xs = torch.from_numpy(rng.integers(112, 142, size=(B, N, M // 32), dtype=np.uint8)).to(DEV)
        e2m1, e8m0 = q.backward_qt_bf16(xq, xs.view(torch.float8_e8m0fnu), h, torch.tensor([alpha], device=DEV))
        rq, rs = oracle.backward_qt_bf16(_np(xq), _np(xs), _np(h), alpha, acc_model=1)
        assert np.array_equal(_np(e8m0), rs), (it, B, N, M, alpha)
        eq = oracle.codes_equal_mod_zero_sign(_np(e2m1).reshape(rq.shape), rq)
        assert int((~eq).sum()) <= max(2, 1e-3 * eq.size), (it, B, N, M, alpha)
    for it in range(6):
        m, n = int(rng.integers(1, 5)) * 128, int(rng.integers(1, 5)) * 128
        x = torch.from_numpy(rng.standard_normal((m, n)).astype(np.float32) * float(rng.choice([1e-3, 1.0, 500.0]))).to(torch.bfloat16).to(DEV)
        y, rs, cs = q.backward_bf16_square_double_mxfp8(x)
        ry, rrs, rcs = oracle.backward_bf16_square_double_mxfp8(_np(x))
        assert np.array_equal(_np(y), ry) and np.array_equal(_np(rs), rrs) and np.array_equal(_np(cs), rcs), (it, m, n)
    for it in range(6):
        m, n = int(rng.integers(1, 4)) * 256, int(rng.integers(1, 4)) * 256
        xq = _rand_codes(rng, m, n // 2)
        xs = torch.from_numpy(rng.integers(110, 140, size=(m, n // 32), dtype=np.uint8)).to(DEV)
        y, e = q.mxfp4_transpose_mxfp8(xq, xs.view(torch.float8_e8m0fnu))
        ry, re = oracle.mxfp4_transpose_mxfp8(_np(xq), _np(xs))
        assert np.array_equal(_np(e), re) and np.array_equal(_np(y), ry), (it, m, n)


def test_fuzz_persistent_deep_kernels(q):
    """Outputs of >= 192 tiles of 256x256 run the persistent deep kernels (gemm_mx_deepp / gemm_mx_deepp8): ragged M / N, K tails,
    odd and even stage counts, 1..3 rounds of tiles per workgroup, balanced grids and tail splits, alpha != 1 -- against the
    oracle on sampled rows (first / last / tile-boundary rows + a random draw).  K stays small so the oracle stays fast."""
    from qutlass_amd.utils import to_blocked

    rng = np.random.default_rng(107 + 1000 * SEED)
    e8 = torch.float8_e8m0fnu
    for it in range(10):
        fp8 = it % 2 == 1
        e5 = fp8 and it % 4 == 3
        m = int(rng.integers(2600, 6200)) if it % 3 else int(rng.choice([3072, 4096, 5120]))
        n = int(rng.integers(2600 // 8, 9000 // 8)) * 8
        if -(-m // 256) * -(-n // 256) < 192:
            n = -(-192 // -(-m // 256)) * 256 + 8
        k = int(rng.integers(1, 13)) * (32 if fp8 else 128)
        alpha = float(rng.choice([1.0, 0.5, -0.25]))
        rows = sorted({0, 1, 255, 256, 257, m - 257, m - 256, m - 1} | set(rng.integers(0, m, 24).tolist()))
        ri = torch.tensor(rows, device=DEV)
        al = torch.tensor([alpha], device=DEV)
        if not fp8:
            a, b = _rand_codes(rng, m, k // 2), _rand_codes(rng, n, k // 2)
            sa = torch.from_numpy(rng.integers(126, 129, size=(m, k // 32), dtype=np.uint8)).to(DEV)
            sb = torch.from_numpy(rng.integers(126, 129, size=(n, k // 32), dtype=np.uint8)).to(DEV)
            out = q.matmul_mxf4_bf16_tn(a, b, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), al)
            ref = oracle.gemm_blockscaled(oracle.KIND_MXFP4, _np(a[ri]), _np(b), oracle.to_blocked(_np(sa[ri])), oracle.to_blocked(_np(sb)), alpha, len(rows), n, k)
            got = _np(out[ri])
            assert np.array_equal(got, ref), (it, m, n, k, int((got != ref).sum()))
        else:
            x = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32) * 4)
            y = torch.from_numpy(rng.standard_normal((n, k)).astype(np.float32) * 4)
            xs = x * torch.exp2(torch.from_numpy(rng.integers(-6, 7, (m, 1)).astype(np.float32))) if e5 else x   # gradient-like row ranges for e5m2
            a = xs.clamp(-448.0 if not e5 else -57344.0, 448.0 if not e5 else 57344.0).to(torch.float8_e5m2 if e5 else torch.float8_e4m3fn).to(DEV)
            b = y.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(DEV)
            sa = torch.from_numpy(rng.integers(122, 131, size=(m, k // 32), dtype=np.uint8)).to(DEV)
            sb = torch.from_numpy(rng.integers(122, 131, size=(n, k // 32), dtype=np.uint8)).to(DEV)
            out = q.matmul_mxf8_bf16_tn(a, b, to_blocked(sa.view(e8)), to_blocked(sb.view(e8)), al)
            kind = oracle.KIND_MXFP8_TN_A5 if e5 else oracle.KIND_MXFP8_TN
            ref = oracle.gemm_blockscaled(kind, _np(a[ri]), _np(b), oracle.to_blocked(_np(sa[ri])), oracle.to_blocked(_np(sb)), alpha, len(rows), n, k)
            got = oracle.bf16_bits_to_f32(_np(out[ri])).astype(np.float64)
            want = oracle.bf16_bits_to_f32(ref).astype(np.float64)
            assert (np.abs(got - want) <= np.abs(want) / 128.0 + 1e-4 * np.abs(want).max()).all(), (it, m, n, k, e5)


def test_fuzz_round4_kernels(q):
    """[r4] The kernels that only large tensors reach through the product rules, forced through the lab build on small random shapes:
    the persistent NVFP4 kernel (nvf4_variant 42: ragged M / N, any K % 256 == 0, 1-3 tiles per workgroup; sampled rows against the oracle)
    and the wave-owned backward quantizers (bwd_variant 2 / 3: must return the bytes of the product path, which the test above holds to the oracle)."""
    from qutlass_amd.utils import to_blocked

    rng = np.random.default_rng(108 + 1000 * SEED)
    e4 = torch.float8_e4m3fn
    for it in range(6):
        m, n = int(rng.integers(1, 700)), int(rng.integers(1, 90)) * 8
        k = int(rng.integers(2, 8)) * 256
        a, b = _rand_codes(rng, m, k // 2), _rand_codes(rng, n, k // 2)
        sa = torch.from_numpy(rng.integers(0x30, 0x48, size=(m, k // 16), dtype=np.uint8)).to(DEV)
        sb = torch.from_numpy(rng.integers(0x30, 0x48, size=(n, k // 16), dtype=np.uint8)).to(DEV)
        al = float(rng.choice([1.0, 0.5, 0.1875]))
        with lab.forced(nvf4_variant=42):
            out = lab.matmul_nvf4_bf16_tn(a, b, to_blocked(sa.view(e4)), to_blocked(sb.view(e4)), torch.tensor([al], device=DEV))
        rows = sorted(set([0, m - 1, min(m - 1, 255), min(m - 1, 256)] + [int(r) for r in rng.integers(0, m, 12)]))
        ref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, _np(a)[rows], _np(b), oracle.to_blocked(_np(sa)[rows]), oracle.to_blocked(_np(sb)), al, len(rows), n, k)
        assert np.array_equal(_np(out)[rows], ref), ("nvf4 persistent", it, m, n, k, int((_np(out)[rows] != ref).sum()))
    h = _hadamard(32)
    for it in range(8):
        B, N, M = int(rng.integers(1, 3)), int(rng.integers(1, 24)) * 32, int(rng.integers(1, 30)) * 32 if it % 2 else int(rng.integers(1, 9)) * 128
        x = torch.from_numpy(rng.standard_normal((B, N, M)).astype(np.float32) * 25.0).to(torch.bfloat16).to(DEV)
        xq = torch.from_numpy(rng.integers(0, 256, size=(B, N, M // 2), dtype=np.uint8)).to(DEV)
        xs = torch.from_numpy(rng.integers(112, 142, size=(B, N, M // 32), dtype=np.uint8)).to(DEV)
        alpha = torch.tensor([float(np.float32(rng.uniform(0.05, 20.0)))], device=DEV)
        t0, q0 = q.backward_t_bf16(x, h), q.backward_qt_bf16(xq, xs.view(torch.float8_e8m0fnu), h, alpha)
        for v in (2, 3) + ((5, 4) if M % 128 == 0 else ()):   # [r5] 5 = the ring kernel (product for large inputs), 4 = the lab's panel kernel: QT only (T takes its product rule)
            with lab.forced(bwd_variant=v):
                t1, q1 = lab.backward_t_bf16(x, h), lab.backward_qt_bf16(xq, xs, h, alpha)
            for got, want in zip(t1 + q1, t0 + q0):
                assert torch.equal(got.reshape(-1), want.view(torch.uint8).reshape(-1)), ("bwd", v, it, B, N, M)
