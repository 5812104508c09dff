"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/qutlass_amd.h declares,
argument validation of the C entry points (no launches), the Python host mirror of the reference
interface (names, signatures, error behaviour), and the padded-shape helpers."""
import ctypes
import numpy as np
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from qutlass_amd import _lib, build

    build.build()  # hipcc cross-compiles gfx950 without a GPU; no-op when up to date
    return _lib.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "qutlass_amd.h")).read()
    declared = set(re.findall(r"\b(qutlass_amd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 10
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/qutlass_amd.h but not exported"
    from qutlass_amd._lib import SYMBOLS

    assert declared == set(SYMBOLS), "ctypes table and header disagree"


def test_c_abi_rejects_bad_arguments_without_launching(lib):
    from qutlass_amd._lib import QAMD_ERR_INVALID

    dummy = ctypes.c_void_p(0x1000)  # never dereferenced: validation fails first
    err = lambda: lib.qutlass_amd_last_error().decode()
    g = lib.qutlass_amd_matmul_mxf4_bf16_tn
    assert g(None, dummy, dummy, dummy, dummy, dummy, 128, 128, 128, None) == QAMD_ERR_INVALID
    assert "null pointer" in err()
    assert g(dummy, dummy, dummy, dummy, dummy, dummy, 128, 128, 96, None) == QAMD_ERR_INVALID
    assert "multiple of 128" in err()
    assert g(dummy, dummy, dummy, dummy, dummy, dummy, 128, 130, 128, None) == QAMD_ERR_INVALID
    assert "multiple of 8" in err()
    assert lib.qutlass_amd_matmul_nvf4_bf16_tn(dummy, dummy, dummy, dummy, dummy, dummy, 8, 8, 48, None) == QAMD_ERR_INVALID
    nn = lib.qutlass_amd_matmul_mxf8_bf16_nn
    assert lib.qutlass_amd_mxf8_nn_workspace_bytes(4096, 4096) == 4096 * 4096
    assert nn(dummy, dummy, dummy, dummy, dummy, dummy, 128, 128, 128, None, 0, None) == QAMD_ERR_INVALID
    assert "null pointer" in err()
    assert nn(dummy, dummy, dummy, dummy, dummy, dummy, 128, 128, 128, dummy, 100, None) == QAMD_ERR_INVALID
    assert "workspace too small" in err()
    assert nn(dummy, dummy, dummy, dummy, dummy, dummy, 24, 128, 128, dummy, 1 << 20, None) == QAMD_ERR_INVALID
    assert "multiple of 16" in err()
    assert lib.qutlass_amd_backward_t_bf16(dummy, dummy, 1, 48, 64, dummy, dummy, None) == QAMD_ERR_INVALID
    assert "N % 32" in err()
    assert lib.qutlass_amd_backward_qt_bf16(dummy, dummy, dummy, dummy, 1, 64, 48, dummy, dummy, None) == QAMD_ERR_INVALID
    assert lib.qutlass_amd_backward_bf16_square_double_mxfp8(dummy, 96, 128, dummy, dummy, dummy, None) == QAMD_ERR_INVALID
    assert "multiples of 128" in err()
    assert lib.qutlass_amd_mxfp4_transpose_mxfp8(dummy, dummy, 128, 128, dummy, dummy, None) == QAMD_ERR_INVALID
    assert "n % 256" in err()
    q = lib.qutlass_amd_fused_quantize_mx
    assert q(dummy, dummy, 48, 4096, 0, dummy, dummy, None, None) == QAMD_ERR_INVALID
    assert "Unsupported rotation size 48" in err()
    assert q(dummy, dummy, 64, 4096, 0, dummy, dummy, dummy, None) == QAMD_ERR_INVALID  # mask: rot 32 only
    assert q(dummy, dummy, 32, 100, 0, dummy, dummy, None, None) == QAMD_ERR_INVALID
    assert "divisible" in err()
    assert q(dummy, dummy, 32, 4096, 1, dummy, dummy, dummy, None) == QAMD_ERR_INVALID  # mask + abs_max
    assert lib.qutlass_amd_fused_quantize_nv(dummy, dummy, 8, 4096, 1, dummy, dummy, dummy, None) == QAMD_ERR_INVALID
    # [r5] rotation sizes >= 64 stage H with 16-byte vector loads: a rotation matrix that is not 16-byte aligned (an offset view) is rejected
    odd = ctypes.c_void_p(0x1002)
    assert q(dummy, odd, 64, 4096, 1, dummy, dummy, None, None) == QAMD_ERR_INVALID
    assert "16-byte aligned" in err()
    assert lib.qutlass_amd_fused_quantize_nv(dummy, odd, 128, 4096, 1, dummy, dummy, dummy, None) == QAMD_ERR_INVALID
    assert "16-byte aligned" in err()
    # [r5] GEMM operands / output: fetched and stored as 16-byte pieces (the reference's CUTLASS kernels ask for 128-bit alignment too)
    al = ctypes.c_void_p(0x1000)
    for k_ in range(5):
        args = [al] * 6
        args[k_ if k_ < 4 else 5] = odd
        assert g(*args, 128, 128, 128, None) == QAMD_ERR_INVALID and "16-byte aligned" in err(), k_
        assert lib.qutlass_amd_matmul_mxf8_bf16_tn(*args, 128, 128, 128, None) == QAMD_ERR_INVALID and "16-byte aligned" in err(), k_
        assert lib.qutlass_amd_matmul_nvf4_bf16_tn(*args, 128, 128, 128, None) == QAMD_ERR_INVALID and "16-byte aligned" in err(), k_
    assert lib.qutlass_amd_to_blocked(dummy, 0, 4, dummy, None) == QAMD_ERR_INVALID
    assert lib.qutlass_amd_set_option(b"no_such_option", 1) == -1
    # the PRODUCT library has no kernel-selecting state: these keys exist only in the lab build (libqutlass_amd_bench.so)
    for key in (b"gemm_variant", b"nvf4_variant", b"pp_flags", b"splitk_wg", b"quant_wg_per_cu"):
        assert lib.qutlass_amd_set_option(key, 31) == -1, key
    assert b"gfx950" in lib.qutlass_amd_version()


def test_split_k_workspace_plan_is_a_pure_host_function(lib):
    """qutlass_amd_gemm_splitk_workspace_bytes is the contract between a caller that owns the scratch and the launcher:
    fp32 partials [splits][M][N] for small outputs (<= 128 tiles of 64x64) with a long K (>= 32 stages of 128 bytes),
    0 otherwise; the *_ws entries validate like the plain ones."""
    from qutlass_amd._lib import QAMD_ERR_INVALID

    ws = lib.qutlass_amd_gemm_splitk_workspace_bytes
    def splits(ebits, m, n, k):   # the documented rule
        t, kt = -(-m // 64) * -(-n // 64), -(-(k * ebits // 8) // 128)
        if t >= 256 or kt < 32:
            return 0
        s = min(8, 256 // t, kt // 8)
        return -(-kt // -(-kt // s)) if s >= 2 else 0

    def layout(m, n, s):   # [s][m][n] fp32 partials
        return s * m * n * 4

    # Llama-3-8B down-proj, batch 64: 64 tiles, KT = 56 stages -> 4 splits of 14 stages (one workgroup per CU).  [r6] MXFP4 shapes whose 32x32 tiles fill the chip one
    # per CU go to the in-workgroup K-split kernel instead (capi.hip ks_plan: no scratch), and MXFP8 ones likewise to the wave-owned kernel (os8_plan) -- the MXFP8 shape
    # below fills only half the chip with 32x16 tiles against 56 stages and still shows the split plan
    assert splits(4, 64, 4096, 14336) == 4 and ws(4, 64, 4096, 14336) == 0 and ws(8, 64, 4096, 7168) == 0
    assert splits(8, 64, 1024, 7168) == 7 and ws(8, 64, 1024, 7168) == layout(64, 1024, 7) == 7 * 64 * 1024 * 4
    assert splits(4, 16, 4096, 14336) == 4 and ws(4, 16, 4096, 14336) == 0   # [r6] the wave-owned-ring form of the one-shot kernel takes it (capi.hip os_plan): no scratch
    assert ws(4, 4, 2048, 14336) == layout(4, 2048, splits(4, 4, 2048, 14336))    # ... a quarter of the chip against 56 stages stays with the split plans
    assert ws(4, 128, 4096, 14336) == 0 and ws(4, 128, 4096, 28672) == layout(128, 4096, 4)   # [r6] 64x32 tiles on wave-owned rings up to 64 stages (os64_plan), split plans beyond
    assert ws(4, 256, 4096, 14336) == 0 and ws(4, 192, 4096, 14336) == 0   # more than 128 tiles: no split
    assert ws(4, 64, 4096, 4096) == 0            # 16 stages: too short to pay for the reduction
    assert ws(4, 64, 4096, 8192) == 0 and ws(4, 16, 4096, 8192) == 0 and ws(8, 16, 4096, 4096) == 0 and ws(8, 64, 4096, 4096) == 0 and ws(8, 64, 1024, 8192) == layout(64, 1024, 8)   # 32 stages (fp8: K = 4096); [r6] fp4: ks_plan, fp8: os8_plan up to M = 128
    assert ws(4, 4096, 4096, 4096) == 0 and ws(4, 0, 4096, 4096) == 0 and ws(5, 64, 4096, 14336) == 0
    for m, n, k in [(8, 512, 28672), (40, 1032, 14464), (1, 64, 12288)]:
        b = ws(4, m, n, k)
        assert b == layout(m, n, splits(4, m, n, k)) and 2 <= splits(4, m, n, k) <= 8, (m, n, k, b)
    # [r3] where that rule leaves a long K to too few, too small tiles, the fitted model of capi.hip (plan_small) corrects it -- larger tiles, more K ranges
    # (measured: profiles/calib_mx_small_r3.txt): 96 x 5120 x 25600 ran 160 unsplit 64x64 tiles (27.5 us), now 40 tiles of 128x128 in 4 ranges (19.5 us)
    assert splits(4, 96, 5120, 25600) == 0 and ws(4, 96, 5120, 25600) == layout(96, 5120, 4)
    assert splits(8, 192, 4096, 14336) == 0 and ws(8, 192, 4096, 14336) == layout(192, 4096, 4)        # MXFP8: 30.9 -> 23.0 us
    assert splits(4, 128, 2048, 57344) == 4 and ws(4, 128, 2048, 57344) == layout(128, 2048, 8)       # 32 tiles of 64x128 x 8 instead of 64 of 64x64 x 4
    dummy = ctypes.c_void_p(0x1000)
    g = lib.qutlass_amd_matmul_mxf4_bf16_tn_ws
    assert g(dummy, dummy, dummy, dummy, dummy, dummy, 128, 128, 96, None, 0, None) == QAMD_ERR_INVALID
    assert "multiple of 128" in lib.qutlass_amd_last_error().decode()
    assert lib.qutlass_amd_matmul_mxf8_bf16_tn_ws(None, dummy, dummy, dummy, dummy, dummy, 128, 128, 128, None, 0, None) == QAMD_ERR_INVALID


def test_nvf4_tile_rule(lib):
    """matmul_nvf4_bf16_tn's tile choice (gemm_nvf4.hip.h: nvf4_plan) through the debug entry: -1 skinny split-K, [r6] -2 wave-owned small-batch kernel, 0 256x256, 1 128x128,
    2 128x64, 3 64x64, 4 256x128 on four waves, + 256 x K ranges when the caller brings a workspace and the shape splits.  [r3] Two fitted
    cost models pick (large outputs: full rounds x the tile's time + the last round priced by its fill; small ones: per-tile time by
    workgroups per CU, K stages per workgroup and the reduce pass); the expectations below are the measured winners of
    profiles/calib_tiles_r3.txt and profiles/calib_nv_small_r3.txt (256 CUs): 2560 x 4096 has 160 tiles of 256x256 in one round (84.5 us)
    against 106 / 88 for the finer grids; 3072 x 6144 has 288 big tiles = two rounds (175 us) against 148 on 128x128; 1024 x 4096 (128 tiles
    of 256x128, half the CUs idle: 52 us) stays on 128x128 (36); 256 x 4096 x 14336 runs 64 tiles of 128x128 in 4 K ranges (39.2 us) instead
    of 256 tiles of 64x64 (54.7)."""
    import ctypes

    from qutlass_amd._lib import QAMD_ERR_INVALID

    f = lib.qutlass_amd_debug_nvf4_plan   # debug entry, deliberately not in the public header
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_int64] * 3 + [ctypes.c_int]
    K = 4096
    for ws in (0, 1):   # large outputs never split
        assert f(8192, 8192, 8192, ws) == 0 and f(4096, 4096, K, ws) == 0 and f(4096, 14336, K, ws) == 0
        assert f(2048, 4096, K, ws) == 4 and f(1536, 4096, K, ws) == 4 and f(1024, 6144, K, ws) == 4        # 256 / 192 / 192 tiles of 256x128: one round
        assert f(2560, 4096, K, ws) == 0 and f(3072, 4096, K, ws) == 0 and f(2048, 6144, K, ws) == 0       # 160 / 192 / 192 tiles of 256x256: one round
        assert f(1024, 4096, K, ws) == 1 and f(512, 6144, K, ws) == 1                                       # the 256x128 grid would leave half the chip idle
        # [r4] 288 / 320 / 320 big tiles = 1.1 - 1.25 rounds: the per-tile 256x256 kernel lost these to 128x128 tiles (two rounds at 56 / 63 %); the persistent kernel
        # walks them in balanced rounds (144 / 160 workgroups x 2 tiles) and wins: 137.6 vs 146.2, 183.3 vs 193.5, 146.3 vs 159.7 us (profiles/calib_tiles_nvf4_r4b.txt)
        assert f(3072, 6144, K, ws) == 0 and f(4096, 5120, 5120, ws) == 0 and f(5120, 4096, K, ws) == 0
        assert f(3072, 6144, K + 128, ws) == 1                                                              # K % 256 != 0: the per-tile kernels, the round-3 choice
        assert f(6144, 4096, K, ws) == 0 and f(4096, 6144, K, ws) == 0                                      # 384 big tiles = 1.5 rounds: 192 workgroups x 2 tiles (161 us; 128x128: 190)
        assert f(4096, 5120, 512, ws) == 0 and f(8192, 4096, 14336, ws) == 0
        # small outputs at K = 4096 (16 stages): nothing to split
        # [r6] -2 = the wave-owned small-batch kernel (csrc/gemm_nvf4_os.hip.h): M <= 128 with K <= 4096 up to three rounds of 32x32 tiles (rounds 3-5: the skinny kernel)
        # [r6] -4 = its decode form (16x16 tiles on the 16x16x32 MFMA) wherever those fit one per CU, and two per CU for M <= 16 with K <= 4096
        assert f(512, 4096, K, ws) == 2 and f(256, 4096, K, ws) == 3 and f(64, 4096, K, ws) == -2 and f(1, 4096, K, ws) == -4 and f(96, 4096, K, ws) == -10
        assert f(16, 4096, K, ws) == -4 and f(17, 4096, K, ws) == -5 and f(32, 4096, K, ws) == -5 and f(33, 4096, K, ws) == -2 and f(32, 6144, K, ws) == -2 and f(64, 1024, K, ws) == -4 and f(128, 512, K, ws) == -4 and f(16, 8192, K, ws) == -6 and f(16, 8208, K, ws) == -7
        # [r6] -6 ... -9: the decode form with 32 / 48 / 56 columns per workgroup (-9: 56 columns, A rows 0 ... 7 only) for M <= 16 against wider weights, K <= 8192
        assert f(1, 6144, K, ws) == -6 and f(16, 12288, K, ws) == -7 and f(8, 14336, K, ws) == -9 and f(9, 14336, K, ws) == -8 and f(8, 14336, 8192, ws) == -8 and f(17, 12288, K, ws) not in (-6, -7, -8, -9)
        assert f(8, 16384, K, ws) not in (-6, -7, -8, -9) and f(8, 14336, 14336, ws) not in (-6, -7, -8, -9)
        assert f(16, 28672, K, ws) == -7 and f(1, 24576, K, ws) == -7 and f(16, 28672, 8192, ws) != -7 and f(17, 28672, K, ws) != -7   # two / three rounds of 48-column workgroups, K <= 4096
        # [r6] -10: the 32x32-MFMA kernel on 64x32 tiles (one round instead of two), priced in nvf4_plan
        assert f(128, 4096, K, ws) == -10 and f(128, 6144, K, ws) == -2 and f(128, 8192, K, ws) != -2 and f(32, 14336, K, ws) == -2 and f(64, 14336, K, ws) != -2   # 512 / 768 tiles yes, 1024 / 896 no
        assert f(32, 28672, K, ws) == -2 and f(64, 28672, K, ws) == 3       # 64 rows against a wide weight: 448 tiles of 64x64 (26.6 us) beat the skinny kernel (36.7)
        assert f(512, 5120, 5120, ws) == 1 and f(384, 5120, 5120, ws) == 2  # 160 tiles of 128x128 (41.1 us) against 320 of 128x64 (47.0); 240 of 128x64 fit one per CU
        assert f(0, 4096, K, ws) == -2 and f(4096, 4096, 0, ws) == -2
    # long K, few tiles: K ranges (only with a workspace)
    assert f(256, 4096, 14336, 0) == -10 and f(256, 4096, 14336, 1) == 1 + 256 * 4      # 64 tiles of 128x128 x 4 ranges of 14 stages
    assert f(128, 4096, 14336, 0) == -10 and f(128, 4096, 14336, 1) == -10 and f(128, 2048, 28672, 0) == -1 and f(128, 2048, 28672, 1) == 2 + 256 * 8     # 32 tiles x 7 ranges of 8 stages (8 asked for; 56 stages)
    # [r6] long K on wave-owned rings: one 32x32 tile per CU at most (two rounds up to 32 stages), at least a quarter of the CUs busy, with or without scratch
    for ws in (0, 1):
        assert f(64, 4096, 14336, ws) == -2 and f(16, 4096, 8192, ws) == -4 and f(128, 4096, 8192, ws) == -10 and f(32, 8192, 8192, ws) == -2 and f(32, 4096, 8192, ws) == -5
        assert f(16, 1024, 14336, ws) == -4 and f(64, 8192, 14336, ws) != -2 and f(16, 4096, 28672, ws) == -4 and f(32, 4096, 28672, ws) == -5 and f(64, 4096, 28672, ws) not in (-2, -4, -5) and f(16, 8192, 8192, ws) == -6
    assert f(768, 4096, 14336, 0) == 1 and f(768, 4096, 14336, 1) == 1 + 256 * 4      # 192 tiles: 768 workgroups balance better than 192 (94.7 vs 108.3 us)
    assert f(1024, 5120, 25600, 0) == 4 and f(1024, 5120, 25600, 1) == 1 + 256 * 4    # 320 tiles of 128x128 x 4 (252 us) against 160 of 256x128 (300)
    assert f(64, 8192, 28672, 0) == -1 and f(64, 8192, 28672, 1) == 3 + 256 * 4       # 128 tiles of 64x64 x 4 (49.9 us) against the skinny kernel (69.5)
    assert f(1024, 4096, 14336, 1) == 1 and f(256, 2048, 2048, 1) == -10 and f(384, 2048, 2048, 1) == -11 and f(512, 2048, 2048, 1) == 3   # a full round already / K too short (8 stages): the skinny
    # kernel up to 256 rows where its 32x32 workgroups still fit two per CU (GPU-only timing: 7.4 us against 9.6 on 64x64 tiles), 64x64 tiles beyond
    assert f(192, 4096, 4096, 1) == -11 and f(256, 4096, 4096, 1) == 3 and f(96, 6144, 4096, 1) == -11 and f(160, 6144, 4096, 1) == -2   # [r6] (-11: 96x32 tiles in one round, 10.9-11.1 us; before it three rounds of the wave-owned 32x32 kernel: 13.9; rounds 3-5: skinny, 16.8)
    # the workspace query describes the same plan: ranges x M x N fp32
    g = lib.qutlass_amd_nvf4_splitk_workspace_bytes
    assert g(256, 4096, 14336) == 4 * 256 * 4096 * 4 and g(128, 4096, 14336) == 0 and g(128, 2048, 28672) == 8 * 128 * 2048 * 4 and g(200, 4104, 14368) == 8 * 200 * 4104 * 4
    assert g(4096, 4096, 4096) == 0 and g(256, 4096, 4096) == 0 and g(16, 4096, 14336) == 0 and g(64, 4096, 14336) == 0 and g(0, 4096, 4096) == 0 and g(256, 4096, 0) == 0
    dummy = ctypes.c_void_p(0x1000)
    h = lib.qutlass_amd_matmul_nvf4_bf16_tn_ws
    assert h(dummy, dummy, dummy, dummy, dummy, dummy, 128, 128, 128, None, 64, None) == QAMD_ERR_INVALID and "invalid workspace" in lib.qutlass_amd_last_error().decode()
    assert h(dummy, dummy, dummy, dummy, dummy, dummy, 128, 128, 48, None, 0, None) == QAMD_ERR_INVALID and "multiple of 32" in lib.qutlass_amd_last_error().decode()


def test_split_plans_are_consistent_over_a_shape_grid(lib):
    """Invariants of the two split planners over 5 000 shapes (no GPU): a plan without caller scratch never splits; the workspace queries return
    ranges x M x N fp32 with 2 ... 8 ranges or 0; the launch plan the dry-run hook records uses exactly the ranges the query sized the scratch for."""
    import ctypes

    nv = lib.qutlass_amd_debug_nvf4_plan
    nv.restype, nv.argtypes = ctypes.c_int, [ctypes.c_int64] * 3 + [ctypes.c_int]
    nv_ws, mx_ws = lib.qutlass_amd_nvf4_splitk_workspace_bytes, lib.qutlass_amd_gemm_splitk_workspace_bytes
    dry = lib.qutlass_amd_debug_gemm_plan
    dry.restype = ctypes.c_int
    dry.argtypes = [ctypes.c_int] + [ctypes.c_int64] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    out = (ctypes.c_int * 24)()
    rng = __import__("numpy").random.default_rng(11)
    for _ in range(5000):
        m = int(rng.choice([1, 7, 33, 64, 96, 128, 200, 256, 384, 512, 1000, 2048]))
        n = int(rng.integers(1, 2048)) * 8
        k = int(rng.integers(1, 256)) * 128
        r0, r1 = nv(m, n, k, 0), nv(m, n, k, 1)
        assert -11 <= r0 <= 4 and (r1 < 256 or (1 <= r1 % 256 <= 3 and 2 <= r1 // 256 <= 8)), (m, n, k, r0, r1)
        b = nv_ws(m, n, k)
        assert b == (r1 // 256 if r1 >= 256 else 0) * m * n * 4, (m, n, k, r1, b)   # (256x256 tiles: the persistent kernel's balanced rounds need no scratch)
        for ebits in (4, 8):
            b = mx_ws(ebits, m, n, k)
            assert b % (m * n * 4) == 0 and b // (m * n * 4) in (0, 2, 3, 4, 5, 6, 7, 8), (ebits, m, n, k, b)
            assert dry(ebits, m, n, k, 0, out, 8) >= 1 and out[2] == 1 and out[0] != 89      # no scratch: one pass
            assert dry(ebits, m, n, k, 1 << 40, out, 8) >= 1 and out[0] != 89 and out[2] == max(1, b // (m * n * 4)), (ebits, m, n, k, out[0], out[2], b)   # (stream-K: lab only)


def test_nvf4_persistent_walk_covers_every_stage_once(lib):
    """[r4] gemm_nvf4_pk.hip.h: the unit walk of the persistent NVFP4 kernel, replayed on the CPU through the debug entry (the device kernel runs the
    same NvPkWalk): every K stage of every tile is computed by exactly one unit; a cut tile has exactly one parking unit (its LAST stages) and one
    adding unit (its FIRST stages) that name the same scratch slot, the parking workgroup is the next one in walk order and parks before any of its
    whole tiles; no unit is shorter than two stages; balanced-round plans hold whole tiles only.  (Reference counterpart: the CUTLASS tile
    scheduler behind qutlass/csrc/gemm.cu:73-75.)"""
    import ctypes

    units = lib.qutlass_amd_debug_nvf4_pk_units
    units.restype, units.argtypes = ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    pk = lib.qutlass_amd_debug_nvf4_pk_plan
    pk.restype, pk.argtypes = ctypes.c_int, [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    out3 = (ctypes.c_int * 3)()
    buf = (ctypes.c_int * (5 * 64))()

    def walk_all(grid, T, sk, KT):
        seen = {}
        parks, adds = {}, {}
        for w in range(grid):
            n = units(w, grid, T, sk, KT, buf, 64)
            assert 1 <= n <= 64, (grid, T, sk, KT, w, n)
            us = [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]
            first_whole = next((i for i, u in enumerate(us) if u[3] == 0 and u[0] >= T - sk), len(us))
            for i, (tile, kb, ke, mode, slot) in enumerate(us):
                assert 0 <= tile < T and 0 <= kb < ke <= KT and ke - kb >= 2 and mode in (0, 1, 2)
                assert (mode == 0) == (kb == 0 and ke == KT) and (mode != 1 or ke == KT) and (mode != 2 or kb == 0)
                for k in range(kb, ke):
                    assert (tile, k) not in seen, ("stage computed twice", grid, T, sk, KT, tile, k)
                    seen[(tile, k)] = w
                if mode == 1:
                    assert slot == w and tile not in parks and i <= first_whole
                    parks[tile] = (w, slot, kb)
                if mode == 2:
                    assert slot == w + 1 and tile not in adds and i == n - 1
                    adds[tile] = (w, slot, ke)
        assert len(seen) == T * KT, ("stages missing", grid, T, sk, KT, len(seen))
        assert parks.keys() == adds.keys()
        for t in parks:
            assert parks[t][1] == adds[t][1] and parks[t][0] == adds[t][0] + 1 and parks[t][2] == adds[t][2]
        return len(parks)

    # the plan of real shapes on 256 CUs: Llama / Qwen weights at the batch sizes the dip scan flags
    for (m, n, k) in [(6144, 4096, 4096), (4096, 5120, 5120), (8192, 8192, 8192), (4096, 28672, 4096), (3072, 28672, 8192), (4096, 14336, 4096), (5120, 4096, 4096)]:
        for may_sk in (0, 1):
            if not pk(m, n, k, may_sk, out3):
                continue
            grid, sk, KT = out3[0], out3[1], out3[2]
            T = ((m + 255) // 256) * ((n + 255) // 256)
            assert grid <= 256 and (sk == 0 or (may_sk and grid == 256 and sk == 256 + T % 256 and T > 256))
            cuts = walk_all(grid, T, sk, KT)
            assert (cuts > 0) == (sk > 0) or sk == 0
    assert pk(6144, 4096, 4096, 1, out3) and (out3[0], out3[1]) == (256, 384)      # 384 tiles = 1.5 rounds: every workgroup walks 24 of the 6144 stages
    assert pk(6144, 4096, 4096, 0, out3) and (out3[0], out3[1]) == (192, 0)        # no scratch: 192 workgroups x 2 whole tiles
    assert pk(8192, 8192, 8192, 1, out3) and (out3[0], out3[1]) == (256, 0)        # 1024 tiles = 4 full rounds: nothing to cut
    assert not pk(4096, 4096, 4096 + 128, 1, out3) and not pk(4096, 4096, 256, 1, out3)   # K % 256 / K < 512: the per-tile kernels
    # synthetic sweeps: every remainder, short and long K
    for KT in (2, 3, 5, 16, 33):
        for grid in (8, 24, 256):
            for T in list(range(grid + 1, 2 * grid + 1, max(1, grid // 8))) + [3 * grid + 5, 7 * grid - 1]:
                walk_all(grid, T, grid + T % grid if T % grid else 0, KT)
                walk_all(grid, T, 0, KT)
    # the MX kernels' walk (gemm_mx_deepp.hip.h): range boundaries on EVEN stages (the stage code is unrolled by LDS-buffer parity)
    sk = lib.qutlass_amd_debug_sk_units
    sk.restype, sk.argtypes = ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]

    def walk_mx(grid, T, sktiles, KT):
        seen, parks, adds = {}, {}, {}
        for w in range(grid):
            n = sk(w, grid, T, sktiles, KT, 2, buf, 64)
            assert 0 <= n <= 64
            us = [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]
            for i, (tile, kb, ke, mode, slot) in enumerate(us):
                assert kb % 2 == 0 and ke % 2 == 0 and ke - kb >= 2 and 0 <= kb < ke <= KT
                for k in range(kb, ke):
                    assert (tile, k) not in seen
                    seen[(tile, k)] = w
                if mode == 1:
                    assert slot == w and i == 0 or us[i - 1][0] < T - sktiles      # parked before anything else of the stream
                    parks[tile] = (w, kb)
                if mode == 2:
                    assert slot == w + 1 and i == n - 1
                    adds[tile] = (w, ke)
        assert len(seen) == T * KT and parks.keys() == adds.keys()
        for t in parks:
            assert parks[t][0] == adds[t][0] + 1 and parks[t][1] == adds[t][1]
        return len(parks)

    # (the planner -- capi.hip sk_tiles_for -- asks for KT >= 8 and MORE than one round of tiles: every range is then at least one tile long -- a tile is
    #  cut at most once, by two neighbouring workgroups -- and none is empty, which the slot pairing "parked by the next workgroup" relies on)
    for KT in (8, 16, 32, 224):
        for (grid, T) in [(256, 384), (256, 320), (256, 257), (256, 511), (256, 576), (256, 1344), (8, 9), (8, 12), (24, 25), (24, 47)]:
            cuts = walk_mx(grid, T, grid + T % grid, KT)
            assert cuts > 0


def test_plan_model_constants_reproduce_from_the_committed_calibration():
    """The constants of the two fitted dispatch models (gemm_nvf4.hip.h: nvf4_plan, capi.hip: plan_small) are what tools/fit_plan_models.py finds in the
    committed calibration data (profiles/calib_nv_small_r3.txt, calib_nv_small_r3_graph.txt, calib_mx_small_r3.txt) -- provenance of the numbers, no GPU."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fit_plan_models", os.path.join(ROOT, "tools", "fit_plan_models.py"))
    fit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fit)
    nv_src = open(os.path.join(ROOT, "qutlass_amd", "csrc", "gemm_nvf4.hip.h")).read()
    mx_src = open(os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")).read()
    arr = lambda src, name: [float(v) for v in re.search(r"\b%s\[3\] = \{([^}]*)\}" % name, src).group(1).split(",")]
    close = lambda got, want, tol=0.012: all(abs(g - w) <= tol * max(abs(w), 1.0) for g, w in zip(got, want))
    a, b, e, r0, bw, rms = fit.fit_nv()
    body = nv_src[nv_src.index("inline NvPlan nvf4_plan"):]
    assert close(arr(body, "A"), a) and close(arr(body, "B"), b) and close(arr(body, "E"), e) and rms < 0.08
    assert "2.897" in body and "4.409e6" in body and abs(r0 - 2.897) < 0.01 and abs(bw - 4.409) < 0.01
    s0, s1, rms = fit.fit_nv_skinny()
    assert "2.53 + 4.66" in body and abs(s0 - 2.53) < 0.01 and abs(s1 - 4.66) < 0.01 and rms < 0.07
    mx = fit.fit_mx()
    for fmt, tag in (("mxf4", "4"), ("mxf8", "8")):
        a, b, e, r0, bw, rms = mx[fmt]
        assert close(arr(mx_src, "A" + tag), a) and close(arr(mx_src, "B" + tag), b) and close(arr(mx_src, "E" + tag), e), fmt
    assert "r0 = EBITS == 4 ? 1.39 : 3.86, bw = EBITS == 4 ? 2.63e6 : 2.79e6" in mx_src
    assert abs(mx["mxf4"][3] - 1.39) < 0.01 and abs(mx["mxf8"][3] - 3.86) < 0.01 and abs(mx["mxf4"][4] - 2.63) < 0.01 and abs(mx["mxf8"][4] - 2.79) < 0.01


def test_nvf4_large_output_rule_against_the_committed_calibration(lib):
    """The large-output model of nvf4_plan (full rounds x the tile's time + the last round priced by its fill), evaluated through the real C++ on the forced-variant
    calibration of 66 shapes (profiles/calib_tiles_r3.txt: microseconds under 256x256 / 256x128 / 128x128 tiles): the chosen tiles sum to within 0.5 % of the best
    measured tile of every shape, and no shape is more than 4 % off."""
    import ctypes

    f = lib.qutlass_amd_debug_nvf4_plan
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_int64] * 3 + [ctypes.c_int]
    col = {0: 1, 4: 2, 1: 3}   # cfg -> column of "auto 256 256x128 128"
    chosen = best = 0.0
    worst = 1.0
    n = 0
    for line in open(os.path.join(ROOT, "profiles", "calib_tiles_nvf4_r4b.txt")):   # [r4] columns: auto, persistent 256x256, 256x128, 128x128
        if not line.startswith("nvf4"):
            continue
        head, vals = line.split("|")
        m, nn, k = (int(v) for v in head.split()[1:4])
        t = [float(v) for v in vals.split()]
        cfg = f(m, nn, k, 0)
        if cfg not in col:
            continue            # below the large-output regime (priced by the small-output model)
        chosen += t[col[cfg]]
        best += min(t[1:])
        worst = max(worst, t[col[cfg]] / min(t[1:]))
        n += 1
    assert n >= 50 and chosen <= 1.005 * best and worst <= 1.04, (n, chosen, best, worst)


def test_mx_large_output_rules_against_the_committed_calibration(lib):
    """The MX GEMMs' dispatch for half-chip and larger outputs (persistent 256x256 tile / heterogeneous launch / 128x128 / 256x128 / 128x128 ring), evaluated through
    the dry-run hook on the forced-variant calibration (profiles/calib_tiles_r3.txt, 66 shapes per format): the planned variants sum to within 2 % of the best
    measured candidate of every shape (MXFP4 5912 against 5840 us, MXFP8 8555 against 8478)."""
    f = lib.qutlass_amd_debug_gemm_plan
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] + [ctypes.c_int64] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    out = (ctypes.c_int * 24)()
    cols = {"mxf4": {90: 1, 98: 2, 24: 3, 58: 4, 73: 5}, "mxf8": {90: 1, 98: 2, 24: 3, 58: 4}}
    tot = {"mxf4": [0.0, 0.0, 0], "mxf8": [0.0, 0.0, 0]}
    for line in open(os.path.join(ROOT, "profiles", "calib_tiles_r3.txt")):
        fmt = line.split()[0] if line.strip() else ""
        if fmt not in cols or line.startswith("#"):
            continue
        head, vals = line.split("|")
        m, n, k = (int(v) for v in head.split()[1:4])
        t = [float(v) for v in vals.split()]
        assert f(4 if fmt == "mxf4" else 8, m, n, k, 0, out, 8) == 1
        c = cols[fmt].get(out[0])
        if c is None:
            continue            # a variant the calibration did not force (64x128 ring for 512-row outputs)
        tot[fmt][0] += t[c]
        tot[fmt][1] += min(t[1:])
        tot[fmt][2] += 1
    for fmt, (chosen, best, cnt) in tot.items():
        assert cnt >= 55 and chosen <= 1.02 * best, (fmt, cnt, chosen, best)


def test_small_output_plans_against_the_gpu_only_calibrations(lib):
    """nvf4_plan and plan_small (with caller scratch) through the real C++, on the GPU-only re-takes of the small-output calibrations (HIP-graph replays; 132 shapes
    per format, M = 16 ... 1024): wherever the planned (tile, K ranges) is one of the measured candidates, the plans sum to within 1.5 % (NVFP4) / 2.5 % (MXFP4, MXFP8)
    of the best measured candidate of each shape."""
    import math

    def rows(path):
        names = None
        for line in open(os.path.join(ROOT, "profiles", path)):
            if line.startswith("#"):
                names = line.split("|")[1].split()
                continue
            head, vals = line.split("|")[:2]
            h = head.split()
            yield h[0], int(h[1]), int(h[2]), int(h[3]), dict(zip(names, (float(v) for v in vals.split())))

    nv = lib.qutlass_amd_debug_nvf4_plan
    nv.restype, nv.argtypes = ctypes.c_int, [ctypes.c_int64] * 3 + [ctypes.c_int]
    name = {-1: "skinny", 1: "128x128", 2: "128x64", 3: "64x64", 4: "256x128"}
    chosen = best = 0.0
    cnt = 0
    for _, m, n, k, d in rows("calib_nv_small_r3_graph.txt"):
        r = nv(m, n, k, 1)
        cfg, s = (r % 256, r // 256) if r >= 256 else (r, 1)
        col = name.get(cfg, "") + ("/%d" % s if s > 1 else "")
        if col in d and not math.isnan(d[col]):
            chosen += d[col]
            best += min(v for v in d.values() if not math.isnan(v))
            cnt += 1
    assert cnt >= 65 and chosen <= 1.015 * best, (cnt, chosen, best)   # ([r6] the shapes the wave-owned small-batch kernels take -- 32x32 / 64x32 tiles in one to four rounds -- have no column in the round-3 calibration)

    dry = lib.qutlass_amd_debug_gemm_plan
    dry.restype = ctypes.c_int
    dry.argtypes = [ctypes.c_int] + [ctypes.c_int64] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    out = (ctypes.c_int * 24)()
    mxname = {60: "skinny", 70: "r64", 72: "r64x128", 73: "r128", 24: "p128", 58: "256x128"}
    tot = {"mxf4": [0.0, 0.0, 0], "mxf8": [0.0, 0.0, 0]}
    for path in ("calib_mx_small_r3_graph_m128.txt", "calib_mx_small_r3_graph_m192_1024.txt"):
        for fmt, m, n, k, d in rows(path):
            assert dry(4 if fmt == "mxf4" else 8, m, n, k, 1 << 40, out, 8) == 1
            col = mxname.get(out[0], "") + ("/%d" % out[2] if out[2] > 1 else "")
            if col in d and not math.isnan(d[col]):
                tot[fmt][0] += d[col]
                tot[fmt][1] += min(v for v in d.values() if not math.isnan(v))
                tot[fmt][2] += 1
    for fmt, (c, b, cnt) in tot.items():
        assert cnt >= 70 and c <= 1.025 * b, (fmt, cnt, c, b)


def test_auto_dispatch_rules_dry_run(lib):
    """The tile / schedule choice of the MX GEMMs (DESIGN.md sections 3.3, 3.7, 3.8) through the library's dry-run hook:
    the real dispatch code runs, launches are recorded instead of issued.  (variant, N of the launch, K splits)."""
    f = lib.qutlass_amd_debug_gemm_plan   # debug entry, deliberately not in the public header
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] + [ctypes.c_int64] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]

    def plan(ebits, m, n, k, ws=0):
        out = (ctypes.c_int * 24)()
        cnt = f(ebits, m, n, k, ws, out, 8)
        return None if cnt < 0 else [(out[3 * i], out[3 * i + 1], out[3 * i + 2]) for i in range(cnt)]

    DEEPP, DEEP, SKINNY, RING64, RING64x128, RING128 = 90, 30, 60, 70, 72, 73
    KS32, KS32x64 = 561, 562   # [r6] in-workgroup K-split kernel (csrc/gemm_mx_ks.hip.h), 32x32 / 32x64 tiles: MXFP4 shapes whose 32x32 tiles fit one per CU (capi.hip ks_plan)
    big = 1 << 30
    # headline and the other BASELINE configs: 256x256 tiles, the persistent deep schedule (fp4 and fp8);
    # C3 = 3.5 rounds of tiles -> one persistent launch with balanced rounds (224 workgroups x 4 tiles); 1.25 rounds (320 tiles) ->
    # the heterogeneous launch: 256 persistent workgroups x 1 tile + the 64 residual tiles as 256 quarter tiles in the same grid;
    # 1.5 rounds (384 tiles) stay with balanced rounds (the cost model of capi.hip: hetero_wins)
    assert plan(4, 4096, 4096, 4096) == [(DEEPP, 4096, 1)]
    assert plan(4, 8192, 8192, 8192) == [(DEEPP, 8192, 1)]
    assert plan(4, 4096, 14336, 4096) == [(DEEPP, 14336, 1)]
    HETERO = 98
    assert plan(4, 4096, 5120, 4096) == [(HETERO, 5120, 1)] and plan(4, 5120, 4096, 4096) == [(HETERO, 4096, 1)]
    assert plan(8, 4096, 5120, 4096) == [(HETERO, 5120, 1)]
    assert plan(4, 3072, 8192, 4096) == [(DEEPP, 8192, 1)]
    # the cost model of capi.hip (hetero_wins), tile counts on both sides of each switch: 272 / 288 / 320 tiles heterogeneous, 352 … 512 balanced,
    # 544 / 576 heterogeneous, 608 … 768 balanced, 832 heterogeneous, 896 (C3) balanced; a (K, M) operand never (matmul_mxf8_bf16_nn keeps balanced rounds)
    for n_cols, want in ((4352, HETERO), (4608, HETERO), (5632, DEEPP), (6144, DEEPP), (7168, DEEPP), (8192, DEEPP), (8704, HETERO), (9216, HETERO), (9728, DEEPP),
                         (11008, DEEPP), (12288, DEEPP), (13312, HETERO), (14336, DEEPP)):
        assert plan(4, 4096, n_cols, 4096) == [(want, n_cols, 1)], n_cols
    assert plan(4, 4096, 5120, 1024) == [(HETERO, 5120, 1)] and plan(4, 4096, 5120, 14336) == [(HETERO, 5120, 1)]   # the rule does not depend on K
    assert plan(8, 4096, 4096, 4096) == [(DEEPP, 4096, 1)]
    assert plan(4, 256, 1 << 22, 128) == [(25, 1 << 22, 1)]          # absurdly wide output: 32-bit tile offsets of the persistent epilogue do not reach
    # decode: LDS-free split-K kernel while the weight has fewer than 128 64-row tiles, ring kernel beyond, 64x128 tiles for huge N
    # [r3] re-measured GPU-only: the split-K kernel up to M = 8 (N <= 8192), up to M = 24 only against small weights (N <= 2048); the ring stays flat in M beyond
    # [r6] ... where the in-workgroup K-split kernel does not take the shape: it does whenever the 32x32 tiles fit one per CU and K <= 24 stages of 256 (or they fill
    # more than half the chip and K <= 16384) -- N = K = 4096: M = 1 ... 64 4.5-5.3 -> 4.1-4.5 us; 32x64 tiles where 32x32 just overflow (N = 14336)
    # [r6] ... and with K <= 4096 (16 stages) the tile's whole K extent fits the LDS: the one-shot kernel (csrc/gemm_mx_os.hip.h), N = K = 4096, M = 1 ... 64 4.05-4.39 -> 3.34-3.67 us
    OS32, OS16 = 568, 569   # 32 / 16 output columns per workgroup: 16 whenever that still leaves at most one workgroup per CU (N = 4096, M <= 32: 256 workgroups)
    OSD = 571               # [r6] its decode form: 16x16 tiles on the 16x16x128 MFMA wherever those fit one per CU (N = 4096: M <= 16)
    assert plan(4, 1, 4096, 4096) == [(OSD, 4096, 1)] and plan(4, 8, 4096, 4096) == [(OSD, 4096, 1)] and plan(4, 8, 8192, 8192) == [(572, 8192, 1)] and plan(4, 17, 8192, 8192) == [(OS32, 8192, 1)]
    assert plan(4, 16, 4096, 4096) == [(OSD, 4096, 1)] and plan(4, 32, 4096, 4096) == [(OS16, 4096, 1)] and plan(4, 24, 2048, 2048) == [(OSD, 2048, 1)]
    assert plan(4, 32, 8192, 4096) == [(OS32, 8192, 1)] and plan(4, 33, 8192, 4096) != [(OS32, 8192, 1)]   # one tile per CU at most
    # longer K: the same kernel on wave-owned rings where the tiles fill a quarter of the chip (M < 8 with more than 32 stages stays on the LDS-free split-K kernel
    # unless the 16-column form applies)
    assert plan(4, 16, 4096, 4224) == [(OSD, 4096, 1)] and plan(4, 16, 4096, 8192) == [(OSD, 4096, 1)] and plan(4, 1, 4096, 8192) == [(OSD, 4096, 1)] and plan(4, 17, 4096, 8192) == [(OS16, 4096, 1)]
    assert plan(4, 1, 8192, 14336) == [(KS32, 8192, 1)] and plan(4, 16, 1024, 5120) == [(KS32, 1024, 1)]   # what the K-split ring kernel keeps
    assert plan(4, 16, 1024, 14336) != [(OS32, 1024, 1)] and plan(4, 8, 4096, 14336, big) == [(OSD, 4096, 1)] and plan(4, 16, 4096, 32768) != [(OS32, 4096, 1)]
    # [r6] M <= 16 against wider weights: the decode form with 32 / 48 / 56 / 64 columns per workgroup (variants 572 ... 575; capi.hip os16_wide_plan) inside its one-shot range
    assert plan(4, 16, 14336, 4096) == [(574, 14336, 1)] and plan(4, 17, 14336, 4096) == [(KS32x64, 14336, 1)] and plan(4, 16, 14336, 8192) != [(574, 14336, 1)]
    assert plan(4, 1, 12288, 4096) == [(573, 12288, 1)] and plan(4, 8, 12288, 5120) == [(573, 12288, 1)] and plan(4, 8, 16384, 4096) == [(575, 16384, 1)] and plan(4, 8, 6144, 4096) == [(572, 6144, 1)]
    assert plan(4, 8, 8192, 2048) == [(OS32, 8192, 1)] and plan(8, 8, 8192, 4096) == [(572, 8192, 1)] and plan(8, 16, 11008, 4096) == [(573, 11008, 1)] and plan(8, 8, 8192, 1024) == [(OS32, 8192, 1)]
    assert plan(4, 1, 4096, 14336) == [(OSD, 4096, 1)] and plan(4, 8, 4096, 11008) == [(OSD, 4096, 1)] and plan(4, 32, 4096, 11008) == [(OS16, 4096, 1)]     # [r6] (rounds 3-5: the LDS-free split-K kernel, 7.6-7.8 us; 16-column wave-owned rings 6.9)
    assert plan(4, 4, 8192, 11008) == [(KS32, 8192, 1)] and plan(4, 4, 2048, 11008) == [(SKINNY, 2048, 1)]     # long K, M < 8, no room for 16-column workgroups / too few tiles: the split-K kernels keep it
    assert plan(4, 24, 11008, 4096) == [(RING64, 11008, 1)] and plan(4, 96, 4096, 4096) == [(RING64, 4096, 1)]  # 32x32 tiles would sit two on a CU
    assert plan(4, 16, 57344, 8192) == [(28, 57344, 1)]
    # [r6] MXFP8 small batches: the wave-owned kernel (csrc/gemm_mx_os.hip.h, EBITS = 8; capi.hip os8_plan) -- 32x16 / 32x32 tiles while they fit one per CU, 64x32 where only
    # those still do; a long K only on (nearly) the whole chip (rounds 1-5: 64x64 ring tiles, N = K = 4096, M = 16: 6.4 -> 5.0 us; 8192 x 4096: 10.6 -> 6.4)
    OS64 = 570
    assert plan(8, 16, 4096, 4096) == [(OSD, 4096, 1)] and plan(8, 1, 4096, 4096) == [(OSD, 4096, 1)] and plan(8, 32, 4096, 4096) == [(OS16, 4096, 1)] and plan(8, 64, 4096, 4096) == [(OS32, 4096, 1)]
    assert plan(8, 32, 8192, 4096) == [(OS32, 8192, 1)] and plan(8, 64, 8192, 4096) == [(OS64, 8192, 1)] and plan(8, 128, 4096, 4096) == [(OS64, 4096, 1)]
    assert plan(8, 128, 2048, 2048) == [(OS32, 2048, 1)] and plan(8, 192, 2048, 2048) == [(RING64, 2048, 1)]       # 64-row tiles only with more than 16 stages
    assert plan(8, 16, 1024, 4096) == [(OSD, 1024, 1)] and plan(8, 16, 4096, 8192) == [(OSD, 4096, 1)] and plan(8, 16, 1024, 8192, big) != [(OS16, 1024, 1)]     # 64 workgroups against 64 stages: the split-K plans
    assert plan(8, 16, 4096, 14336, big) == [(OS16, 4096, 1)] and plan(8, 64, 2048, 14336, big) != [(OS16, 2048, 1)] and plan(8, 16, 4096, 28672, big) == [(OS16, 4096, 1)]
    assert plan(8, 64, 4096, 28672, big) != [(OS32, 4096, 1)] and plan(8, 128, 4096, 28672, big) != [(OS64, 4096, 1)] and plan(8, 16, 14336, 4096) == [(RING64, 14336, 1)]
    # [r6] ... and MXFP4 batches whose 32x32 tiles overflow the chip while 64x32 tiles fit: one shot up to K = 3072, wave-owned rings from ~40 stages (os64_plan)
    assert plan(4, 128, 4096, 2048) == [(OS64, 4096, 1)] and plan(4, 128, 4096, 3072) == [(OS64, 4096, 1)] and plan(4, 128, 4096, 4096) == [(RING64, 4096, 1)]
    assert plan(4, 96, 4096, 11008, big) == [(OS64, 4096, 1)] and plan(4, 128, 4096, 14336) == [(OS64, 4096, 1)] and plan(4, 64, 8192, 8192, big) == [(OS64, 8192, 1)]
    assert plan(4, 128, 4096, 8192, big) == [(RING64, 4096, 1)] and plan(4, 200, 2048, 2048) == [(OS64, 2048, 1)]
    # small outputs: ring schedule; split-K only with caller scratch, <= 128 tiles and >= 32 K stages
    assert plan(4, 64, 4096, 4096) == [(OS32, 4096, 1)] == plan(4, 64, 4096, 4096, big)
    assert plan(4, 64, 4096, 14336) == [(OS32, 4096, 1)] == plan(4, 64, 4096, 14336, big)   # [r6] 256 tiles of 32x32, 56 stages: 11.4 us (4 K ranges + reduce) -> 9.8 (K-split ring) -> 9.1 (wave-owned rings)
    assert plan(8, 64, 1024, 7168) == [(RING64, 1024, 1)]              # (an MXFP8 shape the wave-owned kernel leaves alone: 128 workgroups of 32x16 against 56 stages)
    assert plan(8, 64, 1024, 7168, big) == [(RING64, 1024, 7)]
    assert plan(8, 64, 1024, 7168, 64 * 1024 * 4 * 7 - 1) == [(RING64, 1024, 1)]   # scratch one byte short: single pass
    assert plan(4, 16, 4096, 14336, big) == [(OSD, 4096, 1)]         # [r6] (rounds 3-5: 64x64 tiles x 4 K ranges, 8.85 us; wave-owned rings 8.0)
    assert plan(4, 128, 4096, 28672, big) == [(RING64x128, 4096, 4)]   # (K = 14336: 64x32 tiles on wave-owned rings since round 6, above)
    assert plan(4, 128, 4096, 8192, big) == [(RING64, 4096, 1)]       # a two-way split needs >= 48 K stages to pay for its reduce pass
    assert plan(4, 256, 4096, 14336, big) == [(RING64, 4096, 1)]
    assert plan(8, 64, 1024, 8192, big) == [(RING64, 1024, 8)]
    assert plan(4, 512, 4096, 4096) == [(RING64x128, 4096, 1)]
    assert plan(4, 1024, 4096, 4096) == [(RING128, 4096, 1)] and plan(4, 256, 14336, 4096) == [(RING128, 14336, 1)]
    assert plan(4, 2048, 4096, 4096) == [(24, 4096, 1)]               # 512 tiles of 128x128: two workgroups per CU, pipelined schedule on a 2-deep ring
    # [r3] half-chip outputs with a long K (>= 32 stages of 128 bytes): 256x128 tiles on four waves of 128x64 (3-deep ring), fp4 and fp8, M >= 256
    assert plan(4, 2048, 4096, 8192) == [(58, 4096, 1)] and plan(4, 1024, 8192, 14336) == [(58, 8192, 1)]
    assert plan(8, 2048, 4096, 4096) == [(58, 4096, 1)] and plan(8, 2048, 4096, 2048) == [(24, 4096, 1)] and plan(4, 2048, 4096, 7168) == [(24, 4096, 1)]
    # [r3] ... and whenever the 128x128 grid spills past one tile per CU while the 256x128 grid still fits (fp8 768 x 6144: 288 / 144 tiles, 28.3 -> 23.5 us)
    assert plan(8, 768, 6144, 4096) == [(58, 6144, 1)] and plan(4, 1024, 5120, 25600) == [(58, 5120, 1)] and plan(8, 1024, 4096, 4096) != [(58, 4096, 1)]
    # [r3] more than half a round of 256x256 tiles and >= 8 K stages: the persistent big tile already wins (2560 x 4096 x 4096: 160 tiles, 34.6 -> 28.8 us)
    assert plan(4, 2560, 4096, 4096) == [(DEEPP, 4096, 1)] and plan(4, 3072, 4096, 4096) == [(DEEPP, 4096, 1)] and plan(4, 2560, 4096, 1024) == [(24, 4096, 1)]
    # [r3] the fitted model's corrections of the small-output rule (long K, few tiles: larger ring tiles x more K ranges; only with scratch) and the wide-weight rule
    assert plan(4, 96, 5120, 25600, big) == [(RING128, 5120, 4)] and plan(4, 96, 5120, 25600) == [(RING64, 5120, 1)]
    assert plan(8, 192, 4096, 14336, big) == [(RING128, 4096, 4)] and plan(4, 128, 2048, 57344, big) == [(RING64x128, 2048, 8)]
    assert plan(8, 384, 4096, 14336, big) == [(RING128, 4096, 2)] and plan(8, 384, 4096, 14336) == [(RING64x128, 4096, 1)] and plan(4, 384, 4096, 14336, big) == [(RING64x128, 4096, 1)]
    assert plan(4, 256, 8192, 28672, big) == [(RING128, 8192, 2)] and plan(4, 384, 5120, 25600, big) == [(RING128, 5120, 2)] and plan(4, 256, 8192, 8192, big) == [(RING64x128, 8192, 1)]
    assert plan(4, 96, 57344, 8192) == [(24, 57344, 1)] and plan(8, 128, 51200, 5120) == [(24, 51200, 1)] and plan(4, 192, 57344, 8192) == [(DEEPP, 57344, 1)]
    # an A operand of >= 2 GiB (262400 x 16384 fp4 = 2.15 GB) runs as two row ranges of whole 256-row tiles
    assert plan(4, 262400, 256, 16384) == [(DEEPP, 256, 1), (OS16, 256, 1)]   # 261888 rows, then the last 512 (256 workgroups of 32x16, 64 stages: wave-owned rings; rounds 3-5: 64x64 ring tiles)
    # ... and a B operand of >= 2 GiB (262400 x 16384 fp4 weight) as two column ranges of whole 256-column tiles writing one D
    assert plan(4, 16, 262400, 16384) == [(28, 261888, 1), (SKINNY, 512, 1)]   # [r3] the split-K kernel writes a column range too (row stride ldd)
    assert plan(4, 4096, 262400, 16384) == [(DEEPP, 261888, 1), (71, 512, 1)]
    # rejected arguments never reach the dispatch
    assert plan(4, 128, 128, 96) is None and plan(5, 128, 128, 128) is None


def test_new_round3_entry_points_validate_their_arguments(lib):
    """C-ABI checks of the round-3 entries (no launch: every call is rejected before it reaches the GPU)."""
    from qutlass_amd._lib import QAMD_ERR_INVALID

    dummy = ctypes.c_void_p(0x1000)
    err = lambda: lib.qutlass_amd_last_error().decode()
    qb = lib.qutlass_amd_fused_quantize_mx_blocked
    assert qb(dummy, dummy, 64, 4, 96, 1, dummy, dummy, None, None) == QAMD_ERR_INVALID and "multiple of the rotation size 64" in err()
    assert qb(dummy, dummy, 32, 0, 128, 1, dummy, dummy, None, None) == QAMD_ERR_INVALID and "bad shape" in err()
    assert qb(dummy, dummy, 32, 4, 128, 1, dummy, dummy, dummy, None) == QAMD_ERR_INVALID and "clip mask" in err()   # mask needs quest
    nb = lib.qutlass_amd_fused_quantize_nv_blocked
    assert nb(dummy, dummy, 16, 4, 48, 1, dummy, dummy, dummy, None) == QAMD_ERR_INVALID and "multiple of 32" in err()
    assert nb(dummy, dummy, 16, 4, 64, 1, None, dummy, dummy, None) == QAMD_ERR_INVALID and "null pointer" in err()
    fq = lib.qutlass_amd_fused_quantize_matmul_mxf4_bf16_tn
    assert fq(dummy, dummy, 32, 1, dummy, dummy, dummy, dummy, 33, 128, 128, None) == QAMD_ERR_INVALID and "M must be in 1..32" in err()
    assert fq(dummy, dummy, 64, 1, dummy, dummy, dummy, dummy, 8, 128, 128, None) == QAMD_ERR_INVALID and "Unsupported rotation size 64" in err()
    assert fq(dummy, dummy, 32, 1, dummy, dummy, dummy, dummy, 8, 128, 96, None) == QAMD_ERR_INVALID and "multiple of 128" in err()
    assert fq(dummy, dummy, 32, 7, dummy, dummy, dummy, dummy, 8, 128, 128, None) == QAMD_ERR_INVALID and "invalid method" in err()
    sq = lib.qutlass_amd_backward_bf16_square_double_mxfp8_rows
    assert sq(dummy, 100, 100, 128, dummy, dummy, dummy, None) == QAMD_ERR_INVALID and "multiple of 128 >= m" in err()   # m_pad not a multiple of 128
    assert sq(dummy, 200, 128, 128, dummy, dummy, dummy, None) == QAMD_ERR_INVALID                                        # m_pad < m
    tr = lib.qutlass_amd_mxfp4_transpose_mxfp8_rows
    assert tr(dummy, dummy, 100, 128, 128, dummy, dummy, None) == QAMD_ERR_INVALID and "n % 256" in err()
    assert tr(dummy, dummy, 100, 64, 256, dummy, dummy, None) == QAMD_ERR_INVALID


def test_decode_launch_rule():
    """qutlass_amd._decode_single_launch_wins: the measured one-launch / two-launch rule of fused_quantize_matmul_mxf4_bf16_tn."""
    from qutlass_amd import _decode_single_launch_wins as w

    # [r6] re-taken after the decode forms made the two-launch path's GEMM faster (profiles/calib_actpath_r7.txt): one launch keeps M <= 4 with K <= 4096 and M <= 8 with K <= 2048
    assert w(1, 4096, 4096, 32) and w(4, 6144, 4096, 32) and w(8, 2048, 2048, 32) and w(4, 2048, 2048, 32)
    assert not w(4, 4096, 8192, 32) and not w(8, 6144, 4096, 32) and not w(16, 4096, 4096, 32) and not w(16, 2048, 2048, 32) and not w(1, 4096, 6144, 32)
    assert not w(1, 4096, 14336, 32) and not w(8, 4096, 8192, 32) and not w(16, 4096, 8192, 32) and not w(32, 4096, 4096, 32)
    assert not w(1, 14336, 4096, 32) and not w(1, 4096, 4096, 64) and not w(0, 4096, 4096, 32)


def test_product_kernels_use_no_scratch():
    """a register spill in one of the hand-scheduled kernels is silent and slow; the compiler's own resource report
    (-Rpass-analysis=kernel-resource-usage, device pass of every translation unit) must show 0 bytes of scratch and no VGPR
    spills for every kernel of the product library."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_kres", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kres = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kres)
    ks = kres.collect(lab=False)
    assert len(ks) > 100
    spilled = {k: v for k, v in ks.items() if v.get("scratch", 0) or v.get("vgpr_spill", 0)}
    assert not spilled, spilled


def test_python_surface_matches_reference_signatures():
    import qutlass
    import qutlass_amd
    from qutlass.utils import get_padded_shape_mx, get_padded_shape_nv, pad_to_block, to_blocked  # noqa: F401

    # reference: qutlass/__init__.py:34-203 (parameter names, order, keyword-only markers, defaults)
    want = {
        "matmul_mxf4_bf16_tn": ["a", "b", "a_sf", "b_sf", "alpha", "backend"],
        "matmul_nvf4_bf16_tn": ["a", "b", "a_sf", "b_sf", "alpha", "backend"],
        "matmul_mxf8_bf16_tn": ["a", "b", "block_scale_a", "block_scale_b", "alpha"],
        "matmul_mxf8_bf16_nn": ["a", "b", "block_scale_a", "block_scale_b", "alpha"],
        "fusedQuantizeMx": ["a", "b", "method", "return_mask"],
        "fusedQuantizeNv": ["a", "b", "global_scale", "method"],
    }
    for name, params in want.items():
        f = getattr(qutlass_amd, name)
        assert getattr(qutlass, name) is f
        assert list(inspect.signature(f).parameters) == params, name
    sig = inspect.signature(qutlass_amd.fusedQuantizeMx)
    assert sig.parameters["method"].default == "quest" and sig.parameters["method"].kind is inspect.Parameter.KEYWORD_ONLY
    assert sig.parameters["return_mask"].default is False
    assert inspect.signature(qutlass_amd.fusedQuantizeNv).parameters["method"].default == "abs_max"
    assert inspect.signature(qutlass_amd.matmul_mxf4_bf16_tn).parameters["backend"].default == "cutlass"
    assert list(inspect.signature(to_blocked).parameters) == ["input_matrix", "use_triton_kernel"]
    # QAT-backward wrappers (qutlass/__init__.py:206-315)
    assert list(inspect.signature(qutlass_amd.backward_t_bf16).parameters) == ["x", "h", "xh_e2m1", "xh_e8m0"]
    assert list(inspect.signature(qutlass_amd.backward_qt_bf16).parameters) == ["x_e2m1", "x_e8m0", "h", "alpha", "xh_e2m1", "xh_e8m0"]
    assert list(inspect.signature(qutlass_amd.backward_bf16_square_double_mxfp8).parameters) == ["x_bf16"]
    assert list(inspect.signature(qutlass_amd.mxfp4_transpose_mxfp8).parameters) == ["x_fp4", "scales"]


# exact schema strings of the reference's op library (qutlass/csrc/bindings.cpp:499-513)
REFERENCE_SCHEMAS = {
    "matmul_mxf4_bf16_tn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "matmul_nvf4_bf16_tn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "matmul_ada_mxf4_bf16_tn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "matmul_mxf8_bf16_tn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "matmul_mxf8_bf16_nn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "fusedQuantizeMxQuest": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf) -> (Tensor, Tensor)",
    "fusedQuantizeMxAbsMax": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf) -> (Tensor, Tensor)",
    "fusedQuantizeMxQuestWithMask": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor OUT_mask) -> (Tensor, Tensor, Tensor)",
    "fusedQuantizeNvQuest": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor global_scale) -> (Tensor, Tensor)",
    "fusedQuantizeNvAbsMax": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor global_scale) -> (Tensor, Tensor)",
    "backward_t_bf16": "(Tensor x, Tensor h, Tensor xh_e2m1, Tensor xh_e8m0) -> ()",
    "backward_qt_bf16": "(Tensor x_e2m1, Tensor x_e8m0, Tensor h, Tensor alpha, Tensor xh_e2m1, Tensor xh_e8m0) -> ()",
    "backward_bf16_square_double_mxfp8": "(Tensor x_bf16, Tensor x_fp8, Tensor row_scales, Tensor column_scales) -> ()",
    "mxfp4_transpose_mxfp8": "(Tensor x_fp4, Tensor scales, Tensor x_fp8, Tensor shared_exps) -> ()",
}


def test_torch_ops_registered_with_reference_schemas():
    import qutlass_amd  # noqa: F401

    for name, schema in REFERENCE_SCHEMAS.items():
        op = getattr(torch.ops._qutlass_C, name)
        got = str(op.default._schema)
        assert got == f"_qutlass_C::{name}{schema}", got


def test_ops_are_implemented_by_the_cpp_extension():
    """torch.ops._qutlass_C.* must come from the in-tree C++ extension (qutlass/_CUDA.abi3.so, csrc/torch_ext.cpp), which
    links libqutlass_amd.so -- not from a Python-registered stand-in -- and it must be the reference's module:
    `import qutlass._CUDA` succeeds (PyInit__CUDA, bindings.cpp:537-540) and is the same mapped file."""
    import importlib

    import qutlass_amd  # noqa: F401
    from qutlass_amd import ops

    assert os.path.exists(ops.EXT_PATH) and ops.EXT_PATH.endswith(os.path.join("qutlass", "_CUDA.abi3.so"))
    maps = open("/proc/self/maps").read()
    assert ops.EXT_PATH in maps and "libqutlass_amd.so" in maps
    mod = importlib.import_module("qutlass._CUDA")
    assert os.path.samefile(mod.__file__, ops.EXT_PATH)
    # a C++ kernel registered through the stable ABI is not a Python callable: there is no torch.library python impl
    assert "_qutlass_C::matmul_mxf4_bf16_tn" not in getattr(torch.library, "_impls", {})
    assert torch.ops.qutlass_amd.to_blocked is not None
    # registered for the CUDA dispatch key ONLY, as in the reference (bindings.cpp:516): a CPU tensor never reaches the
    # kernels -- the dispatcher itself refuses
    assert torch._C._dispatch_has_kernel_for_dispatch_key("_qutlass_C::matmul_mxf4_bf16_tn", "CUDA")
    assert not torch._C._dispatch_has_kernel_for_dispatch_key("_qutlass_C::matmul_mxf4_bf16_tn", "CPU")
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops._qutlass_C.matmul_mxf4_bf16_tn(torch.zeros(4, 64, dtype=torch.uint8), torch.zeros(4, 64, dtype=torch.uint8),
                                                 torch.zeros(128, 4, dtype=torch.float8_e8m0fnu), torch.zeros(128, 4, dtype=torch.float8_e8m0fnu), torch.ones(1))


def test_python_level_error_behaviour():
    import qutlass_amd

    a = torch.zeros(4, 64, dtype=torch.bfloat16)
    h = torch.eye(32, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="invalid method"):
        qutlass_amd.fusedQuantizeMx(a, h, method="nope")
    with pytest.raises(ValueError, match="return_mask is only supported"):
        qutlass_amd.fusedQuantizeMx(a, h, method="abs_max", return_mask=True)
    with pytest.raises(ValueError, match="invalid method"):
        qutlass_amd.fusedQuantizeNv(a, h, torch.ones(1), method="nope")
    u8 = torch.zeros(4, 64, dtype=torch.uint8)
    sf = torch.zeros(128, 4, dtype=torch.float8_e8m0fnu)
    with pytest.raises(ValueError, match="invalid backend"):
        qutlass_amd.matmul_mxf4_bf16_tn(u8, u8, sf, sf, torch.ones(1), backend="nope")
    with pytest.raises(ImportError, match="flashinfer"):
        qutlass_amd.matmul_mxf4_bf16_tn(u8, u8, sf, sf, torch.ones(1), backend="flashinfer")
    with pytest.raises(AttributeError):
        qutlass_amd.no_such_function


def test_padded_shapes_and_pad_to_block():
    from qutlass_amd.utils import get_padded_shape_mx, get_padded_shape_nv, pad_to_block

    assert get_padded_shape_mx(torch.empty(4096, 4096)) == (4096, 128)
    assert get_padded_shape_mx(torch.empty(2, 504, 2048)) == (1024, 64)
    assert get_padded_shape_mx(torch.empty(1, 4096)) == (128, 128)
    assert get_padded_shape_mx(torch.empty(3, 96)) == (128, 4)
    assert get_padded_shape_nv(torch.empty(504, 4096)) == (512, 256)
    x = torch.arange(6.0).reshape(2, 3)
    assert pad_to_block(x, [0], 128).shape == (128, 3) and pad_to_block(x, [0, 1], 4).shape == (4, 4)
    assert pad_to_block(x, [0], 128)[:2].equal(x) and pad_to_block(x, [0], 128)[2:].abs().sum() == 0


def test_no_oracle_in_product_path():
    # the product package must never import the test oracle (or any CPU fallback)
    pkg = os.path.join(ROOT, "qutlass_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libqutlass_oracle" not in src, f


def test_in_tree_binaries_are_not_older_than_their_sources():
    """The .so files are git-ignored build products that travel to the GPU box as they are: a library built before the last
    source edit would make every GPU result describe other code than the tree.  (`__graft_entry__.build()` rebuilds stale ones.)"""
    from qutlass_amd import build

    assert not build.needs_build(), "libqutlass_amd.so / qutlass/_CUDA.abi3.so are older than their sources: run __graft_entry__.build()"
    assert not build._stale(build.BENCH_OUT, build._kernel_sources()), "libqutlass_amd_bench.so is older than its sources"


def test_backward_op_kernel_rules(lib):
    """[r4] Which kernel backward_t_bf16 / backward_qt_bf16 / backward_bf16_square_double_mxfp8 launch (capi.hip: bwd_kernel_rule, sq_column_tiles_rule), through
    the dry-run hook (256-CU part): the thresholds of the one-box A/Bs, profiles/ab_bwd_r4m.txt and ab_sq_abl_r4ae.txt."""
    f = lib.qutlass_amd_debug_stream_plan
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    # backward_t (op 0): wave-owned 128-byte lines from 6 units (8 groups x 64 m) per CU on, the round-3 kernel below
    assert f(0, 1, 8192, 8192) == 3 and f(0, 1, 2048, 14336) == 3 and f(0, 1, 4096, 4096) == 1 and f(0, 1, 8192, 1024) == 1 and f(0, 4, 4096, 4096) == 3
    # backward_qt (op 1): wave-owned 64-byte segments from 3 units per CU on; [r5] from 12 units per CU on, with M % 128 == 0, the same kernel fed through the
    # shared whole-line ring (profiles/ab_bwd_r5x_ring_threshold.txt)
    assert f(1, 1, 8192, 8192) == 5 and f(1, 1, 4096, 4096) == 2 and f(1, 1, 2048, 14336) == 2 and f(1, 1, 8192, 1024) == 1 and f(1, 1, 256, 192) == 1
    assert f(1, 1, 14336, 4096) == 5 and f(1, 1, 11008, 4096) == 2 and f(1, 1, 6144, 6144) == 2 and f(1, 1, 8192, 8192 + 64) == 2 and f(1, 2, 8192, 4096) == 5
    # never the other op's kernel (the product library only holds QT x 4 groups and T x 8 groups)
    for shp in [(1, 32, 8), (2, 512, 320), (1, 16384, 16384)]:
        assert f(0, *shp) in (1, 3) and f(1, *shp) in (1, 2, 5)
    # square_double (op 2): 16-wave workgroups of 128 x 512 when n % 512 == 0 and that still gives every CU a workgroup
    assert f(2, 8192, 8192, 0) == 4 and f(2, 4096, 4096, 0) == 4 and f(2, 1024, 4096, 0) == 1 and f(2, 4096, 4224, 0) == 1 and f(2, 16384, 512, 0) == 1 and f(2, 32768, 512, 0) == 4
    assert f(0, 1, 100, 64) == -1 and f(2, 100, 128, 0) == -1 and f(9, 1, 1, 1) == -1


def test_raster_decode_matches_the_divisions(lib):
    """[r4] The persistent kernels turn a tile id into (tile row, tile column) with a host-made reciprocal instead of two integer divisions (common.hip.h
    raster_decode: ~20 scalar instructions instead of ~90, twice before a workgroup's first MFMA).  The same function, run on the host through the debug hook,
    against the plain formula: groups of 4 tile rows walked column by column, the last group with tiles_m % 4 rows -- every tile of small grids, the ends of
    large ones (where the reciprocal's estimate is one too high and the fix-up has to act), and ids past the grid (the kernels decode min(t, T - 1) only, but
    garbage there must still not trap)."""
    f = lib.qutlass_amd_debug_raster_decode
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]

    def ref(t, tm, tn):
        group = 4 * tn
        gid, rem = divmod(t, group)
        gsz = min(tm - 4 * gid, 4)
        return 4 * gid + rem % gsz, rem // gsz

    def check(tm, tn, t0, n):
        out = (ctypes.c_int * (2 * n))()
        assert f(tm, tn, t0, n, out) == n
        got = np.frombuffer(out, dtype=np.int32).reshape(n, 2)
        for i in range(n):
            assert tuple(got[i]) == ref(t0 + i, tm, tn), (tm, tn, t0 + i, tuple(got[i]), ref(t0 + i, tm, tn))

    for tm in range(1, 10):
        for tn in (1, 2, 3, 5, 7, 16, 33):
            check(tm, tn, 0, tm * tn)
    seen = set()
    for tm, tn in [(16, 16), (32, 32), (16, 112), (17, 3), (64, 56), (3, 1000), (1023, 1021), (4099, 4093), (32768, 32768), (46340, 46340), (7, 500000)]:
        T = tm * tn
        assert T < 2**31
        for t0 in {0, max(0, T // 2 - 300), max(0, T - 600)}:
            n = min(600, T - t0)
            check(tm, tn, t0, n)
            seen.add((tm, tn))
    assert len(seen) == 11
    assert f(0, 4, 0, 1, (ctypes.c_int * 2)()) == -1


def test_headline_kernel_prologue_stays_short():
    """[r4] The instructions a persistent workgroup runs ahead of its first MFMA are exposed one for one at 4096^3 (624 -> 555 of them were worth 1.1 %, DESIGN.md 7;
    profiles/ab_lib_gemm_r4bc_magic_decode.txt, ab_lib_gemm_r4bh_scalar_store_offsets_early_kernargs.txt).  Read off the ISA of the product build's headline
    kernel (hipcc -S of translation unit 2, ~15 s): the kernel arguments arrive in ONE scalar-load round before the first LDS-DMA, no integer division is left in
    the tile decode, the first MFMA comes within 580 instructions, and the kernel keeps the registers the scalar store offsets freed (236 of 256)."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from isa_prologue import prologue

    r = prologue(2, "gemm_mx_deepp_kernelINS_7GemmCfgILi256ELi256ELi2ELi2ELi4ELb0ELi0ELi2ELi0EEELi17E")
    assert r["vgpr_spills"] == 0
    assert r["waits_before_dma"] == 1, r
    assert r["first_dma"] <= 225 and r["first_mfma"] <= 580, r
    assert r["arch_vgprs"] <= 240, r


def test_no_valu_write_directly_behind_a_scalar_offset_store():
    """[r4] One pattern the compiler does not guard on gfx950, by the evidence of round 4 (tools/store_data_hazard.py has the story): a buffer_store_dwordx3/x4 whose
    soffset is an SGPR, followed ONE instruction later by a VALU write of its data registers.  A schedule that had it (QAMD_DEEPP_RB2) produced output that differed
    from the product's although the two ISAs agree in order, registers, wait counts and read -> store data flow.  The product's closest case has one instruction in
    between (bit-exact over the GPU suite, fuzz seeds 61-68 and 27 whole-output comparisons against the build before it); distance 1 must not appear in any kernel."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "store_data_hazard.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "108 kernels scanned; 0 with" in r.stdout or " 0 with a VALU write" in r.stdout, r.stdout[-500:]


def test_xcd_tile_blocks_fetch_within_six_percent_of_the_whole_tile_optimum(lib):
    """[r5] VERDICT r4 item 3 asked for a tile BLOCK per XCD instead of tile rows, to cut the 3.0 x operand re-fetch.  The map already is one: xcd_remap gives an XCD a
    contiguous run of G / 8 tiles and the grouped raster folds a run of 32 into 4 tile rows x 8 tile columns.  tools/xcd_panel_traffic.py counts the panels each L2 must
    fetch per round: 53.5 MB at 4096^3 (measured FETCH_SIZE: 52.6 - 53.8 MB) against a bound of 50.4 MB for ANY assignment of whole 256 x 256 tiles, 32 per XCD -- the
    3.0 x is the price of eight separate L2s, not of the map.  The tool's raster is the library's (debug hook), and the product map stays within 15 % of the bound."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("xpt", os.path.join(ROOT, "tools", "xcd_panel_traffic.py"))
    xpt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(xpt)
    f = lib.qutlass_amd_debug_raster_decode
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    for tm, tn in [(16, 16), (16, 56), (20, 16), (7, 5)]:
        n = tm * tn
        out = (ctypes.c_int * (2 * n))()
        assert f(tm, tn, 0, n, out) == n
        got = np.frombuffer(out, dtype=np.int32).reshape(n, 2)
        assert all(tuple(got[t]) == xpt.raster(t, tm, tn) for t in range(n))
    import math
    for (M, N, K) in [(4096, 4096, 4096), (4096, 14336, 4096), (8192, 8192, 8192), (5120, 4096, 4096)]:
        fetched, alg, G, rounds = xpt.traffic(M, N, K, xpt.product)
        bound = rounds * 8 * 2 * math.sqrt(G / 8) * 256 * K // 2 * (1 + 1 / 16)
        assert fetched <= 1.15 * bound, (M, N, K, fetched / bound)
        assert xpt.traffic(M, N, K, xpt.plain_rowmajor)[0] >= fetched      # the grouped raster never loses against plain row-major order
    assert abs(xpt.traffic(4096, 4096, 4096, xpt.product)[0] / 1e6 - 53.5) < 0.1


def test_product_build_does_not_see_the_lab_sources():
    """[r5] The experiments on the persistent kernels (stage traces, ablations, stream-K, the retirement variants) live in csrc/lab/ (gemm_mx_deepp_lab.hip.h, gemm_mx_lab.hip.h, ...);
    the product translation units must not even read them, so a lab-only edit cannot change a byte of libqutlass_amd.so.  Checked on the preprocessor's
    own file list (hipcc -E of the product build): none of the lab headers is a dependency -- and the lab build does read them."""
    import subprocess
    from qutlass_amd import build
    src = os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")
    lab_headers = ("gemm_mx_deepp_lab.hip.h", "gemm_mx_lab.hip.h", "quartet_bwd_lab.hip.h", "gemm_mx_duo.hip.h")   # [r6] all under csrc/lab/, which setup.py does not ship
    def deps(unit, extra):   # the files the preprocessor read (its line markers): one unit is enough per build -- every unit includes the same headers
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "tu.i")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", f"-DQAMD_TU={unit}", "--cuda-device-only", "-E", src, "-o", out] + extra,
                           capture_output=True, text=True, check=True)
            return open(out, errors="replace").read()
    for unit in (build.UNITS[0], 2):
        d = deps(unit, [])
        assert "gemm_mx_deepp.hip.h" in d
        for h in lab_headers:
            assert h + '"' not in d, (unit, h)
    d = deps(2, ["-DQAMD_BENCH=1"])
    assert all(h + '"' in d for h in lab_headers)
