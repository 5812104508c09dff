"""CPU: pin the C oracle (oracle/qutlass_oracle.c) against the golden vectors that
tests/golden/make_golden.py produced from the REFERENCE's own Python test oracles.

Pass rules:
  * to_blocked, e8m0 scale bytes, clip masks, GEMM bf16 outputs: bit-exact.
  * packed e2m1 codes: equal modulo the sign of zero (the reference's `_rtne_fp4` encodes an exact
    +0.0 as 0x8, an artefact of torch.bucketize; kernel/PTX semantics give 0x0).
"""
import os

import numpy as np
import pytest

import oracle


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_e2m1_known_answers(golden_dir):
    g = _load(golden_dir, "e2m1_kat.npz")
    x, y = g["x"], g["y"]
    for xi, yi in zip(x, y):
        code = oracle.e2m1_encode(np.float32(xi))
        assert oracle.e2m1_decode(code) == yi, (xi, yi, code)
        assert (code >> 3) == int(np.signbit(xi)), (xi, code)  # sign preserved, also for -0 / tiny
    # packing order: element 2j -> low nibble (SURVEY 8a KATs)
    pk = lambda a, b: oracle.e2m1_encode(a) | (oracle.e2m1_encode(b) << 4)
    assert pk(0.5, -1.0) == 0xA1 and pk(6.0, 7.5) == 0x77 and pk(2.5, -3.5) == 0xE4
    assert oracle.e2m1_encode(float("nan")) == 0x7 and oracle.e2m1_encode(float("inf")) == 0x7
    assert oracle.e2m1_encode(-float("inf")) == 0xF


def test_e4m3_roundtrip_and_rounding():
    import torch

    # every finite code round-trips; torch's CPU cast is the independent check for rounding
    for b in range(256):
        v = oracle.e4m3_decode(b)
        if np.isnan(v):
            continue
        assert oracle.e4m3_encode(v) == b or (v == 0 and oracle.e4m3_encode(v) == (b & 0x80))
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-3, 0.05, 1, 30, 200)])
    xs = np.concatenate([xs, np.float32([0.0, 2 ** -10, 2 ** -9, 1.5 * 2 ** -10, 447.9, 448, 463.9])])
    want = torch.from_numpy(xs).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = np.array([oracle.e4m3_encode(v) for v in xs], dtype=np.uint8)
    assert np.array_equal(got, want)


def test_to_blocked_bit_exact(golden_dir):
    g = _load(golden_dir, "to_blocked.npz")
    for i in range(4):
        assert np.array_equal(oracle.to_blocked(g[f"in{i}"]), g[f"out{i}"]), i
    # SURVEY 8a swizzle KAT, independent of any import
    r, c = 384, 12
    a = np.arange(r * c, dtype=np.uint32).astype(np.uint8).reshape(r, c)
    out = oracle.to_blocked(a)
    for rr in (0, 31, 32, 127, 128, 383):
        for cc in (0, 3, 4, 11):
            assert out[((rr // 128) * 3 + cc // 4) * 512 + (rr % 32) * 16 + ((rr % 128) // 32) * 4 + cc % 4] == a[rr, cc]


def test_to_blocked_zero_pads_ragged():
    a = np.full((130, 5), 7, dtype=np.uint8)
    out = oracle.to_blocked(a)
    assert out.size == 256 * 8 and int(out.sum()) == 7 * 130 * 5


@pytest.mark.parametrize("acc_model", [0, 1])
def test_quantize_mx_vs_reference_oracle(golden_dir, acc_model):
    g = _load(golden_dir, "quantize_mx.npz")
    tot = bad = 0
    for c in range(int(g["ncases"])):
        R, quest = g[f"meta{c}"]
        q, s, m = oracle.fused_quantize_mx(g[f"x{c}"], g[f"h{c}"], oracle.QUEST if quest else oracle.ABS_MAX,
                                           with_mask=True, acc_model=acc_model)
        assert np.array_equal(s, g[f"e8m0_{c}"].reshape(-1)), f"case {c}: e8m0 differs"
        eq = oracle.codes_equal_mod_zero_sign(q, g[f"e2m1_{c}"])
        tot += eq.size
        bad += int((~eq).sum())
        if quest:  # the reference only defines the mask for quest (epilogue_quant.h:1087-1230)
            assert np.array_equal(m, g[f"mask{c}"].reshape(-1)), f"case {c}: clip mask differs"
    # the fp32 restatement reproduces the fp64 Python oracle code-for-code on these inputs
    assert bad == 0, (bad, tot)


def _blocked(sf_rowmajor):
    return oracle.to_blocked(sf_rowmajor)


def test_gemm_mxfp4_bit_exact(golden_dir):
    g = _load(golden_dir, "gemm_mxfp4.npz")
    for c in range(int(g["ncases"])):
        m, n, k = g[f"meta{c}"]
        d = oracle.gemm_blockscaled(oracle.KIND_MXFP4, g[f"a{c}"], g[f"b{c}"], _blocked(g[f"asf{c}"]),
                                    _blocked(g[f"bsf{c}"]), float(g[f"alpha{c}"]), m, n, k)
        assert np.array_equal(d, g[f"out{c}"]), c


def test_quantize_nv_vs_reference_oracle(golden_dir):
    g = _load(golden_dir, "quantize_nv.npz")
    tot = bad = sbad = 0
    for c in range(int(g["ncases"])):
        (R,) = g[f"meta{c}"]
        q, s = oracle.fused_quantize_nv(g[f"x{c}"], g[f"h{c}"], 6.0, oracle.ABS_MAX)
        want_s = g[f"e4m3_{c}"].reshape(-1)
        sbad += int((s != want_s).sum())
        same = s.repeat(16) == want_s.repeat(16)
        eq = oracle.codes_equal_mod_zero_sign(q, g[f"e2m1_{c}"])
        tot += eq.size
        bad += int((~eq & same).sum())
    # The kernel algorithm (gs*amax/6 -> e4m3, x*rcp(SF/gs)) is not the Python oracle's
    # (amax+1e-8 -> e4m3, x/SF*6) in exact arithmetic; the reference tolerates 1e-1 (nvfp4_test.py:205).
    assert sbad / (tot / 16) <= 1e-2, (sbad, tot)
    assert bad / tot <= 1e-2, (bad, tot)


def test_gemm_nvfp4_bit_exact(golden_dir):
    g = _load(golden_dir, "gemm_nvfp4.npz")
    for c in range(int(g["ncases"])):
        m, n, k = g[f"meta{c}"]
        d = oracle.gemm_blockscaled(oracle.KIND_NVFP4, g[f"a{c}"], g[f"b{c}"], _blocked(g[f"asf{c}"]),
                                    _blocked(g[f"bsf{c}"]), float(g[f"alpha{c}"]), m, n, k)
        assert np.array_equal(d, g[f"out{c}"]), c


def test_mxfp8_pseudoquant_and_gemm(golden_dir):
    g = _load(golden_dir, "gemm_mxfp8.npz")
    for c in range(int(g["ncases"])):
        m, n, k = g[f"meta{c}"]
        for x, q, s in ((g[f"xa{c}"], g[f"a{c}"], g[f"asf{c}"]), (g[f"xb{c}"], g[f"b{c}"], g[f"bsf{c}"])):
            oq, os_ = oracle.pseudoquant_mxfp8(x)
            assert np.array_equal(os_, s) and np.array_equal(oq, q), c
        d = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN, g[f"a{c}"], g[f"b{c}"], _blocked(g[f"asf{c}"]),
                                    _blocked(g[f"bsf{c}"]), 1.0, m, n, k)
        assert np.array_equal(d, g[f"out{c}"]), c
        a_t = np.ascontiguousarray(g[f"a{c}"].T)  # NN: A stored (K, M)  (mxfp8_test.py:92)
        d2 = oracle.gemm_blockscaled(oracle.KIND_MXFP8_NN, a_t, g[f"b{c}"], _blocked(g[f"asf{c}"]),
                                     _blocked(g[f"bsf{c}"]), 1.0, m, n, k)
        assert np.array_equal(d2, d), c


def test_mxfp8_e5m2_operand_extension(golden_dir):
    """BASELINE.json configs[4]'s e5m2-gradient leg (extension: the reference rejects it, so the fixture is built from torch's
    own float8_e5m2 cast + the reference's e4m3 pseudo-quantiser, tests/golden/make_golden_e5m2.py).  Pins the oracle's e5m2
    decode / encode on the whole code space and on 4k values, the e5m2 pseudo-quantiser, and the TN / NN GEMM kinds."""
    g = _load(golden_dir, "gemm_mxfp8_e5m2.npz")
    dec = np.array([oracle.e5m2_decode(b) for b in range(256)], dtype=np.float32)
    want = g["e5m2_decode_f32"]
    assert np.array_equal(np.isnan(dec), np.isnan(want)) and np.array_equal(dec[~np.isnan(dec)], want[~np.isnan(want)])
    enc = np.array([oracle.e5m2_encode(float(v)) for v in g["e5m2_encode_in"]], dtype=np.uint8)
    assert np.array_equal(enc, g["e5m2_encode_out"])
    for c in range(int(g["ncases"])):
        m, n, k = g[f"meta{c}"]
        oq, os_ = oracle.pseudoquant_mxfp8(g[f"xa{c}"], e5m2=True)
        assert np.array_equal(os_, g[f"asf{c}"]) and np.array_equal(oq, g[f"a{c}"]), c
        oq, os_ = oracle.pseudoquant_mxfp8(g[f"xb{c}"])
        assert np.array_equal(os_, g[f"bsf{c}"]) and np.array_equal(oq, g[f"b{c}"]), c
        d = oracle.gemm_blockscaled(oracle.KIND_MXFP8_TN_A5, g[f"a{c}"], g[f"b{c}"], _blocked(g[f"asf{c}"]), _blocked(g[f"bsf{c}"]), 1.0, m, n, k)
        assert np.array_equal(d, g[f"out{c}"]), c
        d2 = oracle.gemm_blockscaled(oracle.KIND_MXFP8_NN_A5, np.ascontiguousarray(g[f"a{c}"].T), g[f"b{c}"], _blocked(g[f"asf{c}"]),
                                     _blocked(g[f"bsf{c}"]), 1.0, m, n, k)
        assert np.array_equal(d2, d), c


# ---- QAT-backward data-prep kernels -----------------------------------------------------------
def _dq_mx(codes_u8, e8m0_u8, alpha):
    """(.., K/2) packed e2m1 + (.., K/32) e8m0 -> float64 dequantised values / alpha (tests/quartet_test.py:76-104)."""
    lut = np.array([0, .5, 1, 1.5, 2, 3, 4, 6, -0., -.5, -1, -1.5, -2, -3, -4, -6], dtype=np.float64)
    c = codes_u8.astype(np.int32)
    v = np.stack([lut[c & 0xF], lut[c >> 4]], axis=-1).reshape(*c.shape[:-1], -1)
    s = np.ldexp(1.0, e8m0_u8.astype(np.int32) - 127)
    return (v.reshape(*v.shape[:-1], -1, 32) * s[..., None]).reshape(v.shape) / alpha


def test_backward_t_bf16_vs_reference_oracle(golden_dir):
    g = _load(golden_dir, "quartet_bwd.npz")
    for c in range(int(g["t_ncases"])):
        for acc_model in (0, 1):
            q, s = oracle.backward_t_bf16(g[f"t_x{c}"], g["h"], acc_model)
            want_q, want_s = g[f"t_e2m1_{c}"], g[f"t_e8m0_{c}"]
            assert np.array_equal(s.reshape(want_s.shape), want_s), (c, acc_model)          # e8m0: bit-exact
            eq = oracle.codes_equal_mod_zero_sign(q.reshape(want_q.shape), want_q)
            assert (~eq).sum() <= 2e-3 * eq.size, (c, acc_model, int((~eq).sum()))            # fp32 vs fp64 rotation ties


def test_backward_qt_bf16_vs_reference_oracle(golden_dir):
    g = _load(golden_dir, "quartet_bwd.npz")
    for c in range(int(g["qt_ncases"])):
        q, s = oracle.backward_qt_bf16(g[f"qt_xq{c}"], g[f"qt_xs{c}"], g["h"], 3.0, acc_model=1)
        want_s, want_dq = g[f"qt_e8m0_{c}"], g[f"qt_dq{c}"]
        assert np.array_equal(s.reshape(want_s.shape), want_s), c
        # the reference asserts equality of the DEQUANTISED values (quartet_test.py:258-260)
        got_dq = _dq_mx(q.reshape(want_s.shape[:-1] + (-1,)), s.reshape(want_s.shape), 3.0)
        bad = (got_dq.astype(np.float32) != want_dq.reshape(got_dq.shape)).sum()
        assert bad <= 2e-3 * got_dq.size, (c, int(bad))


def test_backward_bf16_square_double_mxfp8_bit_exact(golden_dir):
    g = _load(golden_dir, "quartet_bwd.npz")
    for c in range(int(g["sq_ncases"])):
        x = g[f"sq_x{c}"]
        pad = (-x.shape[0]) % 128            # the reference wrapper / oracle pad rows to a multiple of 128
        xp = np.concatenate([x, np.zeros((pad, x.shape[1]), dtype=x.dtype)]) if pad else x
        y, rs, cs = oracle.backward_bf16_square_double_mxfp8(xp)
        assert np.array_equal(rs, g[f"sq_rs{c}"]) and np.array_equal(cs, g[f"sq_cs{c}"]), c
        assert np.array_equal(y, g[f"sq_y{c}"]), (c, int((y != g[f"sq_y{c}"]).sum()))


def test_mxfp4_transpose_mxfp8_bit_exact(golden_dir):
    g = _load(golden_dir, "quartet_bwd.npz")
    for c in range(int(g["tr_ncases"])):
        y, e = oracle.mxfp4_transpose_mxfp8(g[f"tr_xq{c}"], g[f"tr_xs{c}"])
        assert np.array_equal(e, g[f"tr_e{c}"]), c
        assert np.array_equal(y, g[f"tr_y{c}"]), (c, int((y != g[f"tr_y{c}"]).sum()))
