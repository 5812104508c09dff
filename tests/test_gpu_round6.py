"""Round-6 GPU parity tests (nothing here reads /root/reference).

  * Quest on (nearly) CONSTANT groups: the sign of the fp32 variance decides the reference's arm -- MX: `if (var >= 0)` else scale 1.0 (epilogue_quant.h:531-535); NV: no
    guard at all, sqrt of the negative variance is stored as the NaN scale byte 0x7f and every code of the group becomes +-0 (epilogue_quant.h:1631-1640).  The kernel
    re-sums such groups in the reference's order (quantize.hip.h `quest_sums_in_reference_order`), so scale bytes are compared EXACTLY against the oracle's sequential sums.
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


def nearly_constant_groups(n: int, rot: int, seed: int) -> torch.Tensor:
    """n rotation groups (a, a 2^-k, 0, ..., 0), k = 9 .. 14: rotated by a Hadamard matrix every output is +-(a +- a 2^-k) / sqrt(rot) -- a group whose spread is at the
    rounding noise of its fp32 sum of squares (also used by the CPU half, tests/test_round6_cpu.py)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((n, rot), np.float32)
    a = rng.standard_normal(n).astype(np.float32) * 8
    x[:, 0] = a
    x[:, 1] = a * (2.0 ** -rng.integers(9, 15, size=n))
    return torch.from_numpy(x).to(torch.bfloat16)


@pytest.mark.parametrize("rot", [16, 32, 64, 128])
def test_fused_quantize_nv_quest_nearly_constant_groups_nan_scale_byte(q, rot):
    x = nearly_constant_groups(4096, rot, 6).to(DEV)
    h = _hadamard(rot)
    gs = torch.tensor([1.0], device=DEV)
    e2m1, e4m3 = q.fusedQuantizeNv(x, h, gs, method="quest")
    rq, rs = oracle.fused_quantize_nv(_np(x), _np(h), 1.0, oracle.QUEST)
    rs = np.asarray(rs).reshape(-1)
    got_s = _np(e4m3).reshape(-1)[: rs.size]
    if rot == 32:   # (1 / sqrt(32) is not a bf16: long mantissas, about a fifth of these groups come out with a negative variance)
        assert int((rs == 0x7F).sum()) > 500
    bad = np.nonzero(got_s != rs)[0]
    assert bad.size == 0, f"{bad.size} e4m3 scale bytes differ, first groups {bad[:8]}: got {got_s[bad[:8]]}, oracle {rs[bad[:8]]}"
    got_q = _np(e2m1).reshape(-1, 8)
    nan_groups = rs == 0x7F
    assert not (got_q[nan_groups] & 0x77).any(), "a group with a NaN scale byte holds codes other than +-0"
    eq = oracle.codes_equal_mod_zero_sign(_np(e2m1), rq)
    assert int((~eq).sum()) <= 2, f"{int((~eq).sum())} code bytes differ"


@pytest.mark.parametrize("rot", [32, 64, 128])
@pytest.mark.parametrize("mask", [False, True])
def test_fused_quantize_mx_quest_nearly_constant_groups(q, rot, mask):
    if mask and rot != 32:
        pytest.skip("the clip-mask quantizer takes rotation 32 only (fused_quantize_mx_mask.cu:107-123)")
    x = nearly_constant_groups(4096, rot, 7).to(DEV)
    h = _hadamard(rot)
    out = q.fusedQuantizeMx(x, h, method="quest", return_mask=mask)
    rq, rs, rm = oracle.fused_quantize_mx(_np(x), _np(h), oracle.QUEST, with_mask=mask)
    rs = np.asarray(rs).reshape(-1)
    got_s = _np(out[1]).reshape(-1)[: rs.size]
    if rot == 32:
        assert int((rs == 127).sum()) > 500   # the `var < 0` arm: scale 1.0
    bad = np.nonzero(got_s != rs)[0]
    assert bad.size == 0, f"{bad.size} e8m0 bytes differ, first groups {bad[:8]}: got {got_s[bad[:8]]}, oracle {rs[bad[:8]]}"
    eq = oracle.codes_equal_mod_zero_sign(_np(out[0]), rq)
    assert eq.all(), f"{int((~eq).sum())} code bytes differ"
    if mask:
        assert np.array_equal(_np(out[2]).reshape(-1), np.asarray(rm).reshape(-1))


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
