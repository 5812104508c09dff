"""Round-6 CPU tests: oracle behaviour added this round (no GPU, nothing reads /root/reference)."""
import numpy as np
import torch

import oracle


def _nearly_constant_groups(n, rot, seed):   # tests/test_gpu_round6.py uses the same construction on the GPU
    rng = np.random.default_rng(seed)
    x = np.zeros((n, rot), np.float32)
    a = rng.standard_normal(n).astype(np.float32) * 8
    x[:, 0] = a
    x[:, 1] = a * (2.0 ** -rng.integers(9, 15, size=n))
    return torch.from_numpy(x).to(torch.bfloat16).view(torch.uint16).numpy()


def _hadamard_bits(n):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).view(torch.uint16).numpy()


def test_nv_quest_negative_variance_gives_the_nan_scale_byte_and_zero_codes():
    """epilogue_quant.h:1631-1640: `std::sqrt(c_sum2 * rcp(16) - c_mean * c_mean) * (2.92247856 / 6.) + 1e-8` with no guard -- a negative fp32 variance (rounding, on a
    nearly constant group) is NaN all the way into `__nv_fp8_e4m3(scale)` = 0x7f; `scale_q > 0` is then false, the multiplier 0 and every code +-0.  Rounds 1-5 clamped."""
    x, h = _nearly_constant_groups(4096, 32, 6), _hadamard_bits(32)
    q, s = oracle.fused_quantize_nv(x, h, 1.0, oracle.QUEST)
    s = np.asarray(s).reshape(-1)
    q = np.asarray(q).reshape(-1, 8)
    nan = s == 0x7F
    assert 500 < int(nan.sum()) < s.size
    assert not (q[nan] & 0x77).any()
    assert oracle.e4m3_decode(0x7F) != oracle.e4m3_decode(0x7F)   # NaN
    # abs_max never produces it on finite data
    _, s2 = oracle.fused_quantize_nv(x, h, 1.0, oracle.ABS_MAX)
    assert not (np.asarray(s2).reshape(-1) == 0x7F).any()


def test_mx_quest_negative_variance_takes_scale_one():
    """epilogue_quant.h:531-535: `float scale = 1.0; if (var >= 0) scale = ...` -- byte 127."""
    x, h = _nearly_constant_groups(4096, 32, 7), _hadamard_bits(32)
    _, s, _ = oracle.fused_quantize_mx(x, h, oracle.QUEST)
    s = np.asarray(s).reshape(-1)
    assert 500 < int((s == 127).sum()) < s.size
    # the summation ORDER decides some of these groups: the kernel's lane order (acc_model 2) disagrees with the reference's sequential order on a few -- which is
    # why quantize.hip.h re-sums such groups sequentially (quest_sums_in_reference_order)
    _, s_lane, _ = oracle.fused_quantize_mx(x, h, oracle.QUEST, acc_model=2)
    assert int((np.asarray(s_lane).reshape(-1) != s).sum()) > 0
