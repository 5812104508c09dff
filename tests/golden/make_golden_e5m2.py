#!/usr/bin/env python3
"""Golden vectors for the e5m2-gradient x e4m3-activation MXFP8 GEMM (BASELINE.json configs[4]).

The reference has NO oracle for this leg: its MXFP8 entry points reject every element type but e4m3
(qutlass/csrc/bindings.cpp:157-160, 196-199) and tests/mxfp8_test.py:26-46 hard-codes e4m3 in `_pseudoquant_mxfp8`.
What can be pinned independently of this repo's C code is (i) the e5m2 format itself and (ii) the dequantise-matmul
result, so this script produces both with PyTorch only:

  * B operand: the reference's own `_pseudoquant_mxfp8` (imported from /root/reference/tests/mxfp8_test.py, not copied);
  * A operand: the same EXPRESSION with the e5m2 constants -- shared exponent floor(log2 amax) - 15 + 128, clamp to
    +-57344, cast with torch's `.to(torch.float8_e5m2)` (round to nearest even) -- written out below;
  * out: (a_dq.double() @ b_dq.double().T).to(bfloat16), as make_golden.py does for the e4m3 cases.

Run in the build container (needs /root/reference): python tests/golden/make_golden_e5m2.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (load_reference(), bits16(), u8())


def pseudoquant_mxfp8_e5m2(x: torch.Tensor):
    """tests/mxfp8_test.py:26-46 with (8, 448, float8_e4m3fn) -> (15, 57344, float8_e5m2)."""
    orig_shape = x.shape
    x = x.reshape(-1, 32)
    absmax = x.abs().max(dim=-1, keepdim=True).values
    shared_exps = torch.where(absmax > 0, torch.log2(absmax).floor().to(torch.uint8) - 15 + 128, 128).to(torch.uint8).view(torch.float8_e8m0fnu)
    xq = torch.clamp(x / shared_exps.to(x.dtype), -57344.0, 57344.0).to(torch.float8_e5m2)
    xdq = xq.to(x.dtype) * shared_exps.to(x.dtype)
    return xdq.reshape(orig_shape), (xq.reshape(orig_shape), shared_exps.reshape(orig_shape[:-1] + (orig_shape[-1] // 32,)))


def main():
    _utils, _mx, _nv, f8 = mg.load_reference()
    torch.manual_seed(11)
    d, case = {}, 0
    # gradients have a wide dynamic range: the randn cases are scaled per row by 2^U(-8, 8)
    for (m, n, k, dist) in [(16, 64, 256, "randn"), (40, 72, 384, "wide"), (128, 128, 128, "rand"), (48, 136, 1056, "wide")]:
        a = (torch.rand if dist == "rand" else torch.randn)(m, k, dtype=torch.bfloat16) * 25.0
        if dist == "wide":
            a = a * torch.exp2(torch.randint(-8, 9, (m, 1)).float()).to(torch.bfloat16)
        b = torch.randn(n, k, dtype=torch.bfloat16) * 25.0
        a_dq, (a_q, a_s) = pseudoquant_mxfp8_e5m2(a)
        b_dq, (b_q, b_s) = f8._pseudoquant_mxfp8(b)
        out = (a_dq.double() @ b_dq.double().T).to(torch.bfloat16)
        d[f"xa{case}"], d[f"xb{case}"] = mg.bits16(a), mg.bits16(b)
        d[f"a{case}"], d[f"b{case}"] = mg.u8(a_q), mg.u8(b_q)
        d[f"asf{case}"], d[f"bsf{case}"] = mg.u8(a_s), mg.u8(b_s)
        d[f"out{case}"] = mg.bits16(out)
        d[f"meta{case}"] = np.array([m, n, k])
        case += 1
    d["ncases"] = np.array(case)
    # the whole e5m2 code space as decoded by torch, and torch's RNE encoding of a spread of fp32 values
    codes = torch.arange(256, dtype=torch.uint8)
    d["e5m2_decode_f32"] = codes.view(torch.float8_e5m2).float().numpy()
    vals = torch.cat([torch.randn(4096) * torch.exp2(torch.randint(-18, 16, (4096,)).float()),
                      torch.tensor([0.0, -0.0, 57344.0, -57344.0, 2.0 ** -16, 2.0 ** -17, 1.5 * 2.0 ** -17, 61440.0 - 1.0])])
    vals = vals[vals.abs() <= 57344.0]
    d["e5m2_encode_in"] = vals.numpy()
    d["e5m2_encode_out"] = vals.to(torch.float8_e5m2).view(torch.uint8).numpy()
    np.savez_compressed(os.path.join(HERE, "gemm_mxfp8_e5m2.npz"), **d)
    print("wrote gemm_mxfp8_e5m2.npz:", case, "GEMM cases,", vals.numel(), "encode samples")


if __name__ == "__main__":
    main()
