"""torch restatement of the reference's GEMM test oracle -- TEST INFRASTRUCTURE (bench cpu_baseline / tests).

Follows tests/mxfp4_test.py:84-120 (`_dq_fp4`: nibble unpack, 16-entry grid lookup, multiply by the
per-32 e8m0 scale) and :229-231 (`a_dq @ b_dq.T` then cast to bf16); tests/nvfp4_test.py:80-110 for
the e4m3-per-16 variant.  Used as the CPU baseline of bench.py: it is the path the reference itself
runs on the CPU, timed on the host cores.  Never imported by qutlass_amd.
"""
from __future__ import annotations

import torch

_GRID = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0]


def dq_fp4(x_e2m1: torch.Tensor, x_sf: torch.Tensor, gs: int, dtype=torch.float64) -> torch.Tensor:
    """x_e2m1: (rows, K/2) uint8; x_sf: (>=rows, >=K/gs) float8 scales, row-major (un-swizzled)."""
    rows, kh = x_e2m1.shape
    k = kh * 2
    xi = x_e2m1.view(torch.uint8).to(torch.int32)
    codes = torch.stack([xi & 0xF, (xi >> 4) & 0xF], dim=-1).flatten(start_dim=-2)
    vals = torch.tensor(_GRID, dtype=dtype)[codes]
    scales = x_sf[:rows, : k // gs].to(dtype)
    return (vals.unflatten(-1, (-1, gs)) * scales[..., None]).flatten(start_dim=-2)


def dequant_matmul_mxfp4(a_q, a_s, b_q, b_s, alpha: float = 1.0, dtype=torch.float64) -> torch.Tensor:
    a = dq_fp4(a_q, a_s, 32, dtype)
    b = dq_fp4(b_q, b_s, 32, dtype)
    return ((a @ b.T) * alpha).to(torch.bfloat16)


def dequant_matmul_nvfp4(a_q, a_s, b_q, b_s, alpha: float = 1.0, dtype=torch.float64) -> torch.Tensor:
    a = dq_fp4(a_q, a_s, 16, dtype)
    b = dq_fp4(b_q, b_s, 16, dtype)
    return ((a @ b.T) * alpha).to(torch.bfloat16)
