"""Mirror of the reference's ``qutlass/utils.py`` public helpers (same names and semantics).

``to_blocked`` is a HIP kernel here (``csrc/to_blocked.hip.h``) -- there is no Triton and no torch
fallback; ``use_triton_kernel`` is accepted for call-site compatibility and ignored (both of the
reference's paths produce the same bytes; like its Triton path, ragged shapes are zero-padded).
"""
from __future__ import annotations

import torch


def ceil_div(a, b):
    return (a + b - 1) // b


def get_padded_shape_mx(a: torch.Tensor):
    """qutlass/utils.py:140-147."""
    rows, cols = a.numel() // a.size(-1), a.size(-1) // 32
    return ceil_div(rows, 128) * 128, ceil_div(cols, 4) * 4


def get_padded_shape_nv(a: torch.Tensor):
    """qutlass/utils.py:150-157."""
    rows, cols = a.numel() // a.size(-1), a.size(-1) // 16
    return ceil_div(rows, 128) * 128, ceil_div(cols, 4) * 4


def to_blocked(input_matrix: torch.Tensor, use_triton_kernel: bool = False) -> torch.Tensor:
    """qutlass/utils.py:160-193: (H, W) 1-byte matrix -> flat 128x4-tiled block-scale layout of
    32*ceil(H/128) x 16*ceil(W/4) bytes, same dtype."""
    del use_triton_kernel
    from . import ops

    return ops.to_blocked(input_matrix)


def pad_to_block(tensor, dims, blocksize):
    """qutlass/utils.py:196-204 (public helper of the reference; nothing in this package calls it any more -- the ops that needed
    row padding do it inside their kernels): zero-extend `tensor` at the end of every dimension in `dims` to a multiple of
    `blocksize`."""
    shape = list(tensor.shape)
    for d in dims:
        shape[d] = -(-shape[d] // blocksize) * blocksize
    out = tensor.new_zeros(shape)
    out[tuple(slice(0, n) for n in tensor.shape)] = tensor
    return out
