"""qutlass_amd -- MI355X-native (gfx950 / CDNA4) implementation of the qutlass operator surface.

Public API = the hot-path functions of the reference's ``qutlass/__init__.py`` (same names, argument
meaning, defaults and Python-level error behaviour):

    fusedQuantizeMx, fusedQuantizeNv, matmul_mxf4_bf16_tn, matmul_nvf4_bf16_tn,
    matmul_mxf8_bf16_tn, matmul_mxf8_bf16_nn         (+ qutlass_amd.utils.to_blocked & friends)

All compute is hand-written HIP behind the C ABI of ``include/qutlass_amd.h``
(``libqutlass_amd.so``); importing this package loads that library and registers
``torch.ops._qutlass_C.*``.  There is no CPU or eager-PyTorch fallback: if the library is not built
the import fails, and every op requires GPU tensors.
"""
from __future__ import annotations

from typing import Literal

import torch

from . import _lib, ops
from .utils import get_padded_shape_mx, get_padded_shape_nv, pad_to_block, to_blocked  # noqa: F401

__version__ = "0.1.0"

_lib.load()                 # fail loudly at import time if the HIP library is missing
ops.register_torch_ops()    # torch.ops._qutlass_C.<op>  (reference: bindings.cpp:498-535)

qutlass_CUDA = torch.ops._qutlass_C

_FLASHINFER_MSG = (
    "flashinfer backend requested but not installed. flashinfer/cuDNN is an NVIDIA-only backend and is "
    "not available in the MI355X build; use backend='cutlass' (the native gfx950 kernels)."
)


def matmul_mxf4_bf16_tn(a: torch.Tensor, b: torch.Tensor, a_sf: torch.Tensor, b_sf: torch.Tensor,
                        alpha: torch.Tensor,
                        backend: Literal["cutlass", "flashinfer"] = "cutlass") -> torch.Tensor:
    """qutlass/__init__.py:34-76.  ``backend="cutlass"`` selects the native kernel (name kept for
    drop-in compatibility); ``"flashinfer"`` raises ImportError exactly as the reference does when
    flashinfer is absent."""
    if backend == "cutlass":
        return qutlass_CUDA.matmul_mxf4_bf16_tn(a, b, a_sf, b_sf, alpha)
    elif backend == "flashinfer":
        raise ImportError(_FLASHINFER_MSG)
    else:
        raise ValueError(f"invalid backend {backend!r}; use 'cutlass' or 'flashinfer'")


def matmul_nvf4_bf16_tn(a: torch.Tensor, b: torch.Tensor, a_sf: torch.Tensor, b_sf: torch.Tensor,
                        alpha: torch.Tensor,
                        backend: Literal["cutlass", "flashinfer"] = "cutlass") -> torch.Tensor:
    """qutlass/__init__.py:89-131."""
    if backend == "cutlass":
        return qutlass_CUDA.matmul_nvf4_bf16_tn(a, b, a_sf, b_sf, alpha)
    elif backend == "flashinfer":
        raise ImportError(_FLASHINFER_MSG)
    else:
        raise ValueError(f"invalid backend {backend!r}; use 'cutlass' or 'flashinfer'")


def matmul_mxf8_bf16_tn(a: torch.Tensor, b: torch.Tensor, block_scale_a: torch.Tensor,
                        block_scale_b: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """qutlass/__init__.py:134-139."""
    return qutlass_CUDA.matmul_mxf8_bf16_tn(a, b, block_scale_a, block_scale_b, alpha)


def matmul_mxf8_bf16_nn(a: torch.Tensor, b: torch.Tensor, block_scale_a: torch.Tensor,
                        block_scale_b: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """qutlass/__init__.py:141-146 (A stored (K, M))."""
    return qutlass_CUDA.matmul_mxf8_bf16_nn(a, b, block_scale_a, block_scale_b, alpha)


def fusedQuantizeMx(a: torch.Tensor, b: torch.Tensor, *, method: Literal["quest", "abs_max"] = "quest",
                    return_mask: bool = False):
    """qutlass/__init__.py:149-180: allocate packed e2m1 + (padded_rows, padded_cols) e8m0 [+ clip mask]
    and run the fused rotate+quantize kernel.  As in the reference the scale buffer is written flat
    (first numel/32 bytes) and its padding is left uninitialised."""
    padded_rows, padded_cols = get_padded_shape_mx(a)
    xh_e2m1 = torch.empty(*a.shape[:-1], a.size(-1) // 2, dtype=torch.uint8, device=a.device)
    xh_e8m0 = torch.empty(padded_rows, padded_cols, dtype=torch.float8_e8m0fnu, device=a.device)

    if method == "quest":
        if return_mask:
            clip_mask = torch.empty(*a.shape[:-1], a.size(-1) // 8, dtype=torch.uint8, device=a.device)
            return qutlass_CUDA.fusedQuantizeMxQuestWithMask(a, b, xh_e2m1, xh_e8m0, clip_mask)
        else:
            return qutlass_CUDA.fusedQuantizeMxQuest(a, b, xh_e2m1, xh_e8m0)
    elif method == "abs_max":
        if return_mask:
            raise ValueError("return_mask is only supported for method 'quest'")
        return qutlass_CUDA.fusedQuantizeMxAbsMax(a, b, xh_e2m1, xh_e8m0)
    else:
        raise ValueError(f"invalid method {method!r}, must be 'quest' or 'abs_max'")


def fusedQuantizeNv(a: torch.Tensor, b: torch.Tensor, global_scale: torch.Tensor, *,
                    method: Literal["quest", "abs_max"] = "abs_max") -> tuple[torch.Tensor, torch.Tensor]:
    """qutlass/__init__.py:183-203."""
    padded_rows, padded_cols = get_padded_shape_nv(a)
    xh_e2m1 = torch.empty(*a.shape[:-1], a.size(-1) // 2, dtype=torch.uint8, device=a.device)
    xh_e4m3 = torch.empty(padded_rows, padded_cols, dtype=torch.float8_e4m3fn, device=a.device)

    if method == "quest":
        return qutlass_CUDA.fusedQuantizeNvQuest(a, b, xh_e2m1, xh_e4m3, global_scale)
    elif method == "abs_max":
        return qutlass_CUDA.fusedQuantizeNvAbsMax(a, b, xh_e2m1, xh_e4m3, global_scale)
    else:
        raise ValueError(f"invalid method {method!r}, must be 'quest' or 'abs_max'")


_OUT_OF_SCOPE = ("matmul_ada_mxf4_bf16_tn", "backward_t_bf16", "backward_qt_bf16",
                 "backward_bf16_square_double_mxfp8", "mxfp4_transpose_mxfp8")


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise AttributeError(
            f"qutlass_amd does not provide {name!r} yet: it is outside the hot path this build covers "
            "(SURVEY.md section 8f lists it as 'next')."
        )
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
