"""Operator layer: ``torch.ops._qutlass_C.*`` (the 9 hot-path ops of the reference's binding file
``qutlass/csrc/bindings.cpp``) and ``torch.ops.qutlass_amd.to_blocked``.

The ops are implemented in the in-tree C++ extension ``qutlass_amd/_C.so`` (``csrc/torch_ext.cpp``,
LibTorch stable ABI, no device code): argument validation with the reference's order and messages,
output allocation, current-stream lookup, then ONE call into the C ABI of ``libqutlass_amd.so``
(``include/qutlass_amd.h``), where the hand-written HIP kernels live.  The functions below are plain
forwarding stubs with the reference's argument order; there is no Python-side compute and no
fallback -- if the extension is not built, loading it raises.
"""
from __future__ import annotations

import os

import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(_HERE, "_C.so")

# exact schema strings of bindings.cpp:499-513 for the ops this build provides (csrc/torch_ext.cpp registers them)
SCHEMAS = {
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

_registered = False


def register_torch_ops() -> None:
    """Load the C++ extension, which registers ``_qutlass_C::*`` and ``qutlass_amd::to_blocked`` with the
    dispatcher (reference: bindings.cpp:498-535 + registration.h)."""
    global _registered
    if _registered:
        return
    _lib.load()  # libqutlass_amd.so first (the extension links against it)
    if not os.path.exists(EXT_PATH):
        raise ImportError(
            f"{EXT_PATH} is missing: build the extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or python qutlass_amd/build.py). "
            "qutlass_amd has no CPU / eager fallback."
        )
    torch.ops.load_library(EXT_PATH)
    _registered = True


def _C():
    register_torch_ops()
    return torch.ops._qutlass_C


def matmul_mxf4_bf16_tn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:32-66 -> qutlass_amd_matmul_mxf4_bf16_tn."""
    return _C().matmul_mxf4_bf16_tn(A, B, A_sf, B_sf, alpha)


def matmul_ada_mxf4_bf16_tn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:104-138 -> qutlass_amd_matmul_ada_mxf4_bf16_tn (row-major, un-swizzled scales)."""
    return _C().matmul_ada_mxf4_bf16_tn(A, B, A_sf, B_sf, alpha)


def matmul_nvf4_bf16_tn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:68-102 -> qutlass_amd_matmul_nvf4_bf16_tn."""
    return _C().matmul_nvf4_bf16_tn(A, B, A_sf, B_sf, alpha)


def matmul_mxf8_bf16_tn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:140-177 -> qutlass_amd_matmul_mxf8_bf16_tn."""
    return _C().matmul_mxf8_bf16_tn(A, B, A_sf, B_sf, alpha)


def matmul_mxf8_bf16_nn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:179-216 -> qutlass_amd_matmul_mxf8_bf16_nn (A stored (K, M))."""
    return _C().matmul_mxf8_bf16_nn(A, B, A_sf, B_sf, alpha)


def fusedQuantizeMxQuest(A, B, OUT, OUT_sf):
    """bindings.cpp:218-252."""
    return _C().fusedQuantizeMxQuest(A, B, OUT, OUT_sf)


def fusedQuantizeMxAbsMax(A, B, OUT, OUT_sf):
    """bindings.cpp:292-333."""
    return _C().fusedQuantizeMxAbsMax(A, B, OUT, OUT_sf)


def fusedQuantizeMxQuestWithMask(A, B, OUT, OUT_sf, OUT_mask):
    """bindings.cpp:255-289 (rotation size 32 only)."""
    return _C().fusedQuantizeMxQuestWithMask(A, B, OUT, OUT_sf, OUT_mask)


def fusedQuantizeNvQuest(A, B, OUT, OUT_sf, global_scale):
    """bindings.cpp:335-378."""
    return _C().fusedQuantizeNvQuest(A, B, OUT, OUT_sf, global_scale)


def fusedQuantizeNvAbsMax(A, B, OUT, OUT_sf, global_scale):
    """bindings.cpp:380-426."""
    return _C().fusedQuantizeNvAbsMax(A, B, OUT, OUT_sf, global_scale)


def backward_t_bf16(x, h, xh_e2m1, xh_e8m0):
    """bindings.cpp:429-443."""
    return _C().backward_t_bf16(x, h, xh_e2m1, xh_e8m0)


def backward_qt_bf16(x_e2m1, x_e8m0, h, alpha, xh_e2m1, xh_e8m0):
    """bindings.cpp:445-464."""
    return _C().backward_qt_bf16(x_e2m1, x_e8m0, h, alpha, xh_e2m1, xh_e8m0)


def backward_bf16_square_double_mxfp8(x_bf16, x_fp8, row_scales, column_scales):
    """bindings.cpp:466-479."""
    return _C().backward_bf16_square_double_mxfp8(x_bf16, x_fp8, row_scales, column_scales)


def mxfp4_transpose_mxfp8(x_fp4, scales, x_fp8, shared_exps):
    """bindings.cpp:481-494."""
    return _C().mxfp4_transpose_mxfp8(x_fp4, scales, x_fp8, shared_exps)


def to_blocked(input_matrix: torch.Tensor) -> torch.Tensor:
    """Block-scale swizzle as a device op (replaces the torch/Triton paths of qutlass/utils.py:160-193)."""
    register_torch_ops()
    return torch.ops.qutlass_amd.to_blocked(input_matrix)
