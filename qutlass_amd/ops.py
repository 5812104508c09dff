"""Operator registration: ``torch.ops._qutlass_C.*`` (the ops of the reference's binding file ``qutlass/csrc/bindings.cpp``)
and ``torch.ops.qutlass_amd.to_blocked``.

The ops live in the in-tree C++ extension ``qutlass/_CUDA.abi3.so`` (``csrc/torch_ext.cpp``, LibTorch stable ABI, no device
code) -- the same module name the reference's op library has (bindings.cpp:537-540): argument validation with the
reference's order and messages, output allocation, current-stream lookup, then ONE call into the C ABI of
``libqutlass_amd.so`` (``include/qutlass_amd.h``), where the hand-written HIP kernels are.  No Python-side compute and no
fallback: if the extension is not built, loading it raises.
"""
from __future__ import annotations

import os

import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
# QUTLASS_AMD_OP_LIBRARY: another build of the op library, e.g. the trimmed one (qutlass_amd.build.build_extension(minimal=True), the reference's QUTLASS_MINIMAL_BUILD)
EXT_PATH = os.environ.get("QUTLASS_AMD_OP_LIBRARY") or os.path.join(os.path.dirname(_HERE), "qutlass", "_CUDA.abi3.so")

_registered = False


def register_torch_ops() -> None:
    """Load the C++ extension, which registers ``_qutlass_C::*`` and ``qutlass_amd::to_blocked`` with the dispatcher
    (reference: bindings.cpp:498-535 + registration.h).  The file is the Python extension module ``qutlass._CUDA``; it is
    loaded by path here so that ``import qutlass_amd`` does not depend on the alias package, and ``import qutlass._CUDA``
    afterwards finds the same, already initialised library."""
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
    _register_fakes()


def _register_fakes() -> None:
    """Shape-only ("fake" / meta) kernels for every op of the extension, so that callers can be traced: `torch.compile(fullgraph=True)`, `make_fx`,
    `torch.export` under FakeTensorMode.  The reference's default `to_blocked` is plain torch written to be compiled through
    (qutlass/utils.py:160-193); here it is a custom op, and without a fake kernel a compiled caller graph-breaks or fails.  The 14 `_qutlass_C` ops get
    one too (beyond the reference, whose ops have none).  Outputs mirror csrc/torch_ext.cpp: GEMMs allocate (M, N) bf16; the quantizers return their
    OUT / OUT_sf (/ OUT_mask) arguments -- the fake hands back fresh tensors of the same metadata (a fake kernel must not alias its inputs);
    the backward data-prep ops return nothing (they fill caller-provided tensors, as in bindings.cpp:429-494)."""
    def rf(qualname):   # (a trimmed op library -- QUTLASS_MINIMAL_BUILD -- does not define the training-only ops: nothing to register for them)
        ns, op = qualname.split("::")
        if hasattr(getattr(torch.ops, ns), op):
            return torch.library.register_fake(qualname)
        return lambda fn: fn

    def gemm_tn(A, B, A_sf, B_sf, alpha):
        return A.new_empty((A.size(0), B.size(0)), dtype=torch.bfloat16)

    for name in ("matmul_mxf4_bf16_tn", "matmul_nvf4_bf16_tn", "matmul_ada_mxf4_bf16_tn", "matmul_mxf8_bf16_tn"):
        rf(f"_qutlass_C::{name}")(gemm_tn)

    @rf("_qutlass_C::matmul_mxf8_bf16_nn")
    def _(A, B, A_sf, B_sf, alpha):   # A is (K, M) (bindings.cpp:185-214)
        return A.new_empty((A.size(1), B.size(0)), dtype=torch.bfloat16)

    def quant2(A, R, OUT, OUT_sf):
        return torch.empty_like(OUT), torch.empty_like(OUT_sf)

    for name in ("fusedQuantizeMxQuest", "fusedQuantizeMxAbsMax"):
        rf(f"_qutlass_C::{name}")(quant2)

    @rf("_qutlass_C::fusedQuantizeMxQuestWithMask")
    def _(A, R, OUT, OUT_sf, OUT_mask):
        return torch.empty_like(OUT), torch.empty_like(OUT_sf), torch.empty_like(OUT_mask)

    def quant_nv(A, R, OUT, OUT_sf, global_scale):
        return torch.empty_like(OUT), torch.empty_like(OUT_sf)

    for name in ("fusedQuantizeNvQuest", "fusedQuantizeNvAbsMax"):
        rf(f"_qutlass_C::{name}")(quant_nv)

    @rf("_qutlass_C::backward_t_bf16")
    def _(x, h, xh_e2m1, xh_e8m0):
        return None

    @rf("_qutlass_C::backward_qt_bf16")
    def _(x_e2m1, x_e8m0, h, alpha, xh_e2m1, xh_e8m0):
        return None

    @rf("_qutlass_C::backward_bf16_square_double_mxfp8")
    def _(x_bf16, x_fp8, row_scales, column_scales):
        return None

    @rf("_qutlass_C::mxfp4_transpose_mxfp8")
    def _(x_fp4, scales, x_fp8, shared_exps):
        return None

    @rf("qutlass_amd::to_blocked")
    def _(input_matrix):
        rows, cols = input_matrix.shape
        return input_matrix.new_empty(((rows + 127) // 128 * 128) * ((cols + 3) // 4 * 4))

    @rf("qutlass_amd::fusedQuantizeMxBlocked")
    def _(A, R, OUT, OUT_sf, method):
        return torch.empty_like(OUT), torch.empty_like(OUT_sf)

    @rf("qutlass_amd::fusedQuantizeNvBlocked")
    def _(A, R, OUT, OUT_sf, global_scale, method):
        return torch.empty_like(OUT), torch.empty_like(OUT_sf)

    @rf("qutlass_amd::fusedQuantizeMatmulMxf4")
    def _(X, R, B, B_sf, alpha, method):
        k = X.size(-1)
        return X.new_empty((X.numel() // k if k else 0, B.size(0)), dtype=torch.bfloat16)


def to_blocked(input_matrix: torch.Tensor) -> torch.Tensor:
    """qutlass/utils.py:160-193 as a HIP kernel (csrc/to_blocked.hip.h) -> flat blocked byte vector, input dtype."""
    register_torch_ops()
    return torch.ops.qutlass_amd.to_blocked(input_matrix)
