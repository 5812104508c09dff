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
    _register_fakes()
    _registered = True   # only once the fake kernels are in place: a failed registration is retried (and raises again) on the next call


def _register_fakes() -> None:
    """Shape-only ("fake" / meta) kernels for every op of the extension, so that callers can be traced: `torch.compile(fullgraph=True)`, `make_fx`,
    `torch.export` under FakeTensorMode.  The reference's default `to_blocked` is plain torch written to be compiled through
    (qutlass/utils.py:160-193); here it is a custom op, and without a fake kernel a compiled caller graph-breaks or fails.  The 14 `_qutlass_C` ops get
    one where the schema tells the truth (the five GEMMs: they allocate (M, N) bf16); the ops that fill caller tensors are traced through their
    mutation-declaring twins in the `qutlass_amd` namespace (csrc/torch_ext.cpp), which return nothing."""
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

    # The reference's output-filling ops -- the five quantizers and the four QAT-backward data-prep ops of `_qutlass_C` -- get NO fake kernel: their
    # schemas (bindings.cpp:504-513, kept verbatim) declare neither the writes nor the aliasing returns, so a traced graph would treat the call as
    # dead code (AOTAutograd drops a `-> ()` op without declared mutation; inductor reuses OUT's storage while the returned alias is live).  Tracing
    # them fails loudly instead; the Python wrappers call the `qutlass_amd::*_` twins below, whose schemas declare `Tensor(a!)` and return nothing.
    def fills(*args):
        return None

    for name in ("fusedQuantizeMx_", "fusedQuantizeNv_", "fusedQuantizeMxMask_", "fusedQuantizeMxBlocked", "fusedQuantizeNvBlocked",
                 "backward_t_bf16_", "backward_qt_bf16_", "backward_bf16_square_double_mxfp8_", "mxfp4_transpose_mxfp8_"):
        rf(f"qutlass_amd::{name}")(fills)

    @rf("qutlass_amd::to_blocked")
    def _(input_matrix):
        rows, cols = input_matrix.shape
        return input_matrix.new_empty(((rows + 127) // 128 * 128) * ((cols + 3) // 4 * 4))

    @rf("qutlass_amd::fusedQuantizeMatmulMxf4")
    def _(X, R, B, B_sf, alpha, method):
        k = X.size(-1)
        return X.new_empty((X.numel() // k if k else 0, B.size(0)), dtype=torch.bfloat16)


def to_blocked(input_matrix: torch.Tensor) -> torch.Tensor:
    """qutlass/utils.py:160-193 as a HIP kernel (csrc/to_blocked.hip.h) -> flat blocked byte vector, input dtype."""
    register_torch_ops()
    return torch.ops.qutlass_amd.to_blocked(input_matrix)
