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
    _define_functional_ops()
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


def _define_functional_ops() -> None:
    """FUNCTIONAL forms of the output-filling ops (`qutlass_amd::quantize_mx` ...: allocate, call the in-place twin, return fresh tensors), defined in Python with
    `torch.library.custom_op`.  The wrappers of qutlass_amd/__init__.py call them only while a graph is being compiled (`torch.compiler.is_compiling()`): inductor (torch
    2.10) refuses every node that touches a `float8_e8m0fnu` tensor except views / cat / clone / `_scaled_mm` (torch/_inductor/lowering.py `unsupported_input_tensor`), so
    the `auto_functionalized` wrapper of a mutating op with an e8m0 argument is never decomposed and compilation dies ("auto_functionalized_v2 was not removed") -- while a
    functional op with e8m0 results is simply called as an extern kernel.  Eager callers keep the direct C++ path (a Python custom op costs ~10 us per call)."""
    if hasattr(torch.ops.qutlass_amd, "quantize_mx"):
        return
    from torch.library import custom_op

    amd = torch.ops.qutlass_amd
    have_training_ops = hasattr(amd, "backward_t_bf16_")

    def _mx(a, blocked=False):
        rows, cols = a.numel() // a.size(-1), a.size(-1) // 32
        pr, pc = (rows + 127) // 128 * 128, (cols + 3) // 4 * 4
        return (a.new_empty((*a.shape[:-1], a.size(-1) // 2), dtype=torch.uint8),
                a.new_empty((pr * pc,) if blocked else (pr, pc), dtype=torch.float8_e8m0fnu))

    def _nv(a, blocked=False):
        rows, cols = a.numel() // a.size(-1), a.size(-1) // 16
        pr, pc = (rows + 127) // 128 * 128, (cols + 3) // 4 * 4
        return (a.new_empty((*a.shape[:-1], a.size(-1) // 2), dtype=torch.uint8),
                a.new_empty((pr * pc,) if blocked else (pr, pc), dtype=torch.float8_e4m3fn))

    @custom_op("qutlass_amd::quantize_mx", mutates_args=(), schema="(Tensor A, Tensor R, int method) -> (Tensor, Tensor)")
    def quantize_mx(A, R, method):
        o = _mx(A)
        amd.fusedQuantizeMx_(A, R, o[0], o[1], method)
        return o

    quantize_mx.register_fake(lambda A, R, method: _mx(A))

    @custom_op("qutlass_amd::quantize_nv", mutates_args=(), schema="(Tensor A, Tensor R, Tensor global_scale, int method) -> (Tensor, Tensor)")
    def quantize_nv(A, R, global_scale, method):
        o = _nv(A)
        amd.fusedQuantizeNv_(A, R, o[0], o[1], global_scale, method)
        return o

    quantize_nv.register_fake(lambda A, R, global_scale, method: _nv(A))

    @custom_op("qutlass_amd::quantize_mx_blocked", mutates_args=(), schema="(Tensor A, Tensor R, int method) -> (Tensor, Tensor)")
    def quantize_mx_blocked(A, R, method):
        o = _mx(A, True)
        amd.fusedQuantizeMxBlocked(A, R, o[0], o[1], method)
        return o

    quantize_mx_blocked.register_fake(lambda A, R, method: _mx(A, True))

    @custom_op("qutlass_amd::quantize_nv_blocked", mutates_args=(), schema="(Tensor A, Tensor R, Tensor global_scale, int method) -> (Tensor, Tensor)")
    def quantize_nv_blocked(A, R, global_scale, method):
        o = _nv(A, True)
        amd.fusedQuantizeNvBlocked(A, R, o[0], o[1], global_scale, method)
        return o

    quantize_nv_blocked.register_fake(lambda A, R, global_scale, method: _nv(A, True))

    if not have_training_ops:   # QUTLASS_MINIMAL_BUILD: inference ops only
        return

    def _mask(a):
        return _mx(a) + (a.new_empty((*a.shape[:-1], a.size(-1) // 8), dtype=torch.uint8),)

    @custom_op("qutlass_amd::quantize_mx_mask", mutates_args=(), schema="(Tensor A, Tensor R) -> (Tensor, Tensor, Tensor)")
    def quantize_mx_mask(A, R):
        o = _mask(A)
        amd.fusedQuantizeMxMask_(A, R, o[0], o[1], o[2])
        return o

    quantize_mx_mask.register_fake(lambda A, R: _mask(A))

    def _bt(x):   # (.., N, M) -> (.., M, N/2) e2m1x2, (.., M, N/32) e8m0
        return (x.new_empty((*x.shape[:-2], x.size(-1), x.size(-2) // 2), dtype=torch.float4_e2m1fn_x2),
                x.new_empty((*x.shape[:-2], x.size(-1), x.size(-2) // 32), dtype=torch.float8_e8m0fnu))

    @custom_op("qutlass_amd::backward_t", mutates_args=(), schema="(Tensor x, Tensor h) -> (Tensor, Tensor)")
    def backward_t(x, h):
        o = _bt(x)
        amd.backward_t_bf16_(x, h, o[0], o[1])
        return o

    backward_t.register_fake(lambda x, h: _bt(x))

    def _bqt(c, s):
        return (c.new_empty((*c.shape[:-2], c.size(-1) * 2, c.size(-2) // 2), dtype=torch.float4_e2m1fn_x2),
                c.new_empty((*s.shape[:-2], s.size(-1) * 32, s.size(-2) // 32), dtype=torch.float8_e8m0fnu))

    @custom_op("qutlass_amd::backward_qt", mutates_args=(), schema="(Tensor x_e2m1, Tensor x_e8m0, Tensor h, Tensor alpha) -> (Tensor, Tensor)")
    def backward_qt(x_e2m1, x_e8m0, h, alpha):
        o = _bqt(x_e2m1, x_e8m0)
        amd.backward_qt_bf16_(x_e2m1, x_e8m0, h, alpha, o[0], o[1])
        return o

    backward_qt.register_fake(lambda x_e2m1, x_e8m0, h, alpha: _bqt(x_e2m1, x_e8m0))

    def _sq(x):
        m, n = x.shape
        mp = (m + 127) // 128 * 128
        return (x.new_empty((mp, n), dtype=torch.float8_e4m3fn), x.new_empty((mp, n // 32), dtype=torch.float8_e8m0fnu),
                x.new_empty((n, mp // 32), dtype=torch.float8_e8m0fnu))

    @custom_op("qutlass_amd::square_double_mxfp8", mutates_args=(), schema="(Tensor x_bf16) -> (Tensor, Tensor, Tensor)")
    def square_double_mxfp8(x_bf16):
        o = _sq(x_bf16)
        amd.backward_bf16_square_double_mxfp8_(x_bf16, o[0], o[1], o[2])
        return o

    square_double_mxfp8.register_fake(lambda x_bf16: _sq(x_bf16))

    def _tr(x, s):
        m, n = x.shape[0], x.shape[1] * 2
        mp = (m + 255) // 256 * 256
        return x.new_empty((n, mp), dtype=torch.float8_e4m3fn), x.new_empty((n, mp // 32), dtype=torch.float8_e8m0fnu)

    @custom_op("qutlass_amd::transpose_mxfp8", mutates_args=(), schema="(Tensor x_fp4, Tensor scales) -> (Tensor, Tensor)")
    def transpose_mxfp8(x_fp4, scales):
        o = _tr(x_fp4, scales)
        amd.mxfp4_transpose_mxfp8_(x_fp4, scales, o[0], o[1])
        return o

    transpose_mxfp8.register_fake(lambda x_fp4, scales: _tr(x_fp4, scales))


def to_blocked(input_matrix: torch.Tensor) -> torch.Tensor:
    """qutlass/utils.py:160-193 as a HIP kernel (csrc/to_blocked.hip.h) -> flat blocked byte vector, input dtype."""
    register_torch_ops()
    return torch.ops.qutlass_amd.to_blocked(input_matrix)
