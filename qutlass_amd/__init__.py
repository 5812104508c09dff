"""qutlass_amd -- MI355X-native (gfx950 / CDNA4) implementation of the qutlass operator surface.

Public API = the hot-path functions of the reference's ``qutlass/__init__.py`` (same names, argument
meaning, defaults and Python-level error behaviour):

    fusedQuantizeMx, fusedQuantizeNv, matmul_mxf4_bf16_tn, matmul_nvf4_bf16_tn,
    matmul_mxf8_bf16_tn, matmul_mxf8_bf16_nn         (+ qutlass_amd.utils.to_blocked & friends)
    matmul_ada_mxf4_bf16_tn, backward_t_bf16, backward_qt_bf16, backward_bf16_square_double_mxfp8, mxfp4_transpose_mxfp8

All compute is hand-written HIP behind the C ABI of ``include/qutlass_amd.h``
(``libqutlass_amd.so``); importing this package loads that library and registers
``torch.ops._qutlass_C.*``.  There is no CPU or eager-PyTorch fallback: if the library is not built
the import fails, and every op requires GPU tensors.
"""
from __future__ import annotations

from typing import Literal

import torch

from . import _lib, ops
from .utils import ceil_div, get_padded_shape_mx, get_padded_shape_nv, pad_to_block, to_blocked  # noqa: F401

__version__ = "0.2.0"

_lib.load()                 # fail loudly at import time if the HIP library is missing
ops.register_torch_ops()    # torch.ops._qutlass_C.<op>  (reference: bindings.cpp:498-535)

qutlass_CUDA = torch.ops._qutlass_C
_ops_amd = torch.ops.qutlass_amd   # extensions + the mutation-declaring twins of the reference's output-filling ops
_METHOD_CODE = {"quest": 0, "abs_max": 1}

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


def matmul_ada_mxf4_bf16_tn(a: torch.Tensor, b: torch.Tensor, a_sf: torch.Tensor, b_sf: torch.Tensor,
                            alpha: torch.Tensor) -> torch.Tensor:
    """qutlass/__init__.py:79-86: small-batch MXFP4 GEMM taking the UN-swizzled (rows, K/32) scales."""
    return qutlass_CUDA.matmul_ada_mxf4_bf16_tn(a, b, a_sf, b_sf, alpha)


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


def _alloc_mx(a: torch.Tensor, blocked: bool = False):
    padded_rows, padded_cols = get_padded_shape_mx(a)
    xh_e2m1 = torch.empty(*a.shape[:-1], a.size(-1) // 2, dtype=torch.uint8, device=a.device)
    sf_shape = (padded_rows * padded_cols,) if blocked else (padded_rows, padded_cols)
    return xh_e2m1, torch.empty(*sf_shape, dtype=torch.float8_e8m0fnu, device=a.device)


def _alloc_nv(a: torch.Tensor, blocked: bool = False):
    padded_rows, padded_cols = get_padded_shape_nv(a)
    xh_e2m1 = torch.empty(*a.shape[:-1], a.size(-1) // 2, dtype=torch.uint8, device=a.device)
    sf_shape = (padded_rows * padded_cols,) if blocked else (padded_rows, padded_cols)
    return xh_e2m1, torch.empty(*sf_shape, dtype=torch.float8_e4m3fn, device=a.device)


def fusedQuantizeMx(a: torch.Tensor, b: torch.Tensor, *, method: Literal["quest", "abs_max"] = "quest",
                    return_mask: bool = False):
    """qutlass/__init__.py:149-180: allocate packed e2m1 + (padded_rows, padded_cols) e8m0 [+ clip mask]
    and run the fused rotate+quantize kernel.  As in the reference the scale buffer is written flat
    (first numel/32 bytes) and its padding is left uninitialised.
    (The op calls go through `qutlass_amd::` twins of `_qutlass_C.fusedQuantizeMx*` -- same kernels, same checks: the reference's schemas hide the writes from
    torch.compile.  Eager: the in-place twin on tensors allocated here; under torch.compile: the functional form, see ops.py.)"""
    if method not in _METHOD_CODE:
        raise ValueError(f"invalid method {method!r}, must be 'quest' or 'abs_max'")
    if return_mask and method != "quest":
        raise ValueError("return_mask is only supported for method 'quest'")
    if torch.compiler.is_compiling():
        return _ops_amd.quantize_mx_mask(a, b) if return_mask else _ops_amd.quantize_mx(a, b, _METHOD_CODE[method])
    xh_e2m1, xh_e8m0 = _alloc_mx(a)
    if return_mask:
        clip_mask = torch.empty(*a.shape[:-1], a.size(-1) // 8, dtype=torch.uint8, device=a.device)
        _ops_amd.fusedQuantizeMxMask_(a, b, xh_e2m1, xh_e8m0, clip_mask)
        return xh_e2m1, xh_e8m0, clip_mask
    _ops_amd.fusedQuantizeMx_(a, b, xh_e2m1, xh_e8m0, _METHOD_CODE[method])
    return xh_e2m1, xh_e8m0


def fusedQuantizeNv(a: torch.Tensor, b: torch.Tensor, global_scale: torch.Tensor, *,
                    method: Literal["quest", "abs_max"] = "abs_max") -> tuple[torch.Tensor, torch.Tensor]:
    """qutlass/__init__.py:183-203."""
    if method not in _METHOD_CODE:
        raise ValueError(f"invalid method {method!r}, must be 'quest' or 'abs_max'")
    if torch.compiler.is_compiling():
        return _ops_amd.quantize_nv(a, b, global_scale, _METHOD_CODE[method])
    xh_e2m1, xh_e4m3 = _alloc_nv(a)
    _ops_amd.fusedQuantizeNv_(a, b, xh_e2m1, xh_e4m3, global_scale, _METHOD_CODE[method])
    return xh_e2m1, xh_e4m3


def fusedQuantizeMxBlocked(a: torch.Tensor, b: torch.Tensor, *, method: Literal["quest", "abs_max"] = "quest") -> tuple[torch.Tensor, torch.Tensor]:
    """EXTENSION (no reference counterpart): ``fusedQuantizeMx`` whose scales come out GEMM-ready -- the second tensor is
    byte for byte ``to_blocked(fusedQuantizeMx(a, b, method=method)[1])`` (flat, zero padded), written by the quantizer itself:
    one launch instead of two on the activation path (qutlass/__init__.py:149-180 + qutlass/utils.py:160-193)."""
    if method not in _METHOD_CODE:
        raise ValueError(f"invalid method {method!r}, must be 'quest' or 'abs_max'")
    if torch.compiler.is_compiling():
        return _ops_amd.quantize_mx_blocked(a, b, _METHOD_CODE[method])
    xh_e2m1, xh_e8m0 = _alloc_mx(a, blocked=True)
    _ops_amd.fusedQuantizeMxBlocked(a, b, xh_e2m1, xh_e8m0, _METHOD_CODE[method])
    return xh_e2m1, xh_e8m0


def fusedQuantizeNvBlocked(a: torch.Tensor, b: torch.Tensor, global_scale: torch.Tensor, *,
                           method: Literal["quest", "abs_max"] = "abs_max") -> tuple[torch.Tensor, torch.Tensor]:
    """EXTENSION: ``fusedQuantizeNv`` with the e4m3 scales written directly in the ``to_blocked`` layout (see fusedQuantizeMxBlocked)."""
    if method not in _METHOD_CODE:
        raise ValueError(f"invalid method {method!r}, must be 'quest' or 'abs_max'")
    if torch.compiler.is_compiling():
        return _ops_amd.quantize_nv_blocked(a, b, global_scale, _METHOD_CODE[method])
    xh_e2m1, xh_e4m3 = _alloc_nv(a, blocked=True)
    _ops_amd.fusedQuantizeNvBlocked(a, b, xh_e2m1, xh_e4m3, global_scale, _METHOD_CODE[method])
    return xh_e2m1, xh_e4m3


def _decode_single_launch_wins(m: int, n: int, k: int, rot: int, device: torch.device | None = None) -> bool:
    """The measured one-launch / two-launch rule of the activation path.  [r4] It lives in the C library now
    (``qutlass_amd_activation_path_launches``, csrc/capi.hip: thresholds, measurements and the CU-count scaling are documented there), so
    that a caller of the C ABI gets the same rule; this is the Python face of it.  The rule scales with the CU count of HIP's CURRENT device, so it is asked
    with the operand's device current (a mixed or partitioned node would otherwise be judged by device 0)."""
    ask = lambda: _lib.load().qutlass_amd_activation_path_launches(int(m), int(n), int(k), int(rot)) == 1
    if device is not None and device.type == "cuda" and torch.cuda.is_available():
        with torch.cuda.device(device):
            return ask()
    return ask()


def fused_quantize_matmul_mxf4_bf16_tn(x: torch.Tensor, h: torch.Tensor, b: torch.Tensor, b_sf: torch.Tensor, alpha: torch.Tensor, *,
                                       method: Literal["quest", "abs_max"] = "quest", single_launch: bool | None = None) -> torch.Tensor:
    """EXTENSION: ``matmul_mxf4_bf16_tn(*fusedQuantizeMx(x, h, method=method) -> to_blocked, b, b_sf, alpha)`` -- the activation path of
    one linear layer (qutlass/__init__.py:149-180 -> qutlass/utils.py:160-193 -> qutlass/__init__.py:34-76) in fewer launches, same bits:
      * TWO launches: the quantizer writes GEMM-ready scales (``fusedQuantizeMxBlocked``), then the GEMM;
      * ONE launch for decode batches (csrc/gemm_mx_fusedq.hip.h: the small-batch GEMM rotates and quantises its own A operand),
        chosen where it measured faster (``qutlass_amd_activation_path_launches`` in the C library, calibrated on a 256-CU MI355X and scaled
        with the CU count); ``single_launch=True / False`` forces either.
    ``method`` defaults to ``"quest"`` like ``fusedQuantizeMx`` (qutlass/__init__.py:149): swapping the composed calls for this helper keeps the quantizer."""
    if method not in _METHOD_CODE:
        raise ValueError(f"invalid method {method!r}, must be 'quest' or 'abs_max'")
    k = x.size(-1)
    m = x.numel() // k if k else 0
    if single_launch is None:
        single_launch = _decode_single_launch_wins(m, b.size(0), k, h.size(0), x.device)
    if single_launch:
        out = torch.ops.qutlass_amd.fusedQuantizeMatmulMxf4(x, h, b, b_sf, alpha, _METHOD_CODE[method])
        return out.view(*x.shape[:-1], b.size(0))
    a_q, a_sf = fusedQuantizeMxBlocked(x, h, method=method)
    return qutlass_CUDA.matmul_mxf4_bf16_tn(a_q.view(-1, k // 2), b, a_sf, b_sf, alpha).view(*x.shape[:-1], b.size(0))


def backward_t_bf16(x: torch.Tensor, h: torch.Tensor, xh_e2m1: torch.Tensor = None,
                    xh_e8m0: torch.Tensor = None) -> tuple[torch.Tensor, torch.Tensor]:
    """qutlass/__init__.py:206-243: abs-max MXFP4 of x^T (last two dims swapped) rotated per 32 along the old
    second-to-last dim.  Outputs (.., M, N/2) float4_e2m1fn_x2 and (.., M, N/32) e8m0 for x of shape (.., N, M)."""
    if xh_e2m1 is None and xh_e8m0 is None and torch.compiler.is_compiling():
        assert x.dtype == h.dtype == torch.bfloat16 and x.is_contiguous() and h.is_contiguous()
        return _ops_amd.backward_t(x, h)
    if xh_e2m1 is None:
        xh_e2m1 = torch.empty(*x.shape[:-2], x.size(-1), x.size(-2) // 2, dtype=torch.float4_e2m1fn_x2, device=h.device)
    if xh_e8m0 is None:
        xh_e8m0 = torch.empty(*x.shape[:-2], x.size(-1), x.size(-2) // 32, dtype=torch.float8_e8m0fnu, device=h.device)
    assert x.dtype == h.dtype == torch.bfloat16
    assert xh_e2m1.dtype == torch.float4_e2m1fn_x2 and xh_e8m0.dtype == torch.float8_e8m0fnu
    assert x.is_contiguous() and h.is_contiguous() and xh_e2m1.is_contiguous() and xh_e8m0.is_contiguous()
    _ops_amd.backward_t_bf16_(x, h, xh_e2m1, xh_e8m0)
    return xh_e2m1, xh_e8m0


def backward_qt_bf16(x_e2m1: torch.Tensor, x_e8m0: torch.Tensor, h: torch.Tensor, alpha: torch.Tensor,
                     xh_e2m1: torch.Tensor = None, xh_e8m0: torch.Tensor = None) -> tuple[torch.Tensor, torch.Tensor]:
    """qutlass/__init__.py:246-286: the same on an MXFP4 operand (x_e2m1 (.., N, M/2), x_e8m0 (.., N, M/32))."""
    if xh_e2m1 is None and xh_e8m0 is None and torch.compiler.is_compiling():
        assert x_e2m1.is_contiguous() and x_e8m0.is_contiguous() and h.is_contiguous()
        return _ops_amd.backward_qt(x_e2m1, x_e8m0, h, alpha)
    if xh_e2m1 is None:
        xh_e2m1 = torch.empty(*x_e2m1.shape[:-2], x_e2m1.size(-1) * 2, x_e2m1.size(-2) // 2,
                              dtype=torch.float4_e2m1fn_x2, device=h.device)
    if xh_e8m0 is None:
        xh_e8m0 = torch.empty(*x_e8m0.shape[:-2], x_e8m0.size(-1) * 32, x_e8m0.size(-2) // 32,
                              dtype=torch.float8_e8m0fnu, device=h.device)
    assert (x_e2m1.is_contiguous() and x_e8m0.is_contiguous() and h.is_contiguous()
            and xh_e2m1.is_contiguous() and xh_e8m0.is_contiguous())
    _ops_amd.backward_qt_bf16_(x_e2m1, x_e8m0, h, alpha, xh_e2m1, xh_e8m0)
    return xh_e2m1, xh_e8m0


def backward_bf16_square_double_mxfp8(x_bf16: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """qutlass/__init__.py:288-297: e4m3 with one e8m0 per 32 x 32 block, returned row-wise (m_pad, n/32) and column-wise
    (n, m_pad/32), m_pad = rows rounded up to 128.  The missing rows are zeros INSIDE the kernel: no padded copy of x."""
    if torch.compiler.is_compiling():
        return _ops_amd.square_double_mxfp8(x_bf16)
    m, n = x_bf16.shape
    m_pad = ceil_div(m, 128) * 128
    x_fp8 = torch.empty(m_pad, n, device=x_bf16.device, dtype=torch.float8_e4m3fn)
    row_scales = torch.empty(m_pad, n // 32, device=x_bf16.device, dtype=torch.float8_e8m0fnu)
    column_scales = torch.empty(n, m_pad // 32, device=x_bf16.device, dtype=torch.float8_e8m0fnu)
    _ops_amd.backward_bf16_square_double_mxfp8_(x_bf16, x_fp8, row_scales, column_scales)
    return x_fp8, row_scales, column_scales


def mxfp4_transpose_mxfp8(x_fp4: torch.Tensor, scales: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """qutlass/__init__.py:299-315: MXFP4 (m, n/2) + e8m0 (m, n/32) -> transposed e4m3 (n, m_pad) + e8m0 (n, m_pad/32), m_pad = rows
    rounded up to 256 as in the reference.  The padding rows (zero codes, unit scales) exist only inside the kernel: x_fp4 is
    not copied and `scales` is not written (the reference's own "TODO: padding in kernel")."""
    if torch.compiler.is_compiling():
        return _ops_amd.transpose_mxfp8(x_fp4, scales)
    m = x_fp4.shape[0]
    m_pad = ceil_div(m, 256) * 256
    x_fp8 = torch.empty(x_fp4.shape[1] * 2, m_pad, device=x_fp4.device, dtype=torch.float8_e4m3fn)
    shared_exps = torch.empty(x_fp4.shape[1] * 2, m_pad // 32, device=x_fp4.device, dtype=torch.float8_e8m0fnu)
    _ops_amd.mxfp4_transpose_mxfp8_(x_fp4, scales, x_fp8, shared_exps)
    return x_fp8, shared_exps


_OUT_OF_SCOPE = ()


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise AttributeError(
            f"qutlass_amd does not provide {name!r} yet: it is outside the hot path this build covers "
            "(SURVEY.md section 8f lists it as 'next')."
        )
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
