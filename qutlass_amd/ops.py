"""Operator layer: the 10 hot-path ops of ``torch.ops._qutlass_C`` re-implemented over the C ABI.

Each function mirrors one op of the reference's binding file ``qutlass/csrc/bindings.cpp`` (same
argument order, same validation, same error text and ``RuntimeError`` convention, same ownership:
GEMM ops allocate and return the bf16 output, quantize ops write into caller-allocated outputs and
return them).  The compute is the hand-written HIP in ``csrc/`` reached through ``_lib`` -- there is
no fallback path; without the built library every op raises.
"""
from __future__ import annotations

import torch

from . import _lib

_E8M0 = torch.float8_e8m0fnu
_E4M3 = torch.float8_e4m3fn


def _check(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


def _arg(pos: int, name: str) -> str:
    return f"argument #{pos} '{name}'"


def _check_all_contiguous(op, args):  # include/bindings_utils.h:67-80
    for pos, (t, name) in enumerate(args):
        _check(t.is_contiguous(),
               f"Expected contiguous tensor, but got non-contiguous tensor for {_arg(pos, name)} "
               f"(while checking arguments for {op})")


def _check_device_type_cuda(op, tensors):  # include/bindings_utils.h:82-95
    for t in tensors:
        _check(t.device.type == "cuda",
               f"Expected tensor to have cuda DeviceType, but got tensor with {t.device.type} "
               f"DeviceType (while checking arguments for {op})")


def _check_all_same_gpu(op, args):  # include/bindings_utils.h:97-136
    t0, n0 = args[0]
    for pos, (t, name) in enumerate(args[1:], start=1):
        _check(t.device == t0.device,
               f"Expected tensor for {_arg(0, n0)} to have the same device as tensor for {_arg(pos, name)}; "
               f"but device {t0.device} does not equal {t.device} (while checking arguments for {op})")


def _stream(t: torch.Tensor) -> int:
    # launch on torch's current stream of the tensor's device (reference: common.h:40-45)
    return torch.cuda.current_stream(t.device).cuda_stream


def _gemm(op, fn_name, A, B, A_sf, B_sf, alpha, data_dtype, data_msg, sf_dtype, sf_msg, kmin, nn=False,
          contiguous_alpha=False):
    ctg = [(A, "A"), (B, "B"), (A_sf, "A_sf"), (B_sf, "B_sf")]
    if contiguous_alpha:
        ctg.append((alpha, "alpha"))
    _check_all_contiguous(op, ctg)
    _check_device_type_cuda(op, [A, B, A_sf, B_sf, alpha])
    _check_all_same_gpu(op, [(A, "A"), (B, "B"), (A_sf, "A_sf"), (B_sf, "B_sf"), (alpha, "alpha")])
    _check(A.dtype == data_dtype, f"A must be {data_msg}")
    _check(B.dtype == data_dtype, f"B must be {data_msg}")
    _check(A_sf.dtype == sf_dtype, f"A_sf must be {sf_msg}")
    _check(B_sf.dtype == sf_dtype, f"B_sf must be {sf_msg}")
    _check(A.dim() == 2 and B.dim() == 2, "A and B must be 2D")
    if nn:
        _check(A.size(0) == B.size(1), "Inner dimensions must match for A.T @ B.T")
        _check(A.size(0) >= kmin, f"A K-dim must be >= {kmin}")
        M = A.size(1)
    else:
        _check(A.size(1) == B.size(1), "Inner dimensions must match for A @ B.T")
        _check(A.size(1) >= kmin, f"A K-dim must be >= {kmin}")
        M = A.size(0)
    _check(B.size(1) >= kmin, f"B K-dim must be >= {kmin}")
    N = B.size(0)
    K = B.size(1) * (2 if data_dtype == torch.uint8 else 1)
    out = A.new_empty((M, N), dtype=torch.bfloat16)
    lib = _lib.load()
    with torch.cuda.device(A.device):
        if nn:
            # scratch for the (K, M) -> (M, K) re-layout, from torch's stream-ordered caching allocator
            ws_bytes = lib.qutlass_amd_mxf8_nn_workspace_bytes(M, K)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device)
            rc = getattr(lib, fn_name)(A.data_ptr(), B.data_ptr(), A_sf.data_ptr(), B_sf.data_ptr(),
                                       alpha.data_ptr(), out.data_ptr(), M, N, K, ws.data_ptr(), ws_bytes,
                                       _stream(A))
        else:
            rc = getattr(lib, fn_name)(A.data_ptr(), B.data_ptr(), A_sf.data_ptr(), B_sf.data_ptr(),
                                       alpha.data_ptr(), out.data_ptr(), M, N, K, _stream(A))
    _lib.check(rc)
    return out


def matmul_mxf4_bf16_tn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:32-66 -> qutlass_amd_matmul_mxf4_bf16_tn."""
    return _gemm("matmul_mxf4_bf16_tn", "qutlass_amd_matmul_mxf4_bf16_tn", A, B, A_sf, B_sf, alpha,
                 torch.uint8, "uint8", _E8M0, "float8_e8m0fnu", 32)


def matmul_nvf4_bf16_tn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:68-102 -> qutlass_amd_matmul_nvf4_bf16_tn."""
    return _gemm("matmul_nvf4_bf16_tn", "qutlass_amd_matmul_nvf4_bf16_tn", A, B, A_sf, B_sf, alpha,
                 torch.uint8, "uint8", _E4M3, "float8_e4m3fn", 16)


def matmul_mxf8_bf16_tn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:140-177 -> qutlass_amd_matmul_mxf8_bf16_tn."""
    return _gemm("matmul_mxf8_bf16_tn", "qutlass_amd_matmul_mxf8_bf16_tn", A, B, A_sf, B_sf, alpha,
                 _E4M3, "float8_e4m3fn", _E8M0, "float8_e8m0fnu", 32, contiguous_alpha=True)


def matmul_mxf8_bf16_nn(A, B, A_sf, B_sf, alpha):
    """bindings.cpp:179-216 -> qutlass_amd_matmul_mxf8_bf16_nn (A stored (K, M))."""
    return _gemm("matmul_mxf8_bf16_nn", "qutlass_amd_matmul_mxf8_bf16_nn", A, B, A_sf, B_sf, alpha,
                 _E4M3, "float8_e4m3fn", _E8M0, "float8_e8m0fnu", 32, nn=True, contiguous_alpha=True)


def _quant_common(op, A, B, outs, extra_dev=()):
    names = ["OUT", "OUT_sf", "OUT_mask"]
    args = [(A, "A"), (B, "B")] + [(o, names[i]) for i, o in enumerate(outs)]
    _check_all_contiguous(op, args)
    _check_device_type_cuda(op, [A, B, *outs, *[t for t, _ in extra_dev]])
    _check_all_same_gpu(op, args + list(extra_dev))
    _check(A.dtype == torch.bfloat16, "A must be bf16")
    _check(B.dtype == torch.bfloat16, "B must be bf16")


def _quant_mx(op, A, B, OUT, OUT_sf, OUT_mask, method, allowed):
    outs = [OUT, OUT_sf] + ([OUT_mask] if OUT_mask is not None else [])
    _quant_common(op, A, B, outs)
    _check(B.dim() == 2 and B.size(0) == B.size(1), "Rotation matrix must be square")
    rot = B.size(0)
    _check(A.numel() % rot == 0, f"A must be divisible by{rot}")
    _check(rot in allowed, f"Unsupported rotation size {rot}; expected {_fmt_allowed(allowed)}.")
    # the C ABI writes numel/2, numel/32 (and numel/8) bytes: make sure the caller's buffers hold them
    _check(OUT.numel() * OUT.element_size() >= A.numel() // 2, "OUT is too small")
    _check(OUT_sf.numel() * OUT_sf.element_size() >= A.numel() // 32, "OUT_sf is too small")
    if OUT_mask is not None:
        _check(OUT_mask.numel() * OUT_mask.element_size() >= A.numel() // 8, "OUT_mask is too small")
    lib = _lib.load()
    with torch.cuda.device(A.device):
        rc = lib.qutlass_amd_fused_quantize_mx(A.data_ptr(), B.data_ptr(), rot, A.numel(), method,
                                               OUT.data_ptr(), OUT_sf.data_ptr(),
                                               OUT_mask.data_ptr() if OUT_mask is not None else None,
                                               _stream(A))
    _lib.check(rc)


def _fmt_allowed(allowed):
    a = [str(x) for x in allowed]
    return a[0] if len(a) == 1 else ", ".join(a[:-1]) + ", or " + a[-1]


def fusedQuantizeMxQuest(A, B, OUT, OUT_sf):
    """bindings.cpp:218-252."""
    _quant_mx("fusedQuantizeMxQuest", A, B, OUT, OUT_sf, None, _lib.METHOD_QUEST, (32, 64, 128))
    return OUT, OUT_sf


def fusedQuantizeMxAbsMax(A, B, OUT, OUT_sf):
    """bindings.cpp:292-333."""
    _quant_mx("fusedQuantizeMxAbsMax", A, B, OUT, OUT_sf, None, _lib.METHOD_ABSMAX, (32, 64, 128))
    return OUT, OUT_sf


def fusedQuantizeMxQuestWithMask(A, B, OUT, OUT_sf, OUT_mask):
    """bindings.cpp:255-289 (rotation size 32 only)."""
    _quant_mx("fusedQuantizeMxQuestWithMask", A, B, OUT, OUT_sf, OUT_mask, _lib.METHOD_QUEST, (32,))
    return OUT, OUT_sf, OUT_mask


def _quant_nv(op, A, B, OUT, OUT_sf, global_scale, method):
    _quant_common(op, A, B, [OUT, OUT_sf], extra_dev=[(global_scale, "global_scale")])
    _check(global_scale.dtype == torch.float32, "global_scale must be float")
    _check(global_scale.dim() == 1 and global_scale.size(0) == 1, "global_scale must be a scalar")
    _check(B.dim() == 2 and B.size(0) == B.size(1), "Rotation matrix must be square")
    rot = B.size(0)
    _check(A.numel() % rot == 0, f"A must be divisible by{rot}")
    _check(rot in (16, 32, 64, 128), f"Unsupported rotation size {rot}; expected 16, 32, 64, or 128.")
    _check(OUT.numel() * OUT.element_size() >= A.numel() // 2, "OUT is too small")
    _check(OUT_sf.numel() * OUT_sf.element_size() >= A.numel() // 16, "OUT_sf is too small")
    lib = _lib.load()
    with torch.cuda.device(A.device):
        rc = lib.qutlass_amd_fused_quantize_nv(A.data_ptr(), B.data_ptr(), rot, A.numel(), method,
                                               global_scale.data_ptr(), OUT.data_ptr(), OUT_sf.data_ptr(),
                                               _stream(A))
    _lib.check(rc)


def fusedQuantizeNvQuest(A, B, OUT, OUT_sf, global_scale):
    """bindings.cpp:335-378."""
    _quant_nv("fusedQuantizeNvQuest", A, B, OUT, OUT_sf, global_scale, _lib.METHOD_QUEST)
    return OUT, OUT_sf


def fusedQuantizeNvAbsMax(A, B, OUT, OUT_sf, global_scale):
    """bindings.cpp:380-426."""
    _quant_nv("fusedQuantizeNvAbsMax", A, B, OUT, OUT_sf, global_scale, _lib.METHOD_ABSMAX)
    return OUT, OUT_sf


def to_blocked(input_matrix: torch.Tensor) -> torch.Tensor:
    """Block-scale swizzle as a device op (replaces the torch/Triton paths of qutlass/utils.py:160-193)."""
    _check(input_matrix.dim() == 2, "to_blocked expects a 2-D matrix")
    _check(input_matrix.element_size() == 1, "Expected element size to be 1 byte (8 bits)")
    _check(input_matrix.is_contiguous(), "Input tensor must be contiguous")
    _check(input_matrix.device.type == "cuda", "to_blocked: expected a GPU tensor (no CPU path in qutlass_amd)")
    rows, cols = input_matrix.shape
    pr, pc = -(-rows // 128) * 128, -(-cols // 4) * 4
    out = input_matrix.new_empty(pr * pc)
    lib = _lib.load()
    with torch.cuda.device(input_matrix.device):
        rc = lib.qutlass_amd_to_blocked(input_matrix.data_ptr(), rows, cols, out.data_ptr(), _stream(input_matrix))
    _lib.check(rc)
    return out


SCHEMAS = {
    # exact schema strings of bindings.cpp:499-513 for the ops this build provides
    "matmul_mxf4_bf16_tn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "matmul_nvf4_bf16_tn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "matmul_mxf8_bf16_tn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "matmul_mxf8_bf16_nn": "(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor",
    "fusedQuantizeMxQuest": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf) -> (Tensor, Tensor)",
    "fusedQuantizeMxAbsMax": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf) -> (Tensor, Tensor)",
    "fusedQuantizeMxQuestWithMask": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor OUT_mask) -> (Tensor, Tensor, Tensor)",
    "fusedQuantizeNvQuest": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor global_scale) -> (Tensor, Tensor)",
    "fusedQuantizeNvAbsMax": "(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor global_scale) -> (Tensor, Tensor)",
}

_registered = False


def register_torch_ops() -> None:
    """Register the ops under ``torch.ops._qutlass_C`` (dispatch key CUDA = HIP tensors on ROCm) so
    callers written against the reference (``torch.ops._qutlass_C.<op>``) work unchanged."""
    global _registered
    if _registered:
        return
    lib_def = torch.library.Library("_qutlass_C", "FRAGMENT")
    g = globals()
    for name, schema in SCHEMAS.items():
        lib_def.define(name + schema)
        lib_def.impl(name, g[name], "CUDA")
    g["_torch_library"] = lib_def  # keep alive
    _registered = True
