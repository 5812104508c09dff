"""ctypes binding of the C ABI in include/qutlass_amd.h (libqutlass_amd.so, hand-written HIP for gfx950).

The library is built in-tree by ``qutlass_amd.build.build()`` (plain ``hipcc --offload-arch=gfx950``)
and must be present: there is NO CPU / eager fallback -- a missing or unloadable library raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqutlass_amd.so")

QAMD_OK, QAMD_ERR_INVALID, QAMD_ERR_HIP = 0, 1, 2
METHOD_QUEST, METHOD_ABSMAX = 0, 1

_vp, _i64, _i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
_GEMM_ARGS = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]

SYMBOLS = {
    "qutlass_amd_matmul_mxf4_bf16_tn": (_i32, _GEMM_ARGS),
    "qutlass_amd_matmul_nvf4_bf16_tn": (_i32, _GEMM_ARGS),
    "qutlass_amd_matmul_ada_mxf4_bf16_tn": (_i32, _GEMM_ARGS),
    "qutlass_amd_matmul_mxf8_bf16_tn": (_i32, _GEMM_ARGS),
    "qutlass_amd_matmul_mxf8_bf16_nn": (_i32, _GEMM_ARGS[:-1] + [_vp, _i64, _vp]),
    "qutlass_amd_mxf8_nn_workspace_bytes": (_i64, [_i64, _i64]),
    "qutlass_amd_mxf8_nn_workspace_bytes_for": (_i64, [_i64, _i64, _i64]),
    "qutlass_amd_matmul_mxf8_bf16_tn_fmt": (_i32, _GEMM_ARGS[:-1] + [_i32, _i32, _vp, _i64, _vp]),
    "qutlass_amd_matmul_mxf8_bf16_nn_fmt": (_i32, _GEMM_ARGS[:-1] + [_i32, _i32, _vp, _i64, _vp]),
    "qutlass_amd_gemm_splitk_workspace_bytes": (_i64, [_i32, _i64, _i64, _i64]),
    "qutlass_amd_matmul_mxf4_bf16_tn_ws": (_i32, _GEMM_ARGS[:-1] + [_vp, _i64, _vp]),
    "qutlass_amd_matmul_mxf8_bf16_tn_ws": (_i32, _GEMM_ARGS[:-1] + [_vp, _i64, _vp]),
    "qutlass_amd_nvf4_splitk_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "qutlass_amd_matmul_nvf4_bf16_tn_ws": (_i32, _GEMM_ARGS[:-1] + [_vp, _i64, _vp]),
    "qutlass_amd_fused_quantize_mx": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "qutlass_amd_fused_quantize_nv": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "qutlass_amd_fused_quantize_mx_blocked": (_i32, [_vp, _vp, _i32, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "qutlass_amd_fused_quantize_nv_blocked": (_i32, [_vp, _vp, _i32, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "qutlass_amd_fused_quantize_matmul_mxf4_bf16_tn": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "qutlass_amd_activation_path_launches": (_i32, [_i64, _i64, _i64, _i32]),
    "qutlass_amd_to_blocked": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "qutlass_amd_backward_t_bf16": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "qutlass_amd_backward_qt_bf16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "qutlass_amd_backward_bf16_square_double_mxfp8": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "qutlass_amd_mxfp4_transpose_mxfp8": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "qutlass_amd_backward_bf16_square_double_mxfp8_rows": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "qutlass_amd_mxfp4_transpose_mxfp8_rows": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "qutlass_amd_last_error": (ctypes.c_char_p, []),
    "qutlass_amd_version": (ctypes.c_char_p, []),
    "qutlass_amd_set_option": (_i32, [ctypes.c_char_p, _i32]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libqutlass_amd.so and declare every entry point of include/qutlass_amd.h."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or python qutlass_amd/build.py). "
                "qutlass_amd has no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    """Map the C return code to the reference's error convention (STD_TORCH_CHECK -> RuntimeError)."""
    if rc != QAMD_OK:
        msg = load().qutlass_amd_last_error().decode()
        raise RuntimeError(msg)


def set_option(key: str, value: int) -> int:
    return load().qutlass_amd_set_option(key.encode(), int(value))
