"""Test-side binding of the LAB library (qutlass_amd/libqutlass_amd_bench.so).

The product library (libqutlass_amd.so, what `import qutlass_amd` loads) has no kernel-selecting options: a shape gets
the kernel its dispatch rules pick and nothing a caller does can change that.  Parity tests that must force a tile or a
schedule ("this variant too is bit-identical to the oracle") therefore go through the lab build of the same sources,
which keeps every schedule variant and the `gemm_variant` / `nvf4_variant` / `pp_flags` switches.  This module calls its
C ABI (include/qutlass_amd.h) directly with torch tensors' device pointers on the current stream -- the same calls
csrc/torch_ext.cpp makes for the product.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QAMD_LAB_LIB") or os.path.join(os.path.dirname(_HERE), "qutlass_amd", "libqutlass_amd_bench.so")   # (QAMD_LAB_LIB: a side build of the lab library, tools/build_variant.py --lab)

_vp, _i64, _i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
_GEMM = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]
_SIGS = {
    "qutlass_amd_matmul_nvf4_bf16_tn": (_i32, _GEMM),
    "qutlass_amd_matmul_ada_mxf4_bf16_tn": (_i32, _GEMM),
    "qutlass_amd_matmul_mxf8_bf16_nn": (_i32, _GEMM[:-1] + [_vp, _i64, _vp]),
    "qutlass_amd_mxf8_nn_workspace_bytes": (_i64, [_i64, _i64]),
    "qutlass_amd_gemm_splitk_workspace_bytes": (_i64, [_i32, _i64, _i64, _i64]),
    "qutlass_amd_matmul_mxf4_bf16_tn_ws": (_i32, _GEMM[:-1] + [_vp, _i64, _vp]),
    "qutlass_amd_matmul_mxf8_bf16_tn_ws": (_i32, _GEMM[:-1] + [_vp, _i64, _vp]),
    "qutlass_amd_matmul_mxf8_bf16_tn_fmt": (_i32, _GEMM[:-1] + [_i32, _i32, _vp, _i64, _vp]),
    "qutlass_amd_nvf4_splitk_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "qutlass_amd_matmul_nvf4_bf16_tn_ws": (_i32, _GEMM[:-1] + [_vp, _i64, _vp]),
    "qutlass_amd_mxfp4_transpose_mxfp8": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "qutlass_amd_backward_bf16_square_double_mxfp8": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "qutlass_amd_mxfp4_transpose_mxfp8_rows": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "qutlass_amd_fused_quantize_mx": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "qutlass_amd_backward_t_bf16": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "qutlass_amd_backward_qt_bf16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "qutlass_amd_fused_quantize_matmul_mxf4_bf16_tn": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "qutlass_amd_last_error": (ctypes.c_char_p, []),
    "qutlass_amd_version": (ctypes.c_char_p, []),
    "qutlass_amd_set_option": (_i32, [ctypes.c_char_p, _i32]),
    "qutlass_amd_debug_set_trace_buffer": (None, [_vp]),
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (qutlass_amd/build.py build_bench_lib)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def set_option(key: str, value: int) -> int:
    return load().qutlass_amd_set_option(key.encode(), int(value))


@contextlib.contextmanager
def forced(**options):
    """`with lab.forced(gemm_variant=30): ...` -- options are restored on exit, also when the body raises."""
    old = {k: set_option(k, v) for k, v in options.items()}
    try:
        yield
    finally:
        for k, v in old.items():
            set_option(k, v)


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(load().qutlass_amd_last_error().decode())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor) -> int:
    return t.data_ptr()


def _gemm(entry: str, ebits: int, a, b, a_sf, b_sf, alpha, fp8: bool):
    lib = load()
    m, n, k = a.shape[0], b.shape[0], b.shape[1] * (1 if fp8 else 2)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    ws_bytes = lib.qutlass_amd_gemm_splitk_workspace_bytes(ebits, m, n, k)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=a.device)
    _check(getattr(lib, entry)(_p(a), _p(b), _p(a_sf), _p(b_sf), _p(alpha), _p(out), m, n, k, _p(ws) if ws_bytes else None, ws_bytes, _stream()))
    return out


def matmul_mxf4_bf16_tn(a, b, a_sf, b_sf, alpha):
    return _gemm("qutlass_amd_matmul_mxf4_bf16_tn_ws", 4, a, b, a_sf, b_sf, alpha, False)


def matmul_mxf8_bf16_tn(a, b, a_sf, b_sf, alpha):
    return _gemm("qutlass_amd_matmul_mxf8_bf16_tn_ws", 8, a, b, a_sf, b_sf, alpha, True)


def matmul_mxf8_bf16_tn_fmt(a, b, a_sf, b_sf, alpha, a_format: int = 1):
    """the extension entry: A in e5m2 (a_format = 1) or e4m3 (0), B e4m3; no scratch handed over (single-pass plans only)"""
    m, n, k = a.shape[0], b.shape[0], b.shape[1]
    out = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    _check(load().qutlass_amd_matmul_mxf8_bf16_tn_fmt(_p(a), _p(b), _p(a_sf), _p(b_sf), _p(alpha), _p(out), m, n, k, a_format, 0, None, 0, _stream()))
    return out


def matmul_mxf8_bf16_nn(a, b, a_sf, b_sf, alpha):
    lib = load()
    k, m, n = a.shape[0], a.shape[1], b.shape[0]
    out = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    ws_bytes = lib.qutlass_amd_mxf8_nn_workspace_bytes(m, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
    _check(lib.qutlass_amd_matmul_mxf8_bf16_nn(_p(a), _p(b), _p(a_sf), _p(b_sf), _p(alpha), _p(out), m, n, k, _p(ws), ws_bytes, _stream()))
    return out


def _plain(entry: str, a, b, a_sf, b_sf, alpha):
    m, n, k = a.shape[0], b.shape[0], b.shape[1] * 2
    out = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    _check(getattr(load(), entry)(_p(a), _p(b), _p(a_sf), _p(b_sf), _p(alpha), _p(out), m, n, k, _stream()))
    return out


def matmul_nvf4_bf16_tn(a, b, a_sf, b_sf, alpha):
    lib = load()
    m, n, k = a.shape[0], b.shape[0], b.shape[1] * 2
    out = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    ws_bytes = lib.qutlass_amd_nvf4_splitk_workspace_bytes(m, n, k)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=a.device)
    _check(lib.qutlass_amd_matmul_nvf4_bf16_tn_ws(_p(a), _p(b), _p(a_sf), _p(b_sf), _p(alpha), _p(out), m, n, k, _p(ws) if ws_bytes else None, ws_bytes, _stream()))
    return out


def matmul_ada_mxf4_bf16_tn(a, b, a_sf, b_sf, alpha):
    return _plain("qutlass_amd_matmul_ada_mxf4_bf16_tn", a, b, a_sf, b_sf, alpha)


def mxfp4_transpose_mxfp8(x_fp4, scales, m: int, n: int):
    """C-ABI call of the lab build (the Python surface's row padding is not repeated here: m % 128 == 0, n % 256 == 0)."""
    y = torch.empty(n, m, dtype=torch.uint8, device=x_fp4.device)
    sf = torch.empty(n, m // 32, dtype=torch.uint8, device=x_fp4.device)
    _check(load().qutlass_amd_mxfp4_transpose_mxfp8(_p(x_fp4), _p(scales), m, n, _p(y), _p(sf), _stream()))
    return y, sf


def backward_bf16_square_double_mxfp8(x):
    """C-ABI call of the lab build (x: (m, n) bf16, m % 128 == 0, n % 128 == 0): e4m3 (m, n), row scales (m, n/32), column scales (n, m/32)."""
    m, n = x.shape
    y = torch.empty(m, n, dtype=torch.uint8, device=x.device)
    rs = torch.empty(m, n // 32, dtype=torch.uint8, device=x.device)
    cs = torch.empty(n, m // 32, dtype=torch.uint8, device=x.device)
    _check(load().qutlass_amd_backward_bf16_square_double_mxfp8(_p(x), m, n, _p(y), _p(rs), _p(cs), _stream()))
    return y, rs, cs


def mxfp4_transpose_mxfp8_rows(x_fp4, scales, m: int, m_pad: int, n: int):
    """C-ABI call of the lab build, row-padded form: rows m .. m_pad-1 count as zero codes (m_pad % 128 == 0, n % 256 == 0)."""
    y = torch.empty(n, m_pad, dtype=torch.uint8, device=x_fp4.device)
    sf = torch.empty(n, m_pad // 32, dtype=torch.uint8, device=x_fp4.device)
    _check(load().qutlass_amd_mxfp4_transpose_mxfp8_rows(_p(x_fp4), _p(scales), m, m_pad, n, _p(y), _p(sf), _stream()))
    return y, sf


def fused_quantize_mx(x, h, method: str = "quest", return_mask: bool = False):
    """C-ABI call of the LAB build's fused rotate + quantize (the product op of the same name has no encoder switch): allocations as
    qutlass_amd.fusedQuantizeMx (qutlass/__init__.py:149-180)."""
    from qutlass_amd.utils import get_padded_shape_mx

    rows, cols = get_padded_shape_mx(x)
    e2m1 = torch.empty(*x.shape[:-1], x.size(-1) // 2, dtype=torch.uint8, device=x.device)
    e8m0 = torch.empty(rows, cols, dtype=torch.float8_e8m0fnu, device=x.device)
    mask = torch.empty(*x.shape[:-1], x.size(-1) // 8, dtype=torch.uint8, device=x.device) if return_mask else None
    _check(load().qutlass_amd_fused_quantize_mx(_p(x), _p(h), h.size(0), x.numel(), {"quest": 0, "abs_max": 1}[method], _p(e2m1), _p(e8m0),
                                               _p(mask) if return_mask else None, _stream()))
    return (e2m1, e8m0, mask) if return_mask else (e2m1, e8m0)


def fused_quantize_matmul_mxf4_bf16_tn(x, h, b, b_sf, alpha, method: str = "quest"):
    """C-ABI call of the LAB build's one-launch decode path (M <= 32)."""
    k = x.size(-1)
    m, n = x.numel() // k, b.size(0)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=x.device)
    _check(load().qutlass_amd_fused_quantize_matmul_mxf4_bf16_tn(_p(x), _p(h), h.size(0), {"quest": 0, "abs_max": 1}[method], _p(b), _p(b_sf), _p(alpha), _p(out),
                                                                m, n, k, _stream()))
    return out


def backward_t_bf16(x, h):
    """C-ABI call of the LAB build (x: (B, N, M) bf16): e2m1 (B, M, N/2) + e8m0 (B, M, N/32)."""
    B, N, M = (1, *x.shape) if x.dim() == 2 else x.shape
    q = torch.empty(B, M, N // 2, dtype=torch.uint8, device=x.device)
    sf = torch.empty(B, M, N // 32, dtype=torch.uint8, device=x.device)
    _check(load().qutlass_amd_backward_t_bf16(_p(x), _p(h), B, N, M, _p(q), _p(sf), _stream()))
    return q, sf


def backward_qt_bf16(xq, xs, h, alpha):
    """C-ABI call of the LAB build (xq: (B, N, M/2) e2m1 bytes, xs: (B, N, M/32) e8m0)."""
    B, N, M2 = (1, *xq.shape) if xq.dim() == 2 else xq.shape
    M = M2 * 2
    q = torch.empty(B, M, N // 2, dtype=torch.uint8, device=xq.device)
    sf = torch.empty(B, M, N // 32, dtype=torch.uint8, device=xq.device)
    _check(load().qutlass_amd_backward_qt_bf16(_p(xq), _p(xs), _p(h), _p(alpha), B, N, M, _p(q), _p(sf), _stream()))
    return q, sf
