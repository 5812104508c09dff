"""CPU parity oracle for the qutlass hot path -- TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  ``qutlass_amd`` (the product) never does.

The arithmetic lives in ``qutlass_oracle.c`` (plain scalar C, each function citing the reference
file:line it restates); this module is a thin numpy/ctypes front end.  Pinned against golden
vectors generated from the reference's own Python test oracles (``tests/golden/``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libqutlass_oracle.so")

QUEST, ABS_MAX = 0, 1
KIND_MXFP4, KIND_NVFP4, KIND_MXFP8_TN, KIND_MXFP8_NN = 0, 1, 2, 3
KIND_MXFP8_TN_A5, KIND_MXFP8_NN_A5 = 4, 5   # e5m2 A operand x e4m3 B operand (extension; see qutlass_oracle.c orc_e5m2_decode)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "qutlass_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libqutlass_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.orc_e2m1_decode.restype = f32
        L.orc_e2m1_decode.argtypes = [ctypes.c_uint8]
        L.orc_e2m1_encode.restype = ctypes.c_uint8
        L.orc_e2m1_encode.argtypes = [f32]
        L.orc_e4m3_decode.restype = f32
        L.orc_e4m3_decode.argtypes = [ctypes.c_uint8]
        L.orc_e4m3_encode.restype = ctypes.c_uint8
        L.orc_e4m3_encode.argtypes = [f32]
        L.orc_to_blocked.restype = None
        L.orc_to_blocked.argtypes = [vp, i64, i64, vp]
        L.orc_fused_quantize_mx.restype = None
        L.orc_fused_quantize_mx.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp, vp]
        L.orc_fused_quantize_nv.restype = None
        L.orc_fused_quantize_nv.argtypes = [vp, vp, i32, i64, i32, i32, f32, vp, vp]
        L.orc_gemm_blockscaled.restype = None
        L.orc_gemm_blockscaled.argtypes = [i32, vp, vp, vp, vp, f32, i64, i64, i64, vp]
        L.orc_pseudoquant_mxfp8.restype = None
        L.orc_pseudoquant_mxfp8.argtypes = [vp, i64, vp, vp]
        L.orc_pseudoquant_mxfp8_e5m2.restype = None
        L.orc_pseudoquant_mxfp8_e5m2.argtypes = [vp, i64, vp, vp]
        L.orc_e5m2_decode.restype = f32
        L.orc_e5m2_decode.argtypes = [ctypes.c_uint8]
        L.orc_e5m2_encode.restype = ctypes.c_uint8
        L.orc_e5m2_encode.argtypes = [f32]
        L.orc_dequant_fp4.restype = None
        L.orc_dequant_fp4.argtypes = [vp, vp, i32, i32, i64, f32, vp]
        L.orc_backward_t_bf16.restype = None
        L.orc_backward_t_bf16.argtypes = [vp, vp, i64, i64, i64, i32, vp, vp]
        L.orc_backward_qt_bf16.restype = None
        L.orc_backward_qt_bf16.argtypes = [vp, vp, vp, f32, i64, i64, i64, i32, vp, vp]
        L.orc_backward_bf16_square_double_mxfp8.restype = None
        L.orc_backward_bf16_square_double_mxfp8.argtypes = [vp, i64, i64, vp, vp, vp]
        L.orc_mxfp4_transpose_mxfp8.restype = None
        L.orc_mxfp4_transpose_mxfp8.argtypes = [vp, vp, i64, i64, vp, vp]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u8(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    assert a.dtype.itemsize == 1, a.dtype
    return a.view(np.uint8)


def _u16(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    assert a.dtype.itemsize == 2, a.dtype
    return a.view(np.uint16)


# ---------------------------------------------------------------------------------------------
def e2m1_encode(x: float) -> int:
    return int(lib().orc_e2m1_encode(float(x)))


def e2m1_decode(code: int) -> float:
    return float(lib().orc_e2m1_decode(int(code) & 0xF))


def e4m3_encode(x: float) -> int:
    return int(lib().orc_e4m3_encode(float(x)))


def e4m3_decode(b: int) -> float:
    return float(lib().orc_e4m3_decode(int(b) & 0xFF))


def padded_shape(rows: int, cols: int):
    """qutlass/utils.py:140-157 -- round (rows, cols) up to (128, 4)."""
    return -(-rows // 128) * 128, -(-cols // 4) * 4


def to_blocked(sf: np.ndarray) -> np.ndarray:
    """qutlass/utils.py:160-193 (+ Triton path's zero padding).  sf: (rows, cols) 1-byte dtype."""
    sf = _u8(sf)
    rows, cols = sf.shape
    pr, pc = padded_shape(rows, cols)
    out = np.empty(pr * pc, dtype=np.uint8)
    lib().orc_to_blocked(_p(sf), rows, cols, _p(out))
    return out


def fused_quantize_mx(x_bf16, h_bf16, method: int, with_mask: bool = False, acc_model: int = 0):
    """x_bf16: any-shape array of bf16 bit patterns (uint16 view); h_bf16: (R, R) bf16 bits.

    Returns (e2m1 packed u8 [numel/2], e8m0 u8 [numel/32], mask u8 [numel/8] or None), flat.
    """
    x = _u16(x_bf16).reshape(-1)
    h = _u16(h_bf16)
    R = h.shape[0]
    assert h.shape == (R, R) and R in (32, 64, 128) and x.size % R == 0
    n = x.size
    q = np.empty(n // 2, dtype=np.uint8)
    s = np.empty(n // 32, dtype=np.uint8)
    m = np.empty(n // 32, dtype=np.uint32) if with_mask else None
    lib().orc_fused_quantize_mx(_p(x), _p(h), R, n, method, acc_model, _p(q), _p(s),
                                _p(m) if m is not None else None)
    return q, s, (m.view(np.uint8) if m is not None else None)


def fused_quantize_nv(x_bf16, h_bf16, global_scale: float, method: int = ABS_MAX, acc_model: int = 0):
    x = _u16(x_bf16).reshape(-1)
    h = _u16(h_bf16)
    R = h.shape[0]
    assert h.shape == (R, R) and R in (16, 32, 64, 128) and x.size % R == 0
    n = x.size
    q = np.empty(n // 2, dtype=np.uint8)
    s = np.empty(n // 16, dtype=np.uint8)
    lib().orc_fused_quantize_nv(_p(x), _p(h), R, n, method, acc_model, float(global_scale), _p(q), _p(s))
    return q, s


def gemm_blockscaled(kind: int, a, b, sfa_blocked, sfb_blocked, alpha: float, m: int, n: int, k: int):
    """Returns D as bf16 bit patterns, uint16 (m, n)."""
    a, b = _u8(a).reshape(-1), _u8(b).reshape(-1)
    sfa, sfb = _u8(sfa_blocked).reshape(-1), _u8(sfb_blocked).reshape(-1)
    gs = 16 if kind == KIND_NVFP4 else 32
    need = lambda r: padded_shape(r, k // gs)[0] * padded_shape(r, k // gs)[1]
    assert sfa.size >= need(m) and sfb.size >= need(n), (sfa.size, need(m), sfb.size, need(n))
    d = np.empty((m, n), dtype=np.uint16)
    lib().orc_gemm_blockscaled(kind, _p(a), _p(b), _p(sfa), _p(sfb), float(alpha), m, n, k, _p(d))
    return d


def pseudoquant_mxfp8(x_bf16, e5m2: bool = False):
    """tests/mxfp8_test.py:26-46; e5m2=True: the same expression with the e5m2 constants (extension, see the C source)."""
    x = _u16(x_bf16)
    q = np.empty(x.shape, dtype=np.uint8)
    s = np.empty(x.size // 32, dtype=np.uint8)
    (lib().orc_pseudoquant_mxfp8_e5m2 if e5m2 else lib().orc_pseudoquant_mxfp8)(_p(x.reshape(-1)), x.size, _p(q), _p(s))
    return q, s.reshape(x.shape[:-1] + (x.shape[-1] // 32,))


def e5m2_encode(x: float) -> int:
    return int(lib().orc_e5m2_encode(float(x)))


def e5m2_decode(b: int) -> float:
    return float(lib().orc_e5m2_decode(int(b) & 0xFF))


def dequant_fp4(packed, sf_flat, gs: int = 32, is_e4m3: bool = False, alpha: float = 1.0) -> np.ndarray:
    packed = _u8(packed).reshape(-1)
    sf = _u8(sf_flat).reshape(-1)
    n = packed.size * 2
    assert sf.size >= n // gs
    out = np.empty(n, dtype=np.float64)
    lib().orc_dequant_fp4(_p(packed), _p(sf), gs, int(is_e4m3), n, float(alpha), _p(out))
    return out


def bf16_bits_to_f32(u16: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(u16).astype(np.uint32) << 16).view(np.float32)


def codes_equal_mod_zero_sign(a_packed: np.ndarray, b_packed: np.ndarray) -> np.ndarray:
    """Per-element equality of two packed e2m1 arrays treating +0 (0x0) and -0 (0x8) as equal."""
    a, b = _u8(a_packed).reshape(-1), _u8(b_packed).reshape(-1)
    al, ah, bl, bh = a & 0xF, a >> 4, b & 0xF, b >> 4
    z = lambda c: np.where((c & 7) == 0, 0, c)
    return np.stack([z(al) == z(bl), z(ah) == z(bh)], axis=-1).reshape(-1)


# ---- QAT-backward data-prep kernels (quartet_bwd_sm120.cu) ------------------------------------
def backward_t_bf16(x_bf16, h_bf16, acc_model: int = 0):
    """x: (B, N, M) or (N, M) bf16 bits -> (e2m1 (B, M, N/2) u8, e8m0 (B, M, N/32) u8)."""
    x = _u16(x_bf16)
    if x.ndim == 2:
        x = x[None]
    B, N, M = x.shape
    h = _u16(h_bf16)
    assert h.shape == (32, 32) and N % 32 == 0
    q = np.empty((B, M, N // 2), dtype=np.uint8)
    s = np.empty((B, M, N // 32), dtype=np.uint8)
    lib().orc_backward_t_bf16(_p(np.ascontiguousarray(x)), _p(h), B, N, M, acc_model, _p(q), _p(s))
    return q, s


def backward_qt_bf16(x_e2m1, x_e8m0, h_bf16, alpha: float, acc_model: int = 0):
    """x_e2m1: (B, N, M/2), x_e8m0: (B, N, M/32) -> (e2m1 (B, M, N/2), e8m0 (B, M, N/32))."""
    xq, xs = _u8(x_e2m1), _u8(x_e8m0)
    if xq.ndim == 2:
        xq, xs = xq[None], xs[None]
    B, N, M2 = xq.shape
    M = M2 * 2
    h = _u16(h_bf16)
    assert h.shape == (32, 32) and N % 32 == 0 and xs.shape == (B, N, M // 32)
    q = np.empty((B, M, N // 2), dtype=np.uint8)
    s = np.empty((B, M, N // 32), dtype=np.uint8)
    lib().orc_backward_qt_bf16(_p(np.ascontiguousarray(xq)), _p(np.ascontiguousarray(xs)), _p(h), float(alpha), B, N, M,
                               acc_model, _p(q), _p(s))
    return q, s


def backward_bf16_square_double_mxfp8(x_bf16):
    """x: (m, n) bf16 bits, m % 32 == 0, n % 32 == 0 -> (e4m3 (m, n), row_scales (m, n/32), col_scales (n, m/32))."""
    x = np.ascontiguousarray(_u16(x_bf16))
    m, n = x.shape
    assert m % 32 == 0 and n % 32 == 0
    y = np.empty((m, n), dtype=np.uint8)
    rs = np.empty((m, n // 32), dtype=np.uint8)
    cs = np.empty((n, m // 32), dtype=np.uint8)
    lib().orc_backward_bf16_square_double_mxfp8(_p(x), m, n, _p(y), _p(rs), _p(cs))
    return y, rs, cs


def mxfp4_transpose_mxfp8(x_fp4, scales):
    """x_fp4: (m, n/2) packed e2m1, scales: (m, n/32) e8m0 -> (e4m3 (n, m), e8m0 (n, m/32))."""
    xq, xs = np.ascontiguousarray(_u8(x_fp4)), np.ascontiguousarray(_u8(scales))
    m, n2 = xq.shape
    n = n2 * 2
    assert m % 32 == 0 and xs.shape == (m, n // 32)
    y = np.empty((n, m), dtype=np.uint8)
    e = np.empty((n, m // 32), dtype=np.uint8)
    lib().orc_mxfp4_transpose_mxfp8(_p(xq), _p(xs), m, n, _p(y), _p(e))
    return y, e

