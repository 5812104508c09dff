#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own Python test oracles.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
It imports -- it does not copy -- the reference's pure-torch oracles

    tests/mxfp4_test.py : _rtne_fp4, _dq_fp4, _forward_quantize_ref        (MX quantiser + GEMM oracle)
    tests/nvfp4_test.py : _forward_quantize_ref, _dq_fp4                   (NV quantiser + GEMM oracle)
    tests/mxfp8_test.py : _pseudoquant_mxfp8                               (MXFP8 operand producer)
    qutlass/utils.py    : to_blocked (torch path), get_padded_shape_mx/nv

on CPU, feeds them fixed-seed inputs and stores inputs + expected outputs as small .npz files.
The fixtures are data (inputs / expected outputs); no reference source text is stored.

Usage:  python tests/golden/make_golden.py            (re-creates every fixture, deterministic)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from scipy.linalg import hadamard

REF = os.environ.get("QUTLASS_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    utils = _load(os.path.join(REF, "qutlass", "utils.py"), "ref_qutlass_utils")
    # Stub the compiled package so the test modules import; neutralise the CUDA requirement.
    stub = types.ModuleType("qutlass")
    for n in ("matmul_mxf4_bf16_tn", "fusedQuantizeMx", "matmul_nvf4_bf16_tn", "fusedQuantizeNv",
              "matmul_mxf8_bf16_tn", "matmul_mxf8_bf16_nn", "backward_t_bf16", "backward_qt_bf16",
              "backward_bf16_square_double_mxfp8", "mxfp4_transpose_mxfp8"):
        setattr(stub, n, None)
    stub.utils = utils
    sys.modules["qutlass"] = stub
    sys.modules["qutlass.utils"] = utils
    real_avail, real_device, real_compile = torch.cuda.is_available, torch.device, torch.compile
    torch.cuda.is_available = lambda: True
    torch.device = lambda *a, **k: real_device("cpu")
    torch.compile = lambda *a, **k: (lambda f: f)
    try:
        mx = _load(os.path.join(REF, "tests", "mxfp4_test.py"), "ref_mxfp4_test")
        nv = _load(os.path.join(REF, "tests", "nvfp4_test.py"), "ref_nvfp4_test")
        f8 = _load(os.path.join(REF, "tests", "mxfp8_test.py"), "ref_mxfp8_test")
        global QT
        QT = _load(os.path.join(REF, "tests", "quartet_test.py"), "ref_quartet_test")
    finally:
        torch.cuda.is_available, torch.device, torch.compile = real_avail, real_device, real_compile
    return utils, mx, nv, f8


def bits16(t):  # bf16 tensor -> uint16 numpy
    return t.contiguous().view(torch.uint16).numpy().copy()


def u8(t):
    return t.contiguous().view(torch.uint8).numpy().copy()


def had(R):
    return torch.tensor(hadamard(R) * R ** -0.5, dtype=torch.bfloat16)


def main():
    utils, mx, nv, f8 = load_reference()
    torch.manual_seed(0)
    np.random.seed(0)

    # ---- e2m1 known-answer table (mxfp4_test.py:45-81) -------------------------------------
    vals = torch.tensor([0.0, 0.24, 0.25, 0.26, 0.5, 0.74, 0.75, 0.76, 1.0, 1.24, 1.25, 1.26, 1.5, 1.74,
                         1.75, 1.76, 2.0, 2.49, 2.5, 2.51, 3.0, 3.49, 3.5, 3.51, 4.0, 4.99, 5.0, 5.01,
                         6.0, 7.5, 100.0, 1e30], dtype=torch.float64)
    vals = torch.cat([vals, -vals])
    y, packed = mx._rtne_fp4(vals)
    np.savez(os.path.join(OUT, "e2m1_kat.npz"), x=vals.numpy(), y=y.numpy(), packed=packed.numpy())

    # ---- to_blocked (utils.py:160-193, torch path; padded shapes only) ---------------------
    d = {}
    for i, (r, c) in enumerate([(128, 4), (256, 16), (384, 12), (512, 128)]):
        a = torch.randint(0, 256, (r, c), dtype=torch.uint8)
        if i == 0:
            a = torch.arange(r * c, dtype=torch.int32).remainder(251).to(torch.uint8).reshape(r, c)
        d[f"in{i}"] = a.numpy()
        d[f"out{i}"] = utils.to_blocked(a).numpy()
    np.savez(os.path.join(OUT, "to_blocked.npz"), **d)

    # ---- MX quantiser (mxfp4_test.py:135-184) ----------------------------------------------
    d = {}
    case = 0
    for R in (32, 64, 128):
        for shape in ((4, 256), (3, 16, 512), (1, 4096)):
            for quest in (True, False):
                x = torch.randn(*shape, dtype=torch.bfloat16) * 25.0
                h = had(R)
                _, _, (e2m1, e8m0, mask) = mx._forward_quantize_ref(x, h, R, quest=quest)
                d[f"x{case}"], d[f"h{case}"] = bits16(x), bits16(h)
                d[f"e2m1_{case}"], d[f"e8m0_{case}"], d[f"mask{case}"] = u8(e2m1), u8(e8m0), u8(mask)
                d[f"meta{case}"] = np.array([R, int(quest)])
                case += 1
    # identity rotation (quartet_test.py:380 passes torch.eye(32)) + a non-orthogonal random h
    for h in (torch.eye(32, dtype=torch.bfloat16), (torch.randn(32, 32) * 0.2).to(torch.bfloat16)):
        for quest in (True, False):
            x = torch.randn(8, 512, dtype=torch.bfloat16) * 25.0
            _, _, (e2m1, e8m0, mask) = mx._forward_quantize_ref(x, h, 32, quest=quest)
            d[f"x{case}"], d[f"h{case}"] = bits16(x), bits16(h)
            d[f"e2m1_{case}"], d[f"e8m0_{case}"], d[f"mask{case}"] = u8(e2m1), u8(e8m0), u8(mask)
            d[f"meta{case}"] = np.array([32, int(quest)])
            case += 1
    d["ncases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "quantize_mx.npz"), **d)

    # ---- MXFP4 GEMM (mxfp4_test.py:224-237): quantise with the reference oracle, dq, fp64 matmul
    d = {}
    case = 0
    for (m, n, k, alpha, quest) in [(256, 256, 512, 1.0, False), (128, 128, 128, 1.0, True),
                                    (72, 136, 640, 0.5, True), (1, 504, 1024, 1.0, False),
                                    (16, 40, 256, 1.0, True)]:
        h = had(32)
        a = torch.randn(m, k, dtype=torch.bfloat16) * 25.0
        b = torch.randn(n, k, dtype=torch.bfloat16) * 25.0
        _, _, (a_q, a_s, _) = mx._forward_quantize_ref(a, h, 32, quest=quest)
        _, _, (b_q, b_s, _) = mx._forward_quantize_ref(b, h, 32, quest=quest)
        a_dq, *_ = mx._dq_fp4(a_q, a_s, alpha=1.0)
        b_dq, *_ = mx._dq_fp4(b_q, b_s, alpha=1.0)
        out = ((a_dq @ b_dq.T) * alpha).to(torch.bfloat16)
        d[f"a{case}"], d[f"b{case}"] = u8(a_q), u8(b_q)
        d[f"asf{case}"], d[f"bsf{case}"] = u8(a_s), u8(b_s)  # row-major (m, k/32), un-swizzled
        d[f"out{case}"] = bits16(out)
        d[f"meta{case}"] = np.array([m, n, k])
        d[f"alpha{case}"] = np.array(alpha, dtype=np.float32)
        case += 1
    d["ncases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "gemm_mxfp4.npz"), **d)

    # ---- NV quantiser + NVFP4 GEMM (nvfp4_test.py:127-224) ---------------------------------
    d = {}
    case = 0
    for R in (16, 32, 64, 128):
        x = torch.randn(4, 512, dtype=torch.bfloat16) * 25.0
        h = had(R)
        _, _, (e2m1, e4m3, _) = nv._forward_quantize_ref(x, h, R)
        d[f"x{case}"], d[f"h{case}"] = bits16(x), bits16(h)
        d[f"e2m1_{case}"], d[f"e4m3_{case}"] = u8(e2m1), u8(e4m3)
        d[f"meta{case}"] = np.array([R])
        case += 1
    d["ncases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "quantize_nv.npz"), **d)

    d = {}
    case = 0
    for (m, n, k, alpha) in [(128, 128, 128, 1.0), (72, 136, 320, 0.5), (16, 64, 256, 1.0)]:
        h = had(16)
        a = torch.randn(m, k, dtype=torch.bfloat16) * 25.0
        b = torch.randn(n, k, dtype=torch.bfloat16) * 25.0
        _, _, (a_q, a_s, _) = nv._forward_quantize_ref(a, h, 16)
        _, _, (b_q, b_s, _) = nv._forward_quantize_ref(b, h, 16)
        a_dq, *_ = nv._dq_fp4(a_q, a_s, alpha=1.0)
        b_dq, *_ = nv._dq_fp4(b_q, b_s, alpha=1.0)
        out = ((a_dq @ b_dq.T) * alpha).to(torch.bfloat16)
        d[f"a{case}"], d[f"b{case}"] = u8(a_q), u8(b_q)
        d[f"asf{case}"], d[f"bsf{case}"] = u8(a_s), u8(b_s)
        d[f"out{case}"] = bits16(out)
        d[f"meta{case}"] = np.array([m, n, k])
        d[f"alpha{case}"] = np.array(alpha, dtype=np.float32)
        case += 1
    d["ncases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "gemm_nvfp4.npz"), **d)

    # ---- MXFP8 pseudo-quant + GEMM (mxfp8_test.py:26-46, 58-75) ----------------------------
    d = {}
    case = 0
    for (m, n, k, dist) in [(16, 64, 256, "rand"), (40, 72, 384, "randn"), (128, 128, 128, "rand")]:
        gen = torch.rand if dist == "rand" else torch.randn
        a = gen(m, k, dtype=torch.bfloat16) * 25.0
        b = gen(n, k, dtype=torch.bfloat16) * 25.0
        a_dq, (a_q, a_s) = f8._pseudoquant_mxfp8(a)
        b_dq, (b_q, b_s) = f8._pseudoquant_mxfp8(b)
        out = (a_dq.double() @ b_dq.double().T).to(torch.bfloat16)
        d[f"xa{case}"], d[f"xb{case}"] = bits16(a), bits16(b)
        d[f"a{case}"], d[f"b{case}"] = u8(a_q), u8(b_q)
        d[f"asf{case}"], d[f"bsf{case}"] = u8(a_s), u8(b_s)
        d[f"out{case}"] = bits16(out)
        d[f"meta{case}"] = np.array([m, n, k])
        case += 1
    d["ncases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "gemm_mxfp8.npz"), **d)

    # ---- QAT-backward data-prep oracles (quartet_test.py:155-173, 239-260, 284-366) ---------
    qt = QT
    d = {}
    h = had(32)
    # backward_t_bf16: x (B, N, M) -> abs-max MXFP4 of x^T rotated per 32 along N
    for case, shape in enumerate([(1, 64, 96), (2, 128, 64), (1, 32, 8)]):
        x = torch.randn(*shape, dtype=torch.bfloat16) * 25.0
        _, (e2m1, e8m0) = qt._backward_quantize_ref(x.transpose(-2, -1), h)
        d[f"t_x{case}"], d[f"t_e2m1_{case}"], d[f"t_e8m0_{case}"] = bits16(x), u8(e2m1), u8(e8m0)
    d["t_ncases"] = np.array(3)
    # backward_qt_bf16: abs-max MXFP4 input (reference MX oracle) -> dequant / 3, transpose, requantise
    for case, shape in enumerate([(1, 64, 128), (2, 96, 64)]):
        x = torch.randn(*shape, dtype=torch.bfloat16) * 25.0
        _, _, (xq, xs, _) = mx._forward_quantize_ref(x, h, 32, quest=False)
        xs = xs.reshape(*shape[:-1], shape[-1] // 32)
        x_dq = qt._dq_fp4(xq, xs, alpha=3.0)[0]
        dq_ref, (e2m1, e8m0) = qt._backward_quantize_ref(x_dq.transpose(-2, -1), h)
        d[f"qt_xq{case}"], d[f"qt_xs{case}"] = u8(xq), u8(xs)
        d[f"qt_e2m1_{case}"], d[f"qt_e8m0_{case}"] = u8(e2m1), u8(e8m0)
        d[f"qt_dq{case}"] = dq_ref.to(torch.float32).numpy()
    d["qt_ncases"] = np.array(2)
    d["h"] = bits16(h)
    # backward_bf16_square_double_mxfp8 (the reference's own test input: arange rows, plus random and a zero block)
    xs_ = [torch.arange(0, 256, dtype=torch.bfloat16)[None, :].repeat(160, 1),
           torch.randn(128, 128, dtype=torch.bfloat16) * 25.0]
    xs_[1][32:64, 64:96] = 0
    for case, x in enumerate(xs_):
        y, rs, cs = qt._backward_bf16_square_double_mxfp8(x)
        d[f"sq_x{case}"], d[f"sq_y{case}"], d[f"sq_rs{case}"], d[f"sq_cs{case}"] = bits16(x), u8(y), u8(rs), u8(cs)
    d["sq_ncases"] = np.array(2)
    # mxfp4_transpose_mxfp8
    for case, shape in enumerate([(256, 128), (512, 64)]):
        x = torch.randn(*shape, dtype=torch.bfloat16) * 25.0
        _, _, (xq, xs, _) = mx._forward_quantize_ref(x, torch.eye(32, dtype=torch.bfloat16), 32, quest=False)
        xs = xs.reshape(shape[0], shape[1] // 32)
        y, e = qt._mxfp4_transpose_mxfp8(xq.clone(), xs.clone().view(torch.uint8))
        d[f"tr_xq{case}"], d[f"tr_xs{case}"], d[f"tr_y{case}"], d[f"tr_e{case}"] = u8(xq), u8(xs), u8(y), u8(e)
    d["tr_ncases"] = np.array(2)
    np.savez_compressed(os.path.join(OUT, "quartet_bwd.npz"), **d)

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
