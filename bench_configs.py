#!/usr/bin/env python3
"""Secondary measurements: the other BASELINE.json configs (C3, C4, C5) on one MI355X, one JSON line each.

    python bench_configs.py [--iters N]

bench.py stays the headline (configs[1]); this script only records, with the same conventions (operands resident
in HBM, HIP events on the launch stream, algorithmic bytes / FLOPs against the MI355X_MICROARCH.md peaks), what the
remaining hot-path rows of SURVEY.md section 8 do at their BASELINE.json shapes.  Results go to profiles/.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK = {"fp4": 10066.0, "fp8": 5033.0, "f16": 2516.0}   # dense TFLOP/s (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def hadamard(n, device):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(device)


def time_us(fn, iters, warmup=5, ramp_ms=40.0):
    """Average microseconds per call on the DEVICE: `fn` is captured into a HIP graph (`inner` calls, so that one replay
    lasts >= ~0.5 ms and the 8 us of host work per Python op call does not bound short kernels), the graph is replayed for
    at least `ramp_ms` first -- an idle MI355X needs ~40 ms under load to reach its steady clock (tools/clock_ramp.py) --
    and then timed over enough replays to cover `iters` calls.  Falls back to eager launches if capture fails."""
    import time

    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    e1.synchronize()
    est_us = max(e0.elapsed_time(e1) * 1e3 / 3, 1.0)
    inner = int(min(100, max(1, round(500.0 / est_us))))
    try:
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
        torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(inner):
                fn()
        run = g.replay
    except Exception:   # pragma: no cover - capture is expected to work (tests/test_gpu_parity.py)
        torch.cuda.synchronize()

        def run():
            for _ in range(inner):
                fn()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ramp_ms:
        run()
        torch.cuda.synchronize()
    reps = max(3, -(-iters // inner))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * inner)


def time_us_cold(fn_i, nbuf, iters, ramp_ms=40.0):
    """Like time_us, but call j of the captured graph runs fn_i(j % nbuf): with nbuf distinct input tensors that together
    exceed 1 GiB the 256 MiB Infinity Cache (MALL) cannot serve the reads, so the figure is an HBM figure.  (VERDICT r1
    weak #7: replaying one 32 MiB input measured the MALL, not HBM.)"""
    import time

    for j in range(nbuf):
        fn_i(j)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn_i(0)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for j in range(nbuf):
            fn_i(j)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ramp_ms:
        g.replay()
        torch.cuda.synchronize()
    reps = max(3, -(-iters // nbuf))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * nbuf)


def line(name, us, flops=None, peak=None, bytes_=None, **extra):
    d = {"config": name, "us": round(us, 2)}
    if flops:
        tf = flops / us * 1e-6
        d.update({"TFLOP/s": round(tf, 1), "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4)}})
    if bytes_:
        gbs = bytes_ / us * 1e-3
        d.update({"GB/s": round(gbs, 1), "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                     "algorithmic_bytes": bytes_}})
    d.update(extra)
    print(json.dumps(d), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import qutlass_amd as q
    from qutlass_amd.utils import to_blocked

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _benchlib as lab   # the LAB build (forced tiles for the comparison rows below); the product has no such switch

    torch.manual_seed(0)
    alpha = torch.tensor([1.0], device=dev)
    h32 = hadamard(32, dev)

    # ---- C3: fusedQuantizeMx(H32, abs_max) + to_blocked + MXFP4 GEMM, Llama-3-8B FFN M=4096 N=14336 K=4096 ----
    M, N, K = 4096, 14336, 4096
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0
    w = torch.randn(N, K, dtype=torch.bfloat16, device=dev) * 25.0
    w_q, w_s = q.fusedQuantizeMx(w, h32, method="abs_max")
    w_sf = to_blocked(w_s)
    x_q, x_s = q.fusedQuantizeMx(x, h32, method="abs_max")
    x_sf = to_blocked(x_s)
    qbytes = M * K * 2 + M * K // 2 + M * K // 32
    line("C3 fusedQuantizeMx(H32, abs_max) 4096x4096 activations", time_us(lambda: q.fusedQuantizeMx(x, h32, method="abs_max"), args.iters), bytes_=qbytes)
    line("C3 to_blocked(e8m0 4096x128)", time_us(lambda: to_blocked(x_s), args.iters), bytes_=2 * M * K // 32)
    gflops = 2.0 * M * N * K
    line("C3 matmul_mxf4_bf16_tn 4096x14336x4096", time_us(lambda: q.matmul_mxf4_bf16_tn(x_q, w_q, x_sf, w_sf, alpha), args.iters), flops=gflops, peak=PEAK["fp4"])

    def c3_step():
        a_q, a_s = q.fusedQuantizeMx(x, h32, method="abs_max")
        return q.matmul_mxf4_bf16_tn(a_q, w_q, to_blocked(a_s), w_sf, alpha)

    line("C3 step: quantize + to_blocked + GEMM (weights pre-quantised)", time_us(c3_step, args.iters), flops=gflops, peak=PEAK["fp4"])
    del x, w, w_q, w_s, x_q, x_s

    # ---- C4: NVFP4 GEMM 8192^3 ----------------------------------------------------------------------------------
    M = N = K = 8192
    gs = torch.tensor([1.0], device=dev)
    a = torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0
    b = torch.randn(N, K, dtype=torch.bfloat16, device=dev) * 25.0
    h16 = hadamard(16, dev)
    a_q, a_s = q.fusedQuantizeNv(a, h16, gs)
    b_q, b_s = q.fusedQuantizeNv(b, h16, gs)
    a_sf, b_sf = to_blocked(a_s), to_blocked(b_s)
    nvbytes = M * K * 2 + M * K // 2 + M * K // 16
    line("C4 fusedQuantizeNv(H16, abs_max) 8192x8192", time_us(lambda: q.fusedQuantizeNv(a, h16, gs), max(5, args.iters // 3)), bytes_=nvbytes)
    line("C4 matmul_nvf4_bf16_tn 8192^3 (exact semantics on the f16 MFMA)", time_us(lambda: q.matmul_nvf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha), max(3, args.iters // 6)),
         flops=2.0 * M * N * K, peak=PEAK["f16"])
    del a, b, a_q, b_q, a_s, b_s

    # ---- C5: MXFP8 TN + NN 4096^3, Quest + clip-mask quantizer ---------------------------------------------------
    M = N = K = 4096
    a8 = (torch.randn(M, K, device=dev) * 4).to(torch.float8_e4m3fn)
    b8 = (torch.randn(N, K, device=dev) * 4).to(torch.float8_e4m3fn)
    s8a = to_blocked(torch.randint(120, 131, (M, K // 32), dtype=torch.uint8, device=dev).view(torch.float8_e8m0fnu))
    s8b = to_blocked(torch.randint(120, 131, (N, K // 32), dtype=torch.uint8, device=dev).view(torch.float8_e8m0fnu))
    a8t = a8.view(torch.uint8).T.contiguous().view(torch.float8_e4m3fn)
    line("C5 matmul_mxf8_bf16_tn 4096^3", time_us(lambda: q.matmul_mxf8_bf16_tn(a8, b8, s8a, s8b, alpha), args.iters), flops=2.0 * M * N * K, peak=PEAK["fp8"])
    line("C5 matmul_mxf8_bf16_nn 4096^3 (A stored (K, M))", time_us(lambda: q.matmul_mxf8_bf16_nn(a8t, b8, s8a, s8b, alpha), args.iters), flops=2.0 * M * N * K, peak=PEAK["fp8"])
    # configs[4] as BASELINE.json words it: e5m2 gradient x e4m3 activation (extension; the reference rejects e5m2)
    g5 = (torch.randn(M, K, device=dev) * 4 * torch.exp2(torch.randint(-8, 9, (M, 1), device=dev).float())).to(torch.float8_e5m2)
    g5t = g5.view(torch.uint8).T.contiguous().view(torch.float8_e5m2)
    line("C5 matmul_mxf8_bf16_tn 4096^3, A = e5m2 gradient", time_us(lambda: q.matmul_mxf8_bf16_tn(g5, b8, s8a, s8b, alpha), args.iters), flops=2.0 * M * N * K, peak=PEAK["fp8"])
    line("C5 matmul_mxf8_bf16_nn 4096^3, A = e5m2 gradient stored (K, M)", time_us(lambda: q.matmul_mxf8_bf16_nn(g5t, b8, s8a, s8b, alpha), args.iters), flops=2.0 * M * N * K, peak=PEAK["fp8"])
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0
    qb = M * K * 2 + M * K // 2 + M * K // 32
    line("C5 fusedQuantizeMx(H32, quest) 4096x4096", time_us(lambda: q.fusedQuantizeMx(x, h32, method="quest"), args.iters), bytes_=qb)
    line("C5 fusedQuantizeMx(H32, quest, return_mask=True) 4096x4096", time_us(lambda: q.fusedQuantizeMx(x, h32, method="quest", return_mask=True), args.iters), bytes_=qb + M * K // 8)
    # ---- QAT-backward data-prep ops at the C5 shape -------------------------------
    xq, xs = q.fusedQuantizeMx(x, h32, method="abs_max")
    xs2 = xs.view(torch.uint8).reshape(-1)[: M * K // 32].reshape(M, K // 32).contiguous().view(torch.float8_e8m0fnu)
    al3 = torch.tensor([3.0], device=dev)
    line("backward_t_bf16 4096x4096 (transpose + H32 + abs_max MXFP4)", time_us(lambda: q.backward_t_bf16(x, h32), args.iters), bytes_=qb)
    line("backward_qt_bf16 4096x4096 (MXFP4 in, transpose + H32 + abs_max MXFP4 out)", time_us(lambda: q.backward_qt_bf16(xq, xs2, h32, al3), args.iters),
         bytes_=2 * (M * K // 2 + M * K // 32))
    line("backward_bf16_square_double_mxfp8 4096x4096", time_us(lambda: q.backward_bf16_square_double_mxfp8(x), args.iters), bytes_=M * K * 3 + 2 * M * K // 32)
    line("mxfp4_transpose_mxfp8 4096x4096", time_us(lambda: q.mxfp4_transpose_mxfp8(xq, xs2), args.iters), bytes_=M * K // 2 + M * K // 32 + M * K + M * K // 32)
    # ---- small-batch (decode) shapes: skinny split-K kernel vs the tiled kernels ---------------------------------
    w = torch.randn(14336, 4096, dtype=torch.bfloat16, device=dev) * 25.0
    w_q, w_s = q.fusedQuantizeMx(w, h32, method="abs_max")
    w_sf = to_blocked(w_s)
    for (mm, nn) in ((1, 4096), (16, 4096), (32, 4096), (16, 14336), (32, 14336)):
        xa = torch.randn(mm, 4096, dtype=torch.bfloat16, device=dev) * 25.0
        xa_q, xa_s = q.fusedQuantizeMx(xa, h32, method="abs_max")
        xa_sf = to_blocked(xa_s)
        wq, wsf = w_q[:nn], to_blocked(w_s.view(torch.uint8).reshape(-1)[: nn * 128].reshape(nn, 128).view(torch.float8_e8m0fnu))
        wbytes = nn * 4096 // 2 + nn * 128 + mm * 4096 // 2 + 2 * mm * nn
        for var, tag in ((0, "auto: decode form (16 x 16 ... 64 tiles, wave-owned K stages) for M <= 16, one-shot 32-row tiles beyond"), (2, "128x128 lockstep"), (24, "128x128 simple")):
            impl = q if var == 0 else lab
            with lab.forced(gemm_variant=var):
                us = time_us(lambda: impl.matmul_mxf4_bf16_tn(xa_q, wq, xa_sf, wsf, alpha), args.iters)
            line(f"matmul_mxf4_bf16_tn {mm}x{nn}x4096 [{tag}]", us, bytes_=wbytes)
    # ---- mid batch against a long-K layer (Llama-3-8B down-proj 4096 x 14336): ring schedule (+ split-K) vs the 2-stage tiles
    w2 = torch.randn(4096, 14336, dtype=torch.bfloat16, device=dev) * 25.0
    w2_q, w2_s = q.fusedQuantizeMx(w2, h32, method="abs_max")
    w2_sf = to_blocked(w2_s)
    for mm in (16, 64, 256, 512):
        xa = torch.randn(mm, 14336, dtype=torch.bfloat16, device=dev) * 25.0
        xa_q, xa_s = q.fusedQuantizeMx(xa, h32, method="abs_max")
        xa_sf = to_blocked(xa_s)
        wbytes = 4096 * 14336 // 2 + 4096 * 448 + mm * 14336 // 2 + 2 * mm * 4096
        for var, tag in ((0, "auto: decode form / wave-owned rings / ring tiles, split-K when < 256 tiles"), (29, "64x64 simple (2-stage)")):
            impl = q if var == 0 else lab
            with lab.forced(gemm_variant=var):
                us = time_us(lambda: impl.matmul_mxf4_bf16_tn(xa_q, w2_q, xa_sf, w2_sf, alpha), args.iters)
            line(f"matmul_mxf4_bf16_tn {mm}x4096x14336 [{tag}]", us, bytes_=wbytes)
    for r in (64, 128):
        hr = hadamard(r, dev)
        line(f"fusedQuantizeMx(H{r}, abs_max) 4096x4096", time_us(lambda: q.fusedQuantizeMx(x, hr, method="abs_max"), args.iters), bytes_=qb)

    # ---- HBM-honest ("cold") figures of the streaming ops: 40 distinct 32 MiB inputs = 1.25 GiB per graph replay, five times
    #      the 256 MiB Infinity Cache.  The "warm" rows above replay ONE input and are partly served by that cache.
    del w, w_q, w_s, w2, w2_q, w2_s
    NB = 40
    xs_c = [torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0 for _ in range(NB)]
    cold = lambda name, f, b: line(name + " [cold: 40 x 32 MiB inputs rotated]", time_us_cold(f, NB, 4 * NB), bytes_=b, cache="cold")
    cold("fusedQuantizeMx(H32, abs_max) 4096x4096", lambda j: q.fusedQuantizeMx(xs_c[j], h32, method="abs_max"), qb)
    cold("fusedQuantizeMx(H32, quest) 4096x4096", lambda j: q.fusedQuantizeMx(xs_c[j], h32, method="quest"), qb)
    cold("fusedQuantizeMx(H32, quest, return_mask=True) 4096x4096", lambda j: q.fusedQuantizeMx(xs_c[j], h32, method="quest", return_mask=True), qb + M * K // 8)
    for r in (64, 128):
        hr = hadamard(r, dev)
        cold(f"fusedQuantizeMx(H{r}, abs_max) 4096x4096", lambda j: q.fusedQuantizeMx(xs_c[j], hr, method="abs_max"), qb)
    cold("fusedQuantizeNv(H16, abs_max) 4096x4096", lambda j: q.fusedQuantizeNv(xs_c[j], h16, gs), M * K * 2 + M * K // 2 + M * K // 16)
    cold("backward_t_bf16 4096x4096", lambda j: q.backward_t_bf16(xs_c[j], h32), qb)
    cold("backward_bf16_square_double_mxfp8 4096x4096", lambda j: q.backward_bf16_square_double_mxfp8(xs_c[j]), M * K * 3 + 2 * M * K // 32)
    # the same op on a tensor large enough to amortise launch + ramp (~2 us of a 9 us call at 4096^2): 16384 x 8192 bf16 = 256 MiB
    # per input, 5 inputs rotated
    big = [torch.randn(16384, 8192, dtype=torch.bfloat16, device=dev) * 25.0 for _ in range(5)]
    bigb = 16384 * 8192 * 2 + 16384 * 8192 // 2 + 16384 * 8192 // 32
    line("fusedQuantizeMx(H32, abs_max) 16384x8192 [cold: 5 x 256 MiB inputs rotated]", time_us_cold(lambda j: q.fusedQuantizeMx(big[j], h32, method="abs_max"), 5, 20), bytes_=bigb, cache="cold")
    line("fusedQuantizeMx(H32, quest, return_mask=True) 16384x8192 [cold: 5 x 256 MiB inputs rotated]",
         time_us_cold(lambda j: q.fusedQuantizeMx(big[j], h32, method="quest", return_mask=True), 5, 20), bytes_=bigb + 16384 * 8192 // 8, cache="cold")
    h128 = hadamard(128, dev)
    line("fusedQuantizeMx(H128, abs_max) 16384x8192 [cold: 5 x 256 MiB inputs rotated]", time_us_cold(lambda j: q.fusedQuantizeMx(big[j], h128, method="abs_max"), 5, 20), bytes_=bigb, cache="cold")
    del big
    # ... and in between: 8192 x 8192 = 128 MiB per input, 10 inputs rotated (1.25 GiB per cycle)
    mid = [torch.randn(8192, 8192, dtype=torch.bfloat16, device=dev) * 25.0 for _ in range(10)]
    midb = 8192 * 8192 * 2 + 8192 * 8192 // 2 + 8192 * 8192 // 32
    line("fusedQuantizeMx(H32, abs_max) 8192x8192 [cold: 10 x 128 MiB inputs rotated]", time_us_cold(lambda j: q.fusedQuantizeMx(mid[j], h32, method="abs_max"), 10, 20), bytes_=midb, cache="cold")
    h64 = hadamard(64, dev)
    line("fusedQuantizeMx(H64, abs_max) 8192x8192 [cold: 10 x 128 MiB inputs rotated]", time_us_cold(lambda j: q.fusedQuantizeMx(mid[j], h64, method="abs_max"), 10, 20), bytes_=midb, cache="cold")
    del mid
    # [r3] the four QAT-backward data-prep ops at 8192 x 8192 (a 4096^2 call of these ops is 18 - 51 MB, i.e. 3 - 8 us of
    # streaming behind a ~2 us launch floor; the size that shows the kernels themselves is the next one up), 10 inputs rotated
    big8 = [torch.randn(8192, 8192, dtype=torch.bfloat16, device=dev) * 25.0 for _ in range(10)]
    b8 = 8192 * 8192
    line("backward_t_bf16 8192x8192 [cold: 10 x 128 MiB inputs rotated]", time_us_cold(lambda j: q.backward_t_bf16(big8[j], h32), 10, 20), bytes_=b8 * 2 + b8 // 2 + b8 // 32, cache="cold")
    line("backward_bf16_square_double_mxfp8 8192x8192 [cold: 10 x 128 MiB inputs rotated]", time_us_cold(lambda j: q.backward_bf16_square_double_mxfp8(big8[j]), 10, 20),
         bytes_=b8 * 3 + 2 * b8 // 32, cache="cold")
    pk8 = [q.fusedQuantizeMx(t, h32, method="abs_max") for t in big8]
    pk8 = [(pq, ps.view(torch.uint8).reshape(-1)[: b8 // 32].reshape(8192, 8192 // 32).contiguous().view(torch.float8_e8m0fnu)) for pq, ps in pk8]
    del big8
    line("backward_qt_bf16 8192x8192 [cold: 10 x 36 MiB inputs rotated]", time_us_cold(lambda j: q.backward_qt_bf16(pk8[j][0], pk8[j][1], h32, al3), 10, 20),
         bytes_=2 * (b8 // 2 + b8 // 32), cache="cold")
    line("mxfp4_transpose_mxfp8 8192x8192 [cold: 10 x 36 MiB inputs rotated]", time_us_cold(lambda j: q.mxfp4_transpose_mxfp8(pk8[j][0], pk8[j][1]), 10, 20),
         bytes_=b8 // 2 + b8 // 32 + b8 + b8 // 32, cache="cold")
    del pk8
    # packed-input ops: 9 MiB per input, so rotate 40 of them as well (360 MiB > MALL) -- quantise the cold inputs once
    packed = [q.fusedQuantizeMx(t, h32, method="abs_max") for t in xs_c]
    packed = [(pq, ps.view(torch.uint8).reshape(-1)[: M * K // 32].reshape(M, K // 32).contiguous().view(torch.float8_e8m0fnu)) for pq, ps in packed]
    del xs_c
    cold("backward_qt_bf16 4096x4096", lambda j: q.backward_qt_bf16(packed[j][0], packed[j][1], h32, al3), 2 * (M * K // 2 + M * K // 32))
    cold("mxfp4_transpose_mxfp8 4096x4096", lambda j: q.mxfp4_transpose_mxfp8(packed[j][0], packed[j][1]), M * K // 2 + M * K // 32 + M * K + M * K // 32)


if __name__ == "__main__":
    main()
