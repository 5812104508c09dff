#!/usr/bin/env python3
"""Sweep harness in the shape of the reference's benchmarks/bench_mxfp4_sm100.py / bench_nvfp4_sm100.py: TFLOP/s of a
quantised linear layer vs batch size for the layer shapes of a model, three providers

    torch-bf16             torch.nn.functional.linear on bf16 operands (the library GEMM torch ships)
    <fmt>-native           fusedQuantize(activations) + to_blocked + block-scaled GEMM   (weights pre-quantised; the reference's three launches)
    <fmt>-native-fused     [r3, --fused] the same result in fewer launches: fusedQuantize*Blocked + GEMM (two launches), and for MXFP4
                           batches of at most 32 rows ONE launch in which the small-batch GEMM quantises its own A operand
    <fmt>-native-noquant   the GEMM alone on pre-quantised activations ("ideal" provider of the reference)
    mxfp4-vendor-noquant   [r4, --vendor, MXFP4 only] the vendor's block-scaled GEMM on the SAME pre-quantised operands: hipBLASLt through
                           torch.nn.functional.scaled_mm (1x32 e8m0 blocks, row-major scales) -- the counterpart of the reference's cuDNN / flashinfer
                           column (benchmarks/bench_mxfp4_sm100.py:27-31, 216-225).  A reported baseline; nothing in the package imports it.

timed as HIP-graph replays (the reference uses triton.testing.do_bench_cudagraph; triton is not used here), median and
20 / 80 % quantiles over `--reps` replays.  Prints one table per layer and writes CSVs under benchmarks_output/.

    python benchmarks/bench_mxfp4_mi355x.py [--format mxfp4|nvfp4] [--model Llama-3.1-70B] [--had 32] [--quick]
"""
from __future__ import annotations

import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODELS = {   # (K, N) per linear layer, as in the reference harness
    "Llama-3.1-70B": [(8192, 8192), (8192, 57344), (28672, 8192)],
    "Llama-3-8B": [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)],
    "Qwen3-32B": [(5120, 5120), (5120, 51200), (25600, 5120)],
    "Qwen3-8B": [(4096, 4096), (4096, 24576), (12288, 4096)],
}
BATCHES = [1, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 24576, 32768, 65536]


def hadamard(n, device):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(device)


def bench_graph(fn, reps):
    """Median / q20 / q80 milliseconds of one call.  `fn` is captured into a HIP graph `inner` times (so that one replay
    lasts >= ~1 ms), the graph is replayed for >= 50 ms first (an idle MI355X needs ~40 ms under load to reach its steady
    clock, tools/clock_ramp.py), then `reps` replays are timed individually."""
    import time

    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    e1.synchronize()
    est_ms = max(e0.elapsed_time(e1) / 3, 1e-3)
    inner = int(min(200, max(1, round(1.0 / est_ms))))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.05:
        g.replay()
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    t = torch.tensor(ts)
    return t.median().item(), t.quantile(0.2).item(), t.quantile(0.8).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", choices=["mxfp4", "nvfp4"], default="mxfp4")
    ap.add_argument("--model", default="Llama-3.1-70B", choices=sorted(MODELS))
    ap.add_argument("--had", type=int, default=32)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--max-batch", type=int, default=65536)
    ap.add_argument("--quick", action="store_true", help="batch sizes 1, 16, 256, 4096 only, first two layers")
    ap.add_argument("--fused", action="store_true", help="add the <fmt>-native-fused provider (blocked-scale quantizer / one-launch decode path)")
    ap.add_argument("--vendor", action="store_true", help="add the mxfp4-vendor-noquant provider (hipBLASLt block-scaled GEMM via torch scaled_mm)")
    ap.add_argument("--layers", type=int, default=0, help="only the first N layers of the model")
    args = ap.parse_args()

    import qutlass_amd as q
    from qutlass_amd.utils import to_blocked

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    h = hadamard(args.had, dev)
    alpha = torch.tensor([1.0], device=dev)
    gs = torch.tensor([1.0], device=dev)
    nv = args.format == "nvfp4"
    quant = (lambda t: q.fusedQuantizeNv(t, h, gs)) if nv else (lambda t: q.fusedQuantizeMx(t, h, method="abs_max"))
    gemm = q.matmul_nvf4_bf16_tn if nv else q.matmul_mxf4_bf16_tn
    vendor = args.vendor and not nv
    providers = (["torch-bf16", f"{args.format}-native"] + ([f"{args.format}-native-fused"] if args.fused else []) + (["mxfp4-vendor-noquant"] if vendor else [])
                 + [f"{args.format}-native-noquant"])
    if nv:
        fused = lambda t, wq, wsf: gemm(*(lambda aq, asf: (aq, wq, asf, wsf, alpha))(*q.fusedQuantizeNvBlocked(t, h, gs)))
    else:
        fused = lambda t, wq, wsf: q.fused_quantize_matmul_mxf4_bf16_tn(t, h, wq, wsf, alpha, method="abs_max")
    layers = MODELS[args.model][:2] if args.quick else MODELS[args.model]
    if args.layers > 0:
        layers = layers[: args.layers]
    batches = [1, 16, 256, 4096] if args.quick else [b for b in BATCHES if b <= args.max_batch]
    os.makedirs(os.path.join(ROOT, "benchmarks_output"), exist_ok=True)

    for K, N in layers:
        print(f"{args.model}, N={N} K={K}, HAD={args.had}, BF16 vs {args.format.upper()} GEMMs TFLOP/s (median [q20, q80]):")
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        w_q, w_s = quant(w)
        w_sf = to_blocked(w_s)
        rows = []
        print(f"{'batch':>8} " + " ".join(f"{p:>34}" for p in providers))
        for M in batches:
            if M * max(N, K) * 2 > 24 << 30:   # keep single tensors under 24 GiB
                continue
            a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            a_q, a_s = quant(a)
            a_sf = to_blocked(a_s)
            fns = {
                providers[0]: lambda: torch.nn.functional.linear(a, w),
                providers[1]: lambda: gemm(*(lambda aq, asf: (aq, w_q, to_blocked(asf), w_sf, alpha))(*quant(a))),
                providers[-1]: lambda: gemm(a_q, w_q, a_sf, w_sf, alpha),
            }
            if args.fused:
                fns[providers[2]] = lambda: fused(a, w_q, w_sf)
            if vendor:   # same e2m1 bytes, the quantizer's own row-major e8m0 scales (the vendor path takes them un-swizzled on ROCm)
                import torch.nn.functional as F
                A4, B4 = a_q.view(torch.float4_e2m1fn_x2), w_q.view(torch.float4_e2m1fn_x2)
                sa_rm = a_s.view(torch.uint8)[:M, : K // 32].contiguous().view(torch.float8_e8m0fnu)
                sb_rm = w_s.view(torch.uint8)[:N, : K // 32].contiguous().view(torch.float8_e8m0fnu)
                fns["mxfp4-vendor-noquant"] = lambda: F.scaled_mm(A4, B4.t(), sa_rm, F.ScalingType.BlockWise1x32, sb_rm, F.ScalingType.BlockWise1x32,
                                                                  F.SwizzleType.NO_SWIZZLE, F.SwizzleType.NO_SWIZZLE, None, torch.bfloat16)
            row = {"batch": M}
            cells = []
            for pname in providers:
                try:
                    ms, lo, hi = bench_graph(fns[pname], args.reps)
                except Exception as e:   # (the vendor path rejects some shapes: recorded as nan)
                    if pname != "mxfp4-vendor-noquant":
                        raise
                    ms = lo = hi = float("nan")
                tf = lambda t: 2.0 * M * N * K * 1e-12 / (t * 1e-3)
                row[pname], row[pname + "_q20"], row[pname + "_q80"] = tf(ms), tf(hi), tf(lo)
                cells.append(f"{tf(ms):10.1f} [{tf(hi):8.1f},{tf(lo):8.1f}]")
            rows.append(row)
            print(f"{M:8d} " + " ".join(f"{c:>34}" for c in cells), flush=True)
            del a, a_q, a_s, a_sf
        path = os.path.join(ROOT, "benchmarks_output", f"bench_{args.format}_res_n{N}_k{K}_had{args.had}_mi355x.csv")
        with open(path, "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            wr.writeheader()
            wr.writerows(rows)
        print(f"  -> {os.path.relpath(path, ROOT)}\n")
        del w, w_q, w_s, w_sf


if __name__ == "__main__":
    main()
