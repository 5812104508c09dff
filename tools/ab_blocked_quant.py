#!/usr/bin/env python3
"""A/B of the activation path's launch count: per shape, device time (HIP-graph replays, bench_configs.time_us) of
  quantize (flat scales) | to_blocked | quantize + to_blocked (the reference's two launches) | fusedQuantizeMxBlocked (one launch)
and of the whole linear layer  3 launches | 2 launches (blocked quantizer) | 1 launch (decode path, M <= 32) | GEMM alone.
    python tools/ab_blocked_quant.py > gpurun_out/ab_blocked_quant.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench_configs import hadamard, time_us  # noqa: E402


def main():
    import qutlass_amd as q
    from qutlass_amd.utils import to_blocked

    dev = torch.device("cuda", 0)
    h = hadamard(32, dev)
    alpha = torch.tensor([1.0], device=dev)
    print("# quantizer: device us per call")
    print(f"{'shape':>14} {'method':>8} {'quantize':>9} {'to_blocked':>10} {'q + to_blocked':>14} {'q blocked':>10}")
    for rows, k in ((32, 4096), (256, 4096), (1024, 4096), (4096, 4096), (8192, 8192), (4096, 14336)):
        x = torch.randn(rows, k, dtype=torch.bfloat16, device=dev) * 25.0
        for method in ("abs_max", "quest"):
            _, s = q.fusedQuantizeMx(x, h, method=method)
            t_q = time_us(lambda: q.fusedQuantizeMx(x, h, method=method), 200)
            t_b = time_us(lambda: to_blocked(s), 200)
            t_qb = time_us(lambda: to_blocked(q.fusedQuantizeMx(x, h, method=method)[1]), 200)
            t_f = time_us(lambda: q.fusedQuantizeMxBlocked(x, h, method=method), 200)
            print(f"{rows:>6}x{k:<7} {method:>8} {t_q:9.2f} {t_b:10.2f} {t_qb:14.2f} {t_f:10.2f}", flush=True)
    print("# linear layer y = Q(x h) W^T, weights pre-quantised: device us per call")
    print(f"{'M x N x K':>20} {'3 launches':>11} {'2 launches':>11} {'1 launch':>9} {'GEMM alone':>11}")
    for n, k in ((4096, 4096), (6144, 4096), (4096, 8192), (2048, 2048), (4096, 14336), (14336, 4096)):
        w = torch.randn(n, k, dtype=torch.bfloat16, device=dev) * 25.0
        w_q, w_s = q.fusedQuantizeMx(w, h, method="abs_max")
        w_sf = to_blocked(w_s)
        for m in (1, 8, 16, 32, 64, 256, 4096):
            x = torch.randn(m, k, dtype=torch.bfloat16, device=dev) * 25.0
            a_q, a_s = q.fusedQuantizeMx(x, h, method="abs_max")
            a_sf = to_blocked(a_s)

            def three():
                aq, as_ = q.fusedQuantizeMx(x, h, method="abs_max")
                return q.matmul_mxf4_bf16_tn(aq, w_q, to_blocked(as_), w_sf, alpha)

            def two():
                aq, asb = q.fusedQuantizeMxBlocked(x, h, method="abs_max")
                return q.matmul_mxf4_bf16_tn(aq, w_q, asb, w_sf, alpha)

            t3, t2 = time_us(three, 100), time_us(two, 100)
            t1 = time_us(lambda: q.fused_quantize_matmul_mxf4_bf16_tn(x, h, w_q, w_sf, alpha, method="abs_max", single_launch=True), 100) if m <= 32 else float("nan")
            tg = time_us(lambda: q.matmul_mxf4_bf16_tn(a_q, w_q, a_sf, w_sf, alpha), 100)
            print(f"{m:>6}x{n:>6}x{k:<6} {t3:11.2f} {t2:11.2f} {t1:9.2f} {tg:11.2f}", flush=True)


if __name__ == "__main__":
    main()
