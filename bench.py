#!/usr/bin/env python3
"""Headline benchmark: MXFP4 GEMM 4096x4096x4096 (BASELINE.json configs[1]) on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one matmul_mxf4_bf16_tn over pre-quantised, HBM-resident operands ("ideal" provider of the
reference's own harness, benchmarks/bench_mxfp4_sm120.py:27-31,109-127).  The path is a per-layer dense
contraction with no exchange step, so ranks are independent replicas over the same shapes: `value` is
the sum of the ranks' work over the max-over-ranks time ("scaling": "weak", no data-path collective).

Prints ONE JSON line on rank 0 with `roofline` (MFMA-bound: FLOP/s of the dominant kernel from HIP
events on the launch stream vs the FP4 dense peak of MI355X_MICROARCH.md) and `cpu_baseline` (the
reference's dequantise + torch.matmul oracle path timed on this host's cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP4_DENSE_PEAK_TFLOPS = 10066.0  # 256 CU x 4 SIMD x 2048 MAC/clk x 2 x 2.4 GHz (MI355X_MICROARCH.md: ~10 PF dense)
# MFMA-only loop (no memory traffic) with RANDOM fp4 operands, measured on this part: the power limit holds the
# clock near 1.6 GHz (profiles/ubench_r1f_const_vs_random_operands.log).  Informational only.
FP4_SUSTAINED_RANDOM_TFLOPS = 6550.0
PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_bench_r1.json")  # written by tools/pmc_bench.sh on the GPU box
M = N = K = 4096
CPU_ROWS = 4096  # cpu_baseline sample: the whole workload (measured 3.2 s per 1024 rows on the 256-thread host)


def hadamard(n, device):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(device)


def cpu_baseline(a_q, a_s, b_q, b_s, rows):
    """Reference oracle path (tests/mxfp4_test.py:84-120,229-231) restated in oracle/dequant_matmul.py:
    dequantise both packed operands to fp64 and a_dq @ b_dq.T -> bf16, on a `rows`-row slab of A against
    all of B, on the host cores."""
    from oracle import dequant_matmul as dm

    torch.set_num_threads(os.cpu_count() or 1)
    a_q, a_s = a_q[:rows].cpu(), a_s[:rows].cpu()
    b_q, b_s = b_q.cpu(), b_s.cpu()
    t0 = time.perf_counter()
    out = dm.dequant_matmul_mxfp4(a_q, a_s, b_q, b_s, alpha=1.0, dtype=torch.float64)
    dt = time.perf_counter() - t0
    flops = 2.0 * rows * N * K
    return {
        "value": round(flops / dt / 1e12, 4),
        "unit": "TFLOP/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"fp64 dequantise(A[:{rows}],B) + torch.matmul -> bf16, {rows}x{N}x{K} slab of the same operands, {dt:.2f} s",
    }, out


def max_over_ranks(wall: float, device=None) -> float:
    """The job's step time is the slowest rank's (replicas: no data-path collective, only this MAX-reduce and the
    barriers around the timed region go through torch.distributed -- RCCL on GPUs, gloo in the CPU test)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return wall
    tw = torch.tensor([wall], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    return float(tw.item())


def aggregate_value(flop_per_step: float, steps: int, world: int, wall: float) -> float:
    """Whole-job TFLOP/s: every rank runs the same workload (weak scaling, independent replicas)."""
    return flop_per_step * steps * world / wall / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="untimed load before the W warmup steps: an idle MI355X needs ~50 ms under load to leave its clock "
                         "ramp (tools/clock_ramp.py: 54 -> 42 -> 38 -> 36.7 us/step over the first 40 ms)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl")  # RCCL; only used for the timing barrier / max-reduce
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import qutlass_amd
    from qutlass_amd.utils import to_blocked

    # ---- synthetic operands (seed 0, test distribution randn*25, mxfp4_test.py:224-225) ----
    torch.manual_seed(0)
    a = torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0
    b = torch.randn(N, K, dtype=torch.bfloat16, device=dev) * 25.0
    h = hadamard(32, dev)
    alpha = torch.tensor([1.0], device=dev)
    a_q, a_s = qutlass_amd.fusedQuantizeMx(a, h, method="abs_max")
    b_q, b_s = qutlass_amd.fusedQuantizeMx(b, h, method="abs_max")
    a_sf, b_sf = to_blocked(a_s), to_blocked(b_s)
    torch.cuda.synchronize()

    def step():
        return qutlass_amd.matmul_mxf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha)

    # ---- clock ramp (untimed, not part of the W warmup steps): bring the part to its steady clock / power state ----
    ramp_steps = 0
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(200):
            out = step()
        torch.cuda.synchronize()
        ramp_steps += 200

    for _ in range(args.warmup):
        out = step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: EXACTLY `steps` steps, barrier + synchronize on both sides -----------
    stream = torch.cuda.current_stream(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)  # the ops launch on torch's current stream, so these events bracket the kernels
    for _ in range(args.steps):
        out = step()
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch-to-launch duration of the GEMM kernel

    wall = max_over_ranks(wall, dev if world > 1 else None)

    flop_per_step = 2.0 * M * N * K
    value = aggregate_value(flop_per_step, args.steps, world, wall)
    achieved = flop_per_step / (kernel_ms * 1e-3) / 1e12

    result = {
        "metric": "TFLOP/s & %FP4-MFMA-peak, MXFP4 GEMM 4096x4096x4096, 1 MI355X",
        "value": round(value, 2),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "fp4 (e2m1 x e2m1, e8m0 block scales, fp32 accumulate, bf16 out)",
        "data": "synthetic",
        "config": {
            "workload": "matmul_mxf4_bf16_tn 4096x4096x4096, gs=32 e8m0 scales (BASELINE.json configs[1])",
            "operands": "randn*25 bf16, seed 0, fusedQuantizeMx(H32, abs_max) once, to_blocked scales, alpha=1",
            "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
            "clock_ramp": f"{ramp_steps} untimed steps (~{args.ramp_ms:.0f} ms) before the {args.warmup} warmup steps",
            "pct_of_fp4_peak": round(100.0 * value / world / FP4_DENSE_PEAK_TFLOPS, 2),
        },
        "roofline": {
            "bound": "mfma",
            "achieved": round(achieved, 2),
            "peak": FP4_DENSE_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / FP4_DENSE_PEAK_TFLOPS, 4),
            "traffic": None,
            "kernel": "gemm_mx_kernel<GemmCfg<256,256,2,2,4>, SCHED_DEEP>",
            "kernel_us": round(kernel_ms * 1e3, 3),
            "algorithmic_flop_per_launch": flop_per_step,
            "algorithmic_bytes_per_launch": M * K // 2 + N * K // 2 + (M + N) * K // 32 + 2 * M * N,
            "frac_of_sustained_random_operand_mfma_rate": round(achieved / FP4_SUSTAINED_RANDOM_TFLOPS, 4),
        },
    }
    # HBM-side bytes per launch of this kernel from the PMC passes (rocprofv3 cannot wrap the process from inside;
    # tools/pmc_bench.sh runs the two --pmc passes over this same command and leaves the corrected sum here)
    try:
        with open(PMC_TRAFFIC_JSON) as f:
            tj = json.load(f)
        if tj.get("traffic_bytes"):
            result["roofline"]["traffic"] = tj["traffic_bytes"]
            result["roofline"]["traffic_source"] = "profiles/pmc_bench_r1.json (FETCH_SIZE x2 + WRITE_SIZE, per launch)"
    except (OSError, ValueError):
        pass

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cb, ref = cpu_baseline(a_q, a_s, b_q, b_s, rows=CPU_ROWS)
            result["cpu_baseline"] = cb
            # parity of the measured op against the same slab (exact bf16 equality, as the reference asserts)
            got = out[:CPU_ROWS].cpu()
            result["config"]["parity_vs_cpu_oracle_slab"] = bool(torch.equal(got, ref))
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1 or args.gpus > 1:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
