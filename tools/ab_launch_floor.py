#!/usr/bin/env python3
"""A/B of what sits BETWEEN two launches of the headline kernel (launch floor 3.3 us of a 33 us step).

Per process (run once per value of HIP_FORCE_DEV_KERNARG by tools/gpu_r3l.sh): the 4096^3 MXFP4 GEMM
  * launched in-stream through the torch op, as bench.py does (the output tensor is allocated per call),
  * launched in-stream through the C ABI into ONE preallocated output,
  * captured 100 x into a HIP graph and replayed,
each for >= 0.25 s of clock ramp first, then three interleaved repetitions of 2000 steps between two HIP events.
Also the floor itself: a 1-workgroup to_blocked of a 128 x 4 scale matrix, in-stream and graph-replayed.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qutlass_amd  # noqa: E402
from qutlass_amd.utils import to_blocked  # noqa: E402
from bench import hadamard  # noqa: E402

M = N = K = 4096


def timed(run, calls_per_run, steps):
    reps = max(1, steps // calls_per_run)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * calls_per_run)


def graph_of(fn, n):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return g


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    a = torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0
    b = torch.randn(N, K, dtype=torch.bfloat16, device=dev) * 25.0
    h = hadamard(32, dev)
    alpha = torch.tensor([1.0], device=dev)
    a_q, a_s = qutlass_amd.fusedQuantizeMx(a, h, method="abs_max")
    b_q, b_s = qutlass_amd.fusedQuantizeMx(b, h, method="abs_max")
    a_sf, b_sf = to_blocked(a_s), to_blocked(b_s)
    small = torch.randint(0, 255, (128, 4), dtype=torch.uint8, device=dev)

    def op():
        return qutlass_amd.matmul_mxf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha)

    def floor():
        return to_blocked(small)

    g_op = graph_of(op, 100)
    g_floor = graph_of(floor, 200)
    variants = {
        "gemm_in_stream_torch_op": (op, 1),
        "gemm_graph_100": (g_op.replay, 100),
        "floor_in_stream": (floor, 1),
        "floor_graph_200": (g_floor.replay, 200),
    }
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(200):
            op()
        torch.cuda.synchronize()
    res = {k: [] for k in variants}
    for _ in range(3):
        for name, (run, per) in variants.items():
            for _ in range(3):
                run()
            res[name].append(round(timed(run, per, 2000 if name.startswith("gemm") else 4000), 3))
    print(json.dumps({"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"), "us_per_call": res}), flush=True)


if __name__ == "__main__":
    main()
