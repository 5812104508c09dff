#!/usr/bin/env python3
"""Where do the ~2 us per step go that bench.py's wall clock shows over its HIP-event time at the driver's --steps 20?  (687 us of wall for 647 us of kernels.)
Times the same 20-step region of the headline GEMM with (a) torch.cuda.synchronize() as the closing wait, (b) a spin on event.query() before it, and prints the
host-side marks: loop enqueued, last kernel done (event), synchronize returned.  GPU tool, not part of the package."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qutlass_amd  # noqa: E402
from qutlass_amd.utils import to_blocked  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M = N = K = 4096
    torch.manual_seed(0)
    a = torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0
    b = torch.randn(N, K, dtype=torch.bfloat16, device=dev) * 25.0
    h = (torch.tensor([[1.0]], device=dev))
    from bench import hadamard
    h = hadamard(32, dev)
    alpha = torch.tensor([1.0], device=dev)
    a_q, a_s = qutlass_amd.fusedQuantizeMx(a, h, method="abs_max")
    b_q, b_s = qutlass_amd.fusedQuantizeMx(b, h, method="abs_max")
    a_sf, b_sf = to_blocked(a_s), to_blocked(b_s)

    def step():
        return qutlass_amd.matmul_mxf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha)

    t = time.perf_counter()
    while time.perf_counter() - t < 1.0:
        for _ in range(200):
            step()
        torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev)
    steps = 20
    for mode in ("sync", "spin", "sync", "spin", "sync", "spin"):
        rows = []
        for rep in range(30):
            for _ in range(5):
                step()
            torch.cuda.synchronize(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(steps):
                step()
            e1.record(stream)
            t1 = time.perf_counter()
            if mode == "spin":
                while not e1.query():
                    pass
            t2 = time.perf_counter()
            torch.cuda.synchronize(); torch.cuda.synchronize()
            t3 = time.perf_counter()
            rows.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6, (t3 - t0) * 1e6, e0.elapsed_time(e1) * 1e3))
        rows.sort(key=lambda r: r[2])
        med = rows[len(rows) // 2]
        print(f"{mode:5s} median of 30: enqueued {med[0]:7.1f} us | event seen {med[1]:7.1f} | synchronize returned {med[2]:7.1f} | events {med[3]:7.1f} us  "
              f"-> wall/step {med[2] / steps:6.2f}, events/step {med[3] / steps:6.2f}   (min wall/step {rows[0][2] / steps:6.2f})")


if __name__ == "__main__":
    main()
