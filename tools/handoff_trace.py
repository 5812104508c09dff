#!/usr/bin/env python3
"""What does the K loop's hand-off cost?  (persistent MXFP4 kernel, lab variant 91: every wave of workgroup 0 stamps the shader clock before the hand-off's waits, behind
`s_waitcnt lgkmcnt(0)` (its LDS reads of the buffer are back), behind `s_waitcnt vmcnt(0)` (its own LDS-DMA pieces of the next stage have landed) and behind the barrier.)
A K-loop stage is 64 MFMAs = 2 048 cycles and takes ~2 370 (8 workgroups) .. 2 510 (256); the ISA has only ~2 other instructions per MFMA slot, all of which hide behind a
32-cycle MFMA (tests/native/issue_ubench*.hip) -- so the rest is this.      python tools/handoff_trace.py > gpurun_out/handoff_trace.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab  # noqa: E402
from final_stage_contention import operands  # noqa: E402


def run(grid, k, dev):
    n, tiles = 4096, 2 * grid
    m = tiles // 16 * 256
    (a, sa), (b, sb) = operands(m, n, k, dev)
    alpha = torch.ones(1, device=dev)
    buf = torch.zeros(8192, dtype=torch.int32, device=dev)
    with lab.forced(gemm_variant=91, deepp_grid=grid):
        for _ in range(20):
            lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
        torch.cuda.synchronize()
        lab.load().qutlass_amd_debug_set_trace_buffer(buf.data_ptr())
        try:
            res = []
            for _ in range(6):
                buf.zero_()
                lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
                torch.cuda.synchronize()
                t = buf.cpu().numpy().astype("int64")
                res.append(t[3072:3072 + 4096].reshape(4, 256, 4)[:, :60])
        finally:
            lab.load().qutlass_amd_debug_set_trace_buffer(None)
    r = res[len(res) // 2]
    nst = int((r[0, :, 0] != 0).sum())
    d = lambda x, y: ((r[:, 1:nst - 1, y] - r[:, 1:nst - 1, x]) % (1 << 32))          # (skip the first and the last recorded hand-off)
    per_stage = ((r[:, 2:nst - 1, 0] - r[:, 1:nst - 2, 0]) % (1 << 32))
    print(f"grid {grid:3d}, K = {k}: {nst} hand-offs recorded; cycles per wave (median over hand-offs | max): "
          + " | ".join(f"wave {w}: LDS reads {int(np.median(d(0, 1)[w]))} / DMA landed {int(np.median(d(1, 2)[w]))} / barrier {int(np.median(d(2, 3)[w]))} (sum {int(np.median(d(0, 3)[w]))}, max {int(d(0, 3)[w].max())})"
                      for w in range(4)) + f" | hand-off to hand-off {int(np.median(per_stage))} cycles", flush=True)


def main():
    dev = torch.device("cuda:0")
    for k in (4096, 8192):
        for g in (8, 64, 256):
            run(g, k, dev)


if __name__ == "__main__":
    main()
