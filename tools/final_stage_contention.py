#!/usr/bin/env python3
"""Is the final stage of the persistent MXFP4 kernel slow because every workgroup of the chip stores its tile at the same moment?

What is known: the last K stage of a tile carries the epilogue; leaving half of its stores out saves ~1.7 us per tile (4096^3, C3 and 8192^3 alike,
profiles/native_r3_alpha1_and_half_store_ablation.log), leaving the alpha multiplies out saves nothing -- so the stage is bound by the stores, not by instruction issue.
All workgroups of a persistent launch walk their tiles in lockstep, so the chip writes 32 MiB in one burst per round.  If the cost is that burst (back-pressure from the
memory side), de-phasing the workgroups would recover most of it on multi-round shapes (C3: 4 tiles per workgroup x ~3 us of 110); if it is the per-CU store path, it would not.

This tool answers that with the stage trace of workgroup 0 (lab variant 91: gemm_mx_deepp with TRACE, marks = {entry, first stage read, then per tile: end of the K loop, end of
the final stage}) at three grid sizes with the SAME work per workgroup (two tiles of 256 x 256 x K each): 8, 64 and 256 workgroups.
    python tools/final_stage_contention.py > gpurun_out/final_stage_contention.txt
Reads: if `final` grows from 8 to 256 workgroups while `kloop` does not, the burst is the cost (round 4: it does not -- 8 400 cycles with 8 workgroups).  The per-8-slot marks inside the
final stage (added after that run, not yet run themselves) say whether the time is spread evenly (a throughput bound) or piles up behind particular slots."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab  # noqa: E402
import qutlass_amd as q  # noqa: E402
from qutlass_amd.utils import to_blocked  # noqa: E402


def operands(m, n, k, dev):
    torch.manual_seed(m + n + k)
    h = torch.eye(32, dtype=torch.bfloat16, device=dev)
    def mk(r):
        x = torch.randn(r, k, dtype=torch.bfloat16, device=dev) * 25.0
        xq, xs = q.fusedQuantizeMx(x, h, method="abs_max")
        return xq, to_blocked(xs)
    return mk(m), mk(n)


def trace(grid, k, dev):
    n = 4096                                   # 16 tile columns
    tiles = 2 * grid                           # two tiles per workgroup
    m = tiles // 16 * 256
    (a, sa), (b, sb) = operands(m, n, k, dev)
    alpha = torch.ones(1, device=dev)
    buf = torch.zeros(8192, dtype=torch.int32, device=dev)   # (the lab kernel also stamps its hand-offs at dbg[3072 ..], tools/handoff_trace.py)
    rows = []
    with lab.forced(gemm_variant=91, deepp_grid=grid):
        for _ in range(20):                    # clock ramp
            lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
        torch.cuda.synchronize()
        lab.load().qutlass_amd_debug_set_trace_buffer(buf.data_ptr())
        try:
            for _ in range(8):
                buf.zero_()
                lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
                torch.cuda.synchronize()
                t = buf.cpu().numpy().astype("uint32")
                cnt = int(t[0])
                cyc = t[2:2 + 2 * cnt:2].astype("int64"); wall = t[3:3 + 2 * cnt:2].astype("int64")
                dc = (cyc[1:] - cyc[:-1]) % (1 << 32); dw = ((wall[1:] - wall[:-1]) % (1 << 32)) * 10
                # marks: 0 entry, 1 first stage in registers, then (end of K loop, end of final stage) per tile
                fs = t[1024:1024 + 64].astype("int64").reshape(4, 16)   # marks inside the first four final stages: slots 0, 8 ... (mark k = slot 8 k), mark 15 = the end
                def deltas(r):
                    ks = [k for k in range(15) if r[k]] + [15]
                    return [int((r[b] - r[a]) % (1 << 32)) for a, b in zip(ks[:-1], ks[1:])]
                rows.append((cnt, [int(x) for x in dc], [int(x) for x in dw], [deltas(fs[f]) for f in range(4) if fs[f][0]]))
        finally:
            lab.load().qutlass_amd_debug_set_trace_buffer(None)
    rows.sort(key=lambda r: sum(r[2]))
    cnt, dc, dw, fsd = rows[len(rows) // 2]
    names = ["prologue"] + [f"tile{i // 2} {'kloop' if i % 2 == 0 else 'final'}" for i in range(len(dc) - 1)]
    print(f"grid {grid:3d} ({m} x {n} x {k}, {tiles} tiles, {cnt} marks): " + " | ".join(f"{nm} {c} cyc {w} ns" for nm, c, w in zip(names, dc, dw)), flush=True)
    if os.environ.get("QAMD_FS_TRACE_SLOTS"):   # a lab side build with -DQAMD_FS_TRACE_SLOTS: cycles of every slot of the first two last stages
        t = buf.cpu().numpy().astype("int64")
        for f in range(2):
            m = t[2048 + 128 * f:2048 + 128 * f + 128]
            n = int((m != 0).sum())
            print(f"          final stage {f}, cycles per slot: " + " ".join(str(int((m[i + 1] - m[i]) % (1 << 32))) for i in range(n - 1)), flush=True)
    for f, d in enumerate(fsd):
        print(f"          final stage {f}: cycles per 8 slots (256 at best with an MFMA in each) " + " ".join(str(x) for x in d[:-1]) + f" | behind the last mark {d[-1]}", flush=True)
    return dict(zip(names, dw))


def main():
    dev = torch.device("cuda:0")
    for k in (4096, 1024):
        res = {g: trace(g, k, dev) for g in (8, 64, 256)}
        for key in ("tile0 kloop", "tile0 final", "tile1 kloop", "tile1 final"):
            if all(key in r for r in res.values()):
                print(f"K = {k}: {key:12s} ns at 8 / 64 / 256 workgroups: " + " / ".join(str(res[g][key]) for g in (8, 64, 256)))


if __name__ == "__main__":
    main()
