"""matmul_nvf4_bf16_tn: the persistent kernel (gemm_nvf4_pk.hip.h) against the per-tile kernel it replaces and torch's bf16 GEMM (hipBLASLt), one box, interleaved,
GPU-only timing (HIP-graph replays, tools/_timing.py), operands = quantised Gaussians (fusedQuantizeNv):
    python tools/ab_nvpk.py [--trace] [--shapes llama|dip|all] > gpurun_out/ab_nvpk.txt
columns: 41 = per-tile 256x256 kernel (round 3)   42 = persistent, whole tiles (balanced rounds)   43 = persistent + stream-K over the last round
         0 = the product's auto rule   bf16 = torch.matmul on bf16 operands of the same shape.  `same` = variants 41 / 42 / 43 / 0 returned the same bytes.
--trace: stage trace of workgroup 0 (lab variant 44): shader cycles and wall time per K stage of 256 elements (256 MFMAs of 32 cycles = 8192 cycles at best)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
import qutlass_amd as q
from qutlass_amd.utils import to_blocked
from _timing import graph_us

LLAMA = [(8192, 8192, 8192), (4096, 4096, 4096), (4096, 14336, 4096), (4096, 28672, 4096), (2048, 28672, 4096), (512, 28672, 4096), (4096, 6144, 4096), (2048, 6144, 4096)]
DIP = [(6144, 4096, 4096), (4096, 5120, 5120), (5120, 4096, 4096), (3072, 6144, 4096), (3072, 28672, 8192), (4096, 5120, 4096), (2560, 4096, 4096)]


def operands(m, n, k, dev):
    torch.manual_seed(m + n + k)
    h16 = torch.eye(16, dtype=torch.bfloat16, device=dev)
    gs = torch.tensor([1.0], device=dev)
    def mk(r):
        x = torch.randn(r, k, dtype=torch.bfloat16, device=dev)
        xq, xs = q.fusedQuantizeNv(x, h16, gs)
        return xq, to_blocked(xs), x
    return mk(m), mk(n)


def trace(m, n, k, dev):
    (a, sa, _), (b, sb, _) = operands(m, n, k, dev)
    alpha = torch.ones(1, device=dev)
    buf = torch.zeros(4096, dtype=torch.int32, device=dev)
    lab.load().qutlass_amd_debug_set_trace_buffer(buf.data_ptr())
    with lab.forced(nvf4_variant=44):
        for _ in range(20): lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha)
        torch.cuda.synchronize()
    lab.load().qutlass_amd_debug_set_trace_buffer(None)
    t = buf.cpu().numpy().astype("uint32")
    cnt = int(t[0])
    cyc = t[2:2 + 2 * cnt:2].astype("int64"); wall = t[3:3 + 2 * cnt:2].astype("int64")
    dc = (cyc[1:] - cyc[:-1]) % (1 << 32); dw = (wall[1:] - wall[:-1]) % (1 << 32)
    kt = k // 256
    print(f"# trace {m}x{n}x{k}: {cnt} marks, {kt} stages per tile")
    for i in range(0, min(len(dc), 4 * kt), 1):
        tag = "  <- tile boundary (epilogue inside)" if (i + 1) % kt == 0 else ""
        if i < 6 or (i + 1) % kt == 0 or i % kt == 0:
            print(f"#   stage {i:3d}: {int(dc[i]):6d} cycles  {int(dw[i]) * 10:6d} ns  -> {dc[i] / max(1, dw[i] * 10):.2f} GHz{tag}")
    steady = [int(dc[i]) for i in range(len(dc)) if (i + 1) % kt != 0 and i % kt != 0 and i >= 2]
    steady_w = [int(dw[i]) for i in range(len(dw)) if (i + 1) % kt != 0 and i % kt != 0 and i >= 2]
    if steady:
        print(f"#   steady stages: median {sorted(steady)[len(steady) // 2]} cycles, {sorted(steady_w)[len(steady_w) // 2] * 10} ns; MFMA-bound 8192 cycles", flush=True)
    ho = t[1024:1024 + 2 * min(cnt, 120)].astype("int64")
    wait = [int((ho[2 * j + 1] - ho[2 * j]) % (1 << 32)) for j in range(len(ho) // 2) if ho[2 * j] or ho[2 * j + 1]]
    into = [int((ho[2 * j] - cyc[j]) % (1 << 32)) for j in range(min(len(ho) // 2, len(cyc))) if ho[2 * j]]
    if wait:
        print(f"#   hand-off (s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier at k-step 10): median {sorted(wait)[len(wait) // 2]} cycles, min {min(wait)}, max {max(wait)}; "
              f"reached {sorted(into)[len(into) // 2]} cycles into the stage (10 k-steps of 16 MFMAs = 5120 at best)", flush=True)


def main():
    dev = torch.device("cuda:0")
    which = "all"
    if "--shapes" in sys.argv: which = sys.argv[sys.argv.index("--shapes") + 1]
    shapes = {"none": [], "llama": LLAMA, "dip": DIP, "all": LLAMA + DIP, "prof": [(8192, 8192, 8192), (4096, 28672, 4096), (4096, 4096, 4096)]}[which]
    if which == "prof":   # under rocprofv3: the product's own choice only, plain launches (tools/gpu_session.sh profnv)
        alpha = torch.ones(1, device=dev)
        for (m, n, k) in shapes:
            (a, sa, _), (b, sb, _) = operands(m, n, k, dev)
            for _ in range(60): q.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha)
            torch.cuda.synchronize()
        return
    if "--trace" in sys.argv:
        for s in [(8192, 8192, 8192), (4096, 28672, 4096)]: trace(*s, dev)
    alpha = torch.ones(1, device=dev)
    print("%-20s %9s %9s %9s %9s %9s | %7s %7s %7s | TFLOP/s: %6s %6s %6s  same" % ("M x N x K", "41 us", "42 us", "43 us", "auto us", "bf16 us", "42/41", "43/41", "auto/bf", "41", "auto", "bf16"))
    for (m, n, k) in shapes:
        (a, sa, xa), (b, sb, xb) = operands(m, n, k, dev)
        outs, t = {}, {}
        for v in (41, 42, 43, 0):
            with lab.forced(nvf4_variant=v):
                outs[v] = lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha)
        same = all(torch.equal(outs[41].view(torch.int16), outs[v].view(torch.int16)) for v in (42, 43, 0))
        nrep = max(4, min(40, int(4000 / max(1.0, 2.0 * m * n * k / 1.4e9))))
        for rnd in range(2):   # interleaved: every candidate twice, best kept
            for v in (41, 42, 43, 0):
                with lab.forced(nvf4_variant=v):
                    us = graph_us(lambda: lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha), n=nrep)
                t[v] = min(t.get(v, 1e9), us)
            us = graph_us(lambda: torch.matmul(xa, xb.t()), n=nrep)
            t["bf"] = min(t.get("bf", 1e9), us)
        fl = 2.0 * m * n * k / 1e6
        print("%-20s %9.2f %9.2f %9.2f %9.2f %9.2f | %7.3f %7.3f %7.3f | %15.0f %6.0f %6.0f  %s" % (f"{m}x{n}x{k}", t[41], t[42], t[43], t[0], t["bf"], t[42] / t[41], t[43] / t[41],
              t[0] / t["bf"], fl / t[41], fl / t[0], fl / t["bf"], same), flush=True)


main()
