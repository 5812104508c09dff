"""matmul_ada_mxf4_bf16_tn (row-major scale operands) on decode shapes: the LDS-free split-K kernel (60), the 64x64 ring (70), the one-shot / wave-owned-ring kernel (568,
csrc/gemm_mx_os.hip.h) and the product rule (0), GPU-only timing (HIP-graph replays) + equality of the four results.   ADA_NK=4096x8192,... python tools/calib_ada.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import _benchlib as lab
from _timing import graph_us
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
alpha = torch.ones(1, device=dev)
print("# matmul_ada_mxf4_bf16_tn (row-major scales), us per launch (HIP-graph replays): forced 60 = LDS-free split-K kernel, 70 = 64x64 ring, 568 = one-shot, 0 = the product rule")
NK = [tuple(int(d) for d in x.split("x")) for x in os.environ.get("ADA_NK", "4096x4096,2048x2048,8192x4096,1024x4096,6144x4096").split(",")]
for (n, k) in NK:
    for m in [int(v) for v in os.environ.get("ADA_M", "1,8,16,32,64").split(",")]:
        a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
        b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
        sa = torch.randint(125, 129, (m, k // 32), dtype=torch.uint8, device=dev, generator=g)
        sb = torch.randint(125, 129, (n, k // 32), dtype=torch.uint8, device=dev, generator=g)
        t, outs = {}, {}
        VS = [int(v) for v in os.environ.get("ADA_VARIANTS", "60,70,568,569,0").split(",")]   # (570 = 64x32 tiles)
        for v in VS:
            with lab.forced(gemm_variant=v):
                outs[v] = lab.matmul_ada_mxf4_bf16_tn(a, b, sa, sb, alpha)
                t[v] = min(graph_us(lambda: lab.matmul_ada_mxf4_bf16_tn(a, b, sa, sb, alpha), n=40) for _ in range(3))
        eq = all(torch.equal(outs[v].view(torch.int16), outs[VS[0]].view(torch.int16)) for v in VS[1:])
        print("N=%-6d K=%-6d M=%-4d | %s | %s" % (n, k, m, " ".join("%d: %6.2f" % (v, t[v]) for v in VS), "equal" if eq else "DIFFER"), flush=True)
