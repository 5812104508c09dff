"""Small-batch NVFP4 shapes: the product's plan (0) beside the split-K skinny kernel (3) and the wave-owned kernel with 32 / 16 columns per workgroup (46 / 47,
csrc/gemm_nvf4_os.hip.h) and its decode forms on the 16x16x32 MFMA (48 = 16x16 tiles, 49 = 32x16, 50 / 51 / 52 = 16 rows x 32 / 48 / 56 columns, 53 = 56 columns with A rows
0 ... 7 only), GPU-only timing (HIP-graph replays) + equality of the results (exact-regime scale bytes).   NV_VARIANTS=0,46,48 NV_M=1,16 NV_NK=4096x4096,... python tools/calib_nvos.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us
from qutlass_amd.utils import to_blocked

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
alpha = torch.ones(1, device=dev)
variants = [int(v) for v in os.environ.get("NV_VARIANTS", "0,3,46,47").split(",")]
ms = [int(v) for v in os.environ.get("NV_M", "1,8,16,32,64,128").split(",")]
nks = [tuple(int(d) for d in s.split("x")) for s in os.environ.get("NV_NK", "4096x4096,2048x2048,6144x4096,8192x4096,1024x4096,4096x8192,4096x14336,8192x8192,14336x4096").split(",")]
print("# matmul_nvf4_bf16_tn, us per launch (HIP-graph replays), columns = nvf4_variant " + " ".join(str(v) for v in variants) + " | best forced / auto | results equal")
e4 = torch.float8_e4m3fn
for (n, k) in nks:
    for m in ms:
        a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
        b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
        sa = to_blocked(torch.randint(0x30, 0x48, (m, k // 16), dtype=torch.uint8, device=dev, generator=g).view(e4))
        sb = to_blocked(torch.randint(0x30, 0x48, (n, k // 16), dtype=torch.uint8, device=dev, generator=g).view(e4))
        t, outs = {}, {}
        for v in variants:
            with lab.forced(nvf4_variant=v):
                outs[v] = lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha)
                t[v] = min(graph_us(lambda: lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha), n=40) for _ in range(3))
        eq = all(torch.equal(outs[v].view(torch.int16), outs[variants[0]].view(torch.int16)) for v in variants[1:])
        forced = {v: t[v] for v in variants if v != 0}
        bv = min(forced, key=forced.get)
        print("N=%-6d K=%-6d M=%-4d | %s | %d %.2f | %s" % (n, k, m, " ".join("%7.2f" % t[v] for v in variants), bv, forced[bv] / t[0] if 0 in t else 0.0, "equal" if eq else "DIFFER"), flush=True)
