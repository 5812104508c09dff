"""Small-batch MXFP4 shapes: the product's plan (auto) beside the forced in-workgroup K-split tiles of csrc/gemm_mx_ks.hip.h (lab variants 561 = 32x32, 562 = 32x64,
563 = 64x32, 564 = 64x64), GPU-only timing (HIP-graph replays), + a bit-exactness check of each forced variant against auto on exact-regime operands.
    python tools/calib_ks.py > gpurun_out/calib_ks.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us


def main():
    dev = torch.device("cuda:0")
    variants = [int(v) for v in os.environ.get("KS_VARIANTS", "0,561,562,563,564").split(",")]
    ms = [int(v) for v in os.environ.get("KS_M", "9,16,24,32,48,64,96,128,192,256").split(",")]
    nks = [tuple(int(d) for d in s.split("x")) for s in os.environ.get("KS_NK", "4096x4096,14336x4096,4096x14336,6144x4096,8192x8192,5120x5120,28672x4096,1024x4096,2048x2048").split(",")]
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    print("# us per launch (HIP-graph replays), columns = gemm_variant " + " ".join(str(v) for v in variants) + " | best forced / auto | forced variants equal auto (exact-regime operands)")
    for (n, k) in nks:
        for m in ms:
            a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
            b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
            cb = (k // 32 + 3) // 4 * 4
            sa = torch.randint(125, 129, (pad(m) * cb,), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(125, 129, (pad(n) * cb,), dtype=torch.uint8, device=dev, generator=g)
            t, outs = {}, {}
            for v in variants:
                with lab.forced(gemm_variant=v):
                    outs[v] = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
                    t[v] = min(graph_us(lambda: lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha), n=40) for _ in range(3))
            torch.cuda.synchronize()
            eq = all(torch.equal(outs[v].view(torch.int16), outs[variants[0]].view(torch.int16)) for v in variants[1:])
            forced = {v: t[v] for v in variants if v != 0}
            bv = min(forced, key=forced.get) if forced else 0
            print("N=%-6d K=%-6d M=%-4d | %s | %d %.2f | %s" % (n, k, m, " ".join("%7.2f" % t[v] for v in variants), bv, (forced[bv] / t[0]) if forced and 0 in t else 0.0, "equal" if eq else "DIFFER"), flush=True)


main()
