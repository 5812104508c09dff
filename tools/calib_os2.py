"""The wave-owned small-batch kernel's two new axes (csrc/gemm_mx_os.hip.h, round 6 third session): MXFP8 (EBITS = 8) and 64-row tiles (TM = 64).  Per shape: the product
rule (0) beside the forced tiles 568 = 32x32, 569 = 32x16, 570 = 64x32, GPU-only timing (HIP-graph replays), and each forced variant's output against the rule's
(MXFP4: bit-equal on exact-regime operands; MXFP8: worst difference in bf16 ulps -- the summation order differs).
    OS2_FMT=8 python tools/calib_os2.py > gpurun_out/calib_os2_fp8.txt        OS2_FMT=4 OS2_M=64,96,128,160,192,256 python tools/calib_os2.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us


def ulps(x, y):
    """difference of two bf16 tensors in units of the larger one's last place"""
    xi, yi = x.view(torch.int16).to(torch.int32), y.view(torch.int16).to(torch.int32)
    key = lambda v: torch.where(v < 0, -(v & 0x7fff), v)   # sign-magnitude -> monotone integers
    return int((key(xi) - key(yi)).abs().max())


def main():
    dev = torch.device("cuda:0")
    fmt = int(os.environ.get("OS2_FMT", "8"))
    variants = [int(v) for v in os.environ.get("OS2_VARIANTS", "0,568,569,570").split(",")]
    ms = [int(v) for v in os.environ.get("OS2_M", "1,16,32,64,96,128,192,256").split(",")]
    nks = [tuple(int(d) for d in s.split("x")) for s in os.environ.get("OS2_NK", "4096x4096,2048x2048,8192x4096,4096x8192,4096x14336,14336x4096,6144x4096,1024x4096,8192x8192").split(",")]
    a5 = os.environ.get("OS2_A5", "0") == "1"
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    call = lab.matmul_mxf4_bf16_tn if fmt == 4 else (lambda *x: lab.matmul_mxf8_bf16_tn_fmt(*x, a_format=1)) if a5 else lab.matmul_mxf8_bf16_tn
    print("# MXFP%d%s: us per launch (HIP-graph replays), columns = gemm_variant %s | best forced / rule | worst bf16-ulp difference of a forced variant from the rule's output" % (fmt, " (e5m2 A)" if a5 else "", " ".join(str(v) for v in variants)))
    for (n, k) in nks:
        for m in ms:
            kb = k // 2 if fmt == 4 else k
            hi = 256 if fmt == 4 else 120   # (fp8: bytes 0 ... 119 -- finite, positive, wide exponent range)
            a = torch.randint(0, hi, (m, kb), dtype=torch.uint8, device=dev, generator=g)
            b = torch.randint(0, hi, (n, kb), dtype=torch.uint8, device=dev, generator=g)
            cb = (k // 32 + 3) // 4 * 4
            sa = torch.randint(125, 129, (pad(m) * cb,), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(125, 129, (pad(n) * cb,), dtype=torch.uint8, device=dev, generator=g)
            t, outs = {}, {}
            for v in variants:
                if 571 <= v <= 575 and ((m + 15) // 16) * -(-n // {571: 16, 572: 32, 573: 48, 574: 56, 575: 64}[v]) > 2 * 256:
                    continue   # (the decode form: at most two workgroups per CU)
                if v in (568, 569, 570) and ((m + (63 if v == 570 else 31)) // (64 if v == 570 else 32)) * ((n + (15 if v == 569 else 31)) // (16 if v == 569 else 32)) > 4 * 256:
                    continue   # more than four rounds of tiles: not a candidate
                with lab.forced(gemm_variant=v):
                    outs[v] = call(a, b, sa, sb, alpha)
                    t[v] = min(graph_us(lambda: call(a, b, sa, sb, alpha), n=40) for _ in range(3))
            torch.cuda.synchronize()
            worst = max([ulps(outs[v], outs[variants[0]]) for v in outs if v != variants[0]] or [0])
            forced = {v: t[v] for v in t if v != 0}
            bv = min(forced, key=forced.get) if forced else 0
            print("N=%-6d K=%-6d M=%-4d | %s | %d %.2f | %d ulp" % (n, k, m, " ".join(("%7.2f" % t[v]) if v in t else "      -" for v in variants), bv, (forced[bv] / t[0]) if forced and 0 in t else 0.0, worst), flush=True)


main()
