"""GPU-only timing (HIP-graph replays of 40 launches) of the MX GEMMs' small-output plan against the round-2 rule's choice and neighbouring candidates, for the shapes
the fitted-model correction of capi.hip (plan_small) changes.      python tools/instream_mx_check.py > gpurun_out/instream_mx_check.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import _benchlib as lab

# (fmt, M, N, K, [(name, gemm_variant, splitk_force)])  -- the first candidate after "plan" is what the tile-count rule alone picks
R64, R64x128, R128 = 70, 72, 73
CASES = [("mxf4", 96, 5120, 25600, [("r64", R64, 1), ("r128/4", R128, 4), ("r64x128/2", R64x128, 2)]), ("mxf4", 128, 5120, 25600, [("r64", R64, 1), ("r128/4", R128, 4)]),
         ("mxf4", 96, 8192, 28672, [("r64", R64, 1), ("r128/4", R128, 4), ("r64x128/2", R64x128, 2)]), ("mxf4", 128, 8192, 28672, [("r64", R64, 1), ("r128/4", R128, 4)]),
         ("mxf4", 64, 8192, 28672, [("r64/2", R64, 2), ("r64x128/4", R64x128, 4)]), ("mxf4", 96, 57344, 8192, [("p256", 90, 0), ("p128", 24, 0), ("r128", R128, 1)]),
         ("mxf4", 128, 51200, 5120, [("p256", 90, 0), ("p128", 24, 0)]), ("mxf4", 128, 4096, 14336, [("r64/2", R64, 2), ("r128/4", R128, 4), ("r64x128/4", R64x128, 4)]),
         ("mxf8", 96, 5120, 25600, [("r64", R64, 1), ("r128/4", R128, 4)]), ("mxf8", 192, 4096, 14336, [("r64", R64, 1), ("r128/4", R128, 4), ("r64x128/2", R64x128, 2)]),
         ("mxf8", 256, 4096, 14336, [("r64", R64, 1), ("r128/4", R128, 4)]), ("mxf8", 128, 8192, 28672, [("r64", R64, 1), ("r128/4", R128, 4)]),
         ("mxf8", 256, 8192, 28672, [("r64x128", R64x128, 1), ("r128/2", R128, 2)]), ("mxf8", 384, 5120, 25600, [("r64x128", R64x128, 1), ("r128/2", R128, 2)]),
         ("mxf8", 16, 8192, 8192, [("r64/2", R64, 2), ("r64x128/4", R64x128, 4)]), ("mxf8", 96, 57344, 8192, [("p256", 90, 0), ("p128", 24, 0)]),
         ("mxf8", 384, 4096, 14336, [("r64x128", R64x128, 1), ("r128/2", R128, 2)]), ("mxf8", 512, 4096, 14336, [("r64x128", R64x128, 1), ("r128/2", R128, 2), ("r128", R128, 1)])]


def graph_us(call):
    call(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): call()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(40): call()
    for _ in range(4): gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    return best


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    for (fmt, m, n, k, cands) in CASES:
        epb = 1 if fmt == "mxf8" else 2
        fn = lab.matmul_mxf8_bf16_tn if fmt == "mxf8" else lab.matmul_mxf4_bf16_tn
        a = torch.randint(0, 256, (m, k // epb), dtype=torch.uint8, device=dev, generator=g)
        if fmt == "mxf8": a &= 0x77
        b = torch.randint(0, 256, (n, k // epb), dtype=torch.uint8, device=dev, generator=g)
        sa = torch.randint(118, 126, (pad(m) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
        sb = torch.randint(118, 126, (pad(n) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
        res = []
        for name, var, sf in [("plan", 0, 0)] + cands:
            try:
                with lab.forced(gemm_variant=var, splitk_force=sf):
                    res.append((name, graph_us(lambda: fn(a, b, sa, sb, alpha))))
            except Exception:
                res.append((name, float("nan")))
        print("%s %5d %6d %6d | " % (fmt, m, n, k) + "  ".join("%s %.2f" % r for r in res) + "   | rule's choice / plan = %.3f" % (res[1][1] / res[0][1]), flush=True)


main()
