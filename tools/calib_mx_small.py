"""Calibration data for the small / mid batch end of the MX GEMMs' dispatch (capi.hip: plan_small and the half-chip rules): the same GEMM under the skinny kernel,
the ring schedule on 64x64 / 64x128 / 128x128 tiles with 1 / 2 / 4 / 8 K ranges, the pipelined 128x128 schedule and the 256x128 tile (lab library, forced variants),
M = 16 ... 1024 against the (N, K) of the reference's benchmark models.      python tools/calib_mx_small.py [mxf4|mxf8 ...] > gpurun_out/calib_mx_small.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import _benchlib as lab
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _timing import graph_us, stream_us

GRAPH = "--stream" not in sys.argv   # default: GPU-only timing through HIP-graph replays; --stream = the Python-in-the-loop protocol of the first calibrations

NK = [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (8192, 8192), (57344, 8192), (8192, 28672), (5120, 5120), (51200, 5120), (5120, 25600), (2048, 2048), (14336, 4096)]
MS = [m for m in [16, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024] if m <= int(os.environ.get("CALIB_MAX_M", "1024"))]
if os.environ.get("CALIB_MS"): MS = [int(v) for v in os.environ["CALIB_MS"].split(",")]   # e.g. CALIB_MS=1,4,8,16,32 for the decode end
# (name, gemm_variant, splitk_force)
CAND = [("auto", 0, 0), ("skinny", 60, 0), ("r64", 70, 1), ("r64/2", 70, 2), ("r64/4", 70, 4), ("r64/8", 70, 8), ("r64x128", 72, 1), ("r64x128/2", 72, 2), ("r64x128/4", 72, 4),
        ("r128", 73, 1), ("r128/2", 73, 2), ("r128/4", 73, 4), ("p128", 24, 0), ("256x128", 58, 0)]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    for fmt in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["mxf4", "mxf8"]):
        epb = 1 if fmt == "mxf8" else 2
        fn = lab.matmul_mxf8_bf16_tn if fmt == "mxf8" else lab.matmul_mxf4_bf16_tn
        print("# %s: M N K | %s" % (fmt, " ".join(n for n, _, _ in CAND)), flush=True)
        for (n, k) in NK:
            b = torch.randint(0, 256, (n, k // epb), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(118, 126, (pad(n) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            for m in MS:
                a = torch.randint(0, 256, (m, k // epb), dtype=torch.uint8, device=dev, generator=g)
                if fmt == "mxf8": a &= 0x77
                sa = torch.randint(118, 126, (pad(m) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
                fl = 2.0 * m * n * k
                reps = max(8, min(300, int(10e-3 / max(fl / 3e15, 5e-6))))
                res = []
                for name, var, sf in CAND:
                    if (var == 60 and (m > 32 or fmt == "mxf8")) or (var == 58 and m < 256):
                        res.append(float("nan")); continue
                    try:
                        with lab.forced(gemm_variant=var, splitk_force=sf):
                            call = lambda: fn(a, b, sa, sb, alpha)
                            best = graph_us(call, n=max(8, min(40, int(2.5e3 / max(fl / 1.0e15 * 1e6, 5.0))))) if GRAPH else stream_us(call, reps)
                            res.append(best)
                    except Exception:
                        res.append(float("nan"))
                print("%s %5d %6d %6d | %s" % (fmt, m, n, k, " ".join("%8.2f" % r for r in res)), flush=True)


main()
