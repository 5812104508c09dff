"""Calibration data for the small / mid batch end of matmul_nvf4_bf16_tn's tile rule: the same GEMM under the skinny split-K kernel, 64x64, 128x64 and
128x128 tiles (lab library, forced variants) for M = 16 ... 1024 against the (N, K) of the reference's benchmark models.
    python tools/calib_nv_small.py > gpurun_out/calib_nv_small.txt"""
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
CAND = [("auto", 0), ("skinny", 3), ("64x64", 7), ("128x64", 6), ("128x128", 5), ("256x128", 40), ("128x128/2", 112), ("128x128/4", 114), ("128x128/8", 118), ("128x64/2", 122),
        ("128x64/4", 124), ("64x64/2", 132), ("64x64/4", 134)]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    print("# nvf4: M N K | %s" % " ".join(n for n, _ in CAND), flush=True)
    for (n, k) in NK:
        b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
        sb = torch.randint(118, 126, (pad(n) * ((k // 16 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
        for m in MS:
            a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
            sa = torch.randint(118, 126, (pad(m) * ((k // 16 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            fl = 2.0 * m * n * k
            reps = max(8, min(300, int(15e-3 / max(fl / 1e15, 6e-6))))
            res = []
            ref = lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha).float()
            worst = 0.0
            for name, var in CAND:
                if (var == 3 and m > 256) or (var == 40 and m < 256):
                    res.append(float("nan")); continue
                try:
                    with lab.forced(nvf4_variant=var):
                        call = lambda: lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha)
                        if var >= 100:   # split-K against the single pass: equal up to the bf16 rounding of a differently ordered fp32 sum
                            worst = max(worst, float(((call().float() - ref).abs() / (ref.abs() + 1.0)).max()))
                        best = graph_us(call, n=max(8, min(40, int(2.5e3 / max(fl / 1.0e15 * 1e6, 5.0))))) if GRAPH else stream_us(call, reps)
                        res.append(best)
                except Exception:
                    res.append(float("nan"))
            print("nvf4 %5d %6d %6d | %s" % (m, n, k, " ".join("%8.2f" % r for r in res)) + "  | split rel err %.1e" % worst, flush=True)


main()
