"""Calibration data for the small / mid batch end of matmul_nvf4_bf16_tn's tile rule: the same GEMM under the skinny split-K kernel, 64x64, 128x64 and
128x128 tiles (lab library, forced variants) for M = 16 ... 1024 against the (N, K) of the reference's benchmark models.
    python tools/calib_nv_small.py > gpurun_out/calib_nv_small.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import _benchlib as lab

NK = [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (8192, 8192), (57344, 8192), (8192, 28672), (5120, 5120), (51200, 5120), (5120, 25600), (2048, 2048), (14336, 4096)]
MS = [16, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024]
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
                        for _ in range(max(3, reps // 4)): call()
                        torch.cuda.synchronize()
                        best = 1e9
                        for _ in range(2):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            for _ in range(reps): call()
                            e1.record(); torch.cuda.synchronize()
                            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
                        res.append(best)
                except Exception:
                    res.append(float("nan"))
            print("nvf4 %5d %6d %6d | %s" % (m, n, k, " ".join("%8.2f" % r for r in res)) + "  | split rel err %.1e" % worst, flush=True)


main()
