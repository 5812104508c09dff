"""Calibration data for the tile rules: the same GEMM under every candidate tile / schedule (lab library, forced variants), for the shapes where
the candidates compete (outputs of 0.3 .. 3 rounds of 256x256 tiles).  One line per (format, N, K, M): microseconds per candidate.
    python tools/calib_tiles.py > gpurun_out/calib_tiles.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import _benchlib as lab
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _timing import graph_us, stream_us

GRAPH = "--stream" not in sys.argv   # default: GPU-only timing through HIP-graph replays (tools/_timing.py); --stream = the Python-in-the-loop protocol of profiles/calib_tiles_r3.txt

NK = [(4096, 4096), (6144, 4096), (5120, 5120), (4096, 14336), (8192, 8192), (5120, 25600), (28672, 4096)]
MS = [512, 768, 1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192]
CAND = {"mxf4": [("auto", 0), ("p256", 90), ("het", 98), ("128", 24), ("256x128", 58), ("r128", 73)], "mxf8": [("auto", 0), ("p256", 90), ("het", 98), ("128", 24), ("256x128", 58)],
        "nvf4": [("auto", 0), ("pk256", 42), ("256x128", 40), ("128", 5)]}   # [r4] pk256 = the persistent 256x256 kernel (41 = the per-tile one it replaced)


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    for fmt in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["mxf4", "mxf8", "nvf4"]):
        epb, gs = (1, 32) if fmt == "mxf8" else (2, 16 if fmt == "nvf4" else 32)
        fn = {"mxf4": lab.matmul_mxf4_bf16_tn, "mxf8": lab.matmul_mxf8_bf16_tn, "nvf4": lab.matmul_nvf4_bf16_tn}[fmt]
        opt = "nvf4_variant" if fmt == "nvf4" else "gemm_variant"
        print("# %s: M N K | %s" % (fmt, " ".join(n for n, _ in CAND[fmt])), flush=True)
        for (n, k) in NK:
            b = torch.randint(0, 256, (n, k // epb), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(118, 126, (pad(n) * ((k // gs + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            for m in MS:
                if m * n > 8192 * 8192 or (fmt == "nvf4" and m * n * k > 8192 ** 3): continue
                a = torch.randint(0, 256, (m, k // epb), dtype=torch.uint8, device=dev, generator=g)
                if fmt == "mxf8": a &= 0x77
                sa = torch.randint(118, 126, (pad(m) * ((k // gs + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
                fl = 2.0 * m * n * k
                reps = max(8, min(300, int(20e-3 / max(fl / 3e15, 4e-6))))
                res = []
                for name, var in CAND[fmt]:
                    try:
                        with lab.forced(**{opt: var}):
                            call = lambda: fn(a, b, sa, sb, alpha)
                            res.append(graph_us(call, n=max(4, min(40, int(2.5e3 / max(fl / 1.0e15 * 1e6, 5.0))))) if GRAPH else stream_us(call, reps))
                    except Exception as e:   # a variant the lab dispatch rejects for this shape
                        res.append(float("nan"))
                print("%s %5d %6d %6d | %s" % (fmt, m, n, k, " ".join("%8.2f" % r for r in res)), flush=True)


main()
