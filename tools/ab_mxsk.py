"""matmul_mxf4_bf16_tn / matmul_mxf8_bf16_tn on outputs of 1.1 ... 5.3 rounds of 256x256 tiles: the persistent kernel in balanced rounds (90), the heterogeneous
launch (98: residual tiles as 128x128 quarter tiles), the stream-K form (89: tiles of the part-filled round cut along K, fp32 parts parked in scratch) and the
product's own choice (0), one box, interleaved, GPU-only timing (HIP-graph replays), random operand bytes.
    python tools/ab_mxsk.py > gpurun_out/ab_mxsk.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us

SHAPES = {"mxf4": [(4096, 4096, 4096), (5120, 4096, 4096), (6144, 4096, 4096), (7168, 4096, 4096), (8192, 4096, 4096), (4096, 5120, 5120), (6144, 4096, 12288), (6144, 4096, 14336), (3072, 8192, 8192), (5120, 8192, 8192),
                   (4096, 14336, 4096), (1536, 28672, 4096), (3072, 28672, 4096), (384, 57344, 8192), (768, 57344, 8192)],
          "mxf8": [(4096, 4096, 4096), (5120, 4096, 4096), (6144, 4096, 4096), (4096, 5120, 5120), (3072, 8192, 28672), (1536, 8192, 28672), (5120, 8192, 8192), (3072, 28672, 4096)]}


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    if os.environ.get("AB_SK_SHAPES"):   # "MxNxK,..." for the formats named on the command line
        for f_ in SHAPES:
            SHAPES[f_] = [tuple(int(d) for d in sh.split("x")) for sh in os.environ["AB_SK_SHAPES"].split(",")]
    print("# operands: %s" % ("all-zero codes under unit scales" if os.environ.get("AB_DATA") == "zero" else "random bytes"))
    for fmt in ([a for a in sys.argv[1:]] or ["mxf4", "mxf8"]):
        epb = 1 if fmt == "mxf8" else 2
        fn = lab.matmul_mxf4_bf16_tn if fmt == "mxf4" else lab.matmul_mxf8_bf16_tn
        print("# %s  %-20s %6s | %9s %9s %9s %9s | %7s %7s | TFLOP/s auto" % (fmt, "M x N x K", "tiles", "90 us", "98 us", "89 us", "auto us", "89/90", "auto/min"), flush=True)
        for (m, n, k) in SHAPES[fmt]:
            a = torch.randint(0, 256, (m, k // epb), dtype=torch.uint8, device=dev, generator=g)
            b = torch.randint(0, 256, (n, k // epb), dtype=torch.uint8, device=dev, generator=g)
            if fmt == "mxf8":
                a &= 0x77; b &= 0x77
            if os.environ.get("AB_DATA") == "zero":   # nothing toggles in the matrix pipe: the clock stays up, the times are the schedules' in cycles (tools/power_data_probe.py)
                a.zero_(); b.zero_()
            sa = torch.randint(118, 126, (pad(m) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(118, 126, (pad(n) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            if os.environ.get("AB_DATA") == "zero":
                sa.fill_(127); sb.fill_(127)
            t = {}
            nrep = max(4, min(40, int(3000 / max(1.0, 2.0 * m * n * k / 3.5e9))))
            for rnd in range(2):
                for v in (90, 98, 89, 0):
                    opts = {"gemm_variant": v}
                    if v == 90: opts["pp_flags"] = 1 | 64
                    with lab.forced(**opts):
                        us = graph_us(lambda: fn(a, b, sa, sb, alpha), n=nrep)
                    t[v] = min(t.get(v, 1e9), us)
            T = ((m + 255) // 256) * ((n + 255) // 256)
            print("  %s  %-20s %6d | %9.2f %9.2f %9.2f %9.2f | %7.3f %7.3f | %6.0f" % (fmt, f"{m}x{n}x{k}", T, t[90], t[98], t[89], t[0], t[89] / t[90], t[0] / min(t[90], t[98], t[89]),
                  2.0 * m * n * k / t[0] / 1e6), flush=True)


main()
