"""Lab variants of matmul_mxf4_bf16_tn side by side (option gemm_variant of the LAB library), GPU-only timing (HIP-graph replays), on random code bytes and on
all-zero codes under unit scales (no data-dependent power: the schedule in cycles -- tools/power_data_probe.py).  Default: the output-store cache policies of the
persistent kernel (90 = product: sc0 sc1; 92 nt; 93 sc1; 94 none = write-back; 95 sc0 sc1 nt; 96 sc0).
    AB_VARIANTS=90,94 AB_SHAPES=4096x4096x4096 python tools/ab_gemm_variants.py > gpurun_out/ab_gemm_variants.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us


def main():
    dev = torch.device("cuda:0")
    variants = [int(v) for v in os.environ.get("AB_VARIANTS", "90,92,93,94,95,96").split(",")]
    shapes = [tuple(int(d) for d in sh.split("x")) for sh in os.environ.get("AB_SHAPES", "4096x4096x4096,4096x14336x4096,8192x8192x8192").split(",")]
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    print("# us per launch, best of 3 rounds; columns = gemm_variant " + " ".join(str(v) for v in variants))
    for (m, n, k) in shapes:
        for data in ("zero", "random"):
            a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
            b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
            sa = torch.randint(118, 126, (pad(m) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(118, 126, (pad(n) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            if data == "zero":
                a.zero_(); b.zero_(); sa.fill_(127); sb.fill_(127)
            t = {}
            nrep = max(4, min(40, int(3000 / max(1.0, 2.0 * m * n * k / 3.5e9))))
            for rnd in range(3):
                for v in variants:
                    with lab.forced(gemm_variant=v, pp_flags=1 | 64):
                        t[v] = min(t.get(v, 1e9), graph_us(lambda: lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha), n=nrep))
            print("%-22s %-6s | %s" % (f"{m}x{n}x{k}", data, " ".join("%8.2f" % t[v] for v in variants)), flush=True)


main()
