"""A/B (lab): NVFP4 128x128 tiles with the compiler's order against the fixed one-MFMA-then-its-dequantisation order of the 256-row tiles, where a K split leaves ONE workgroup
per CU (one wave per SIMD: nobody else fills the MFMA shadows).  python tools/ab_nv_fenced.py > gpurun_out/ab_nv_fenced.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import _benchlib as lab

CASES = [(256, 4096, 14336, 4), (128, 4096, 14336, 8), (192, 4096, 14336, 4), (96, 8192, 28672, 8), (256, 5120, 25600, 8), (512, 5120, 25600, 4), (768, 4096, 14336, 4), (1024, 4096, 4096, 1),
         (512, 4096, 4096, 1), (1024, 4096, 14336, 1), (2048, 4096, 4096, 1), (256, 8192, 8192, 2)]


def timed(call, reps):
    for _ in range(max(3, reps // 4)): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    print("# M N K ranges | us compiler order, fixed order, change")
    for (m, n, k, S) in CASES:
        a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
        b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
        sa = torch.randint(0x30, 0x48, (pad(m) * ((k // 16 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
        sb = torch.randint(0x30, 0x48, (pad(n) * ((k // 16 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
        va, vb = (5 if S == 1 else 110 + S), 150 + S
        outs, best = {}, {va: 1e9, vb: 1e9}
        for v in (va, vb):
            with lab.forced(nvf4_variant=v):
                outs[v] = lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha)
        same = torch.equal(outs[va].view(torch.int16), outs[vb].view(torch.int16))
        reps = max(20, min(200, int(8e3 / (2.0 * m * n * k / 1e9 / 1.0))))
        for _ in range(3):
            for v in (va, vb):
                with lab.forced(nvf4_variant=v):
                    best[v] = min(best[v], timed(lambda: lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha), reps))
        print("%5d %6d %6d %d | %8.2f %8.2f %+6.1f %%  %s" % (m, n, k, S, best[va], best[vb], 100 * (best[vb] / best[va] - 1), "equal" if same else "DIFFERENT"), flush=True)


main()
