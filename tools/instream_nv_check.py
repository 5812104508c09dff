"""GPU-only timing (HIP-graph replays of 40 launches, no Python between launches) of matmul_nvf4_bf16_tn's plan against the candidates it was chosen over, for the
shapes whose plan changed in round 3 and whose kernels are short enough for the Python-timed calibration to be in doubt.
    python tools/instream_nv_check.py > gpurun_out/instream_nv_check.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import _benchlib as lab

# (M, N, K, [candidates])   0 = the plan (auto), 3 skinny, 5 / 6 / 7 = 128x128 / 128x64 / 64x64 single pass, 110 + S / 120 + S / 130 + S = those tiles with S K ranges
CASES2W = [(256, 4096, 14336, [0, 114, 164, 168]), (128, 4096, 14336, [0, 118, 168]), (192, 4096, 14336, [0, 114, 168]), (96, 8192, 28672, [0, 118, 168, 164]), (256, 5120, 25600, [0, 118, 168]),
           (512, 5120, 25600, [0, 114, 164, 168]), (768, 4096, 14336, [0, 114, 162, 164]), (1024, 4096, 4096, [0, 5, 161, 162]), (512, 4096, 4096, [0, 6, 161, 162]), (256, 8192, 8192, [0, 112, 162, 164]),
           (2048, 4096, 4096, [0, 5, 161]), (1024, 4096, 14336, [0, 5, 161, 162])]
CASES = [(64, 28672, 4096, [0, 3, 7]), (64, 14336, 4096, [0, 3, 7]), (64, 8192, 28672, [0, 3, 134, 7]), (64, 5120, 25600, [0, 3, 134]), (128, 4096, 14336, [0, 3, 118, 7]),
         (96, 4096, 14336, [0, 3, 118]), (256, 4096, 14336, [0, 7, 114, 112]), (192, 4096, 14336, [0, 7, 114]), (512, 5120, 5120, [0, 5, 6, 114]), (256, 5120, 5120, [0, 6, 7, 114]), (256, 8192, 8192, [0, 6, 112, 114]),
         (96, 8192, 8192, [0, 3, 7, 114]), (128, 8192, 8192, [0, 7, 114]), (64, 4096, 4096, [0, 3, 7, 134]), (128, 4096, 4096, [0, 3, 7]), (256, 4096, 4096, [0, 7, 6, 112]),
         (512, 4096, 4096, [0, 6, 5, 7]), (768, 4096, 14336, [0, 5, 114]), (1024, 5120, 25600, [0, 40, 114])]
NAME = {0: "plan", 3: "skinny", 5: "128x128", 6: "128x64", 7: "64x64", 40: "256x128"}


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    pad = lambda r: (r + 127) // 128 * 128
    for (m, n, k, cands) in (CASES2W if "2w" in sys.argv[1:] else CASES):
        a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
        b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
        sa = torch.randint(0x30, 0x48, (pad(m) * ((k // 16 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
        sb = torch.randint(0x30, 0x48, (pad(n) * ((k // 16 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
        res = []
        for v in cands:
            with lab.forced(nvf4_variant=v):
                call = lambda: lab.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha)
                try:
                    call(); torch.cuda.synchronize()
                    s = torch.cuda.Stream()
                    with torch.cuda.stream(s):
                        for _ in range(3): call()
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        for _ in range(40): call()
                    best = 1e9
                    for _ in range(4): gr.replay()           # warm-up / clock ramp: ~4 x 40 launches
                    torch.cuda.synchronize()
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(5): gr.replay()
                        e1.record(); torch.cuda.synchronize()
                        best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
                    res.append((v, best))
                    del gr
                except Exception as e:
                    res.append((v, float("nan")))
        nm = lambda v: NAME[v] if v in NAME else "%s/%d" % ({11: "128x128", 12: "128x64", 13: "64x64", 16: "2wave128x128"}[v // 10], v % 10)
        plan = res[0][1]
        print("%5d %6d %6d | " % (m, n, k) + "  ".join("%s %.2f" % (nm(v), t) for v, t in res) + "   | best other / plan = %.3f" % (min(t for v, t in res[1:]) / plan), flush=True)


main()
