"""mxfp4_transpose_mxfp8: the wave-owned-lines kernel ([r4], lab option transpose_nc = 4 whole lines / 2 64-byte segments) against the one-shot kernel
(128 = the round-3 product kernel, 256), one box, interleaved, GPU-only timing, warm (one input) and cold (inputs rotated through > 256 MB);
also checks that all kernels return the same bytes.      python tools/ab_transpose.py > gpurun_out/ab_transpose.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us

VARS = [int(v) for v in os.environ.get("AB_TR_VARS", "0,128,256,4,5,6,7,8").split(",")]
NAMES = {5: "[r5] persistent one-shot tile, next tile prefetched, 4 workgroups per CU", 6: "[r5] the same, 3 per CU", 7: "[r5] 2 per CU", 8: "[r5] 6 per CU (4 resident)", 0: "the product rule", 128: "one-shot 128 m x 128 n (round 3)", 3: "one-shot 256 m x 128 n (8 waves)", 256: "one-shot 128 m x 256 n", 4: "wave-owned 128-byte lines", 2: "wave-owned 64-byte segments"}


def main():
    dev = torch.device("cuda:0")
    print("variants (lab option transpose_nc): " + ", ".join(f"{v}={NAMES[v]}" for v in VARS))
    print("%-34s | warm us: %s | cold us: %s |" % ("mxfp4_transpose_mxfp8 (m x n)", " ".join("%7s" % ("v%d" % v) for v in VARS), " ".join("%7s" % ("v%d" % v) for v in VARS)))
    for (m, n) in [(4096, 4096), (8192, 8192), (2048, 14336), (14336, 2048), (1024, 8192), (16384, 8192)]:
        nbuf = max(2, int(300e6 / (m * n * 0.53)) + 1)
        g = torch.Generator(device=dev).manual_seed(1)
        qs = [torch.randint(0, 256, (m, n // 2), dtype=torch.uint8, device=dev, generator=g) for _ in range(nbuf)]
        ss = [torch.randint(118, 134, (m, n // 32), dtype=torch.uint8, device=dev, generator=g) for _ in range(nbuf)]
        calls = [(lambda a=a, b=b: lab.mxfp4_transpose_mxfp8(a, b, m, n)) for a, b in zip(qs, ss)]
        nbytes = m * n // 2 + m * n // 32 + m * n + m * n // 32
        outs, t = {}, {}
        for v in VARS:
            with lab.forced(transpose_nc=v):
                outs[v] = calls[0]()
        same = all(all(torch.equal(a, b) for a, b in zip(outs[VARS[0]], outs[v])) for v in VARS[1:])
        state = {"i": 0}
        def cold():
            state["i"] = (state["i"] + 1) % nbuf
            return calls[state["i"]]()
        for rnd in range(2):
            for v in VARS:
                with lab.forced(transpose_nc=v):
                    t[(v, "w")] = min(t.get((v, "w"), 1e9), graph_us(calls[0], n=20))
                    t[(v, "c")] = min(t.get((v, "c"), 1e9), graph_us(cold, n=2 * nbuf))
        best = min(t[(v, "c")] for v in VARS)
        tb = nbytes / best * 1e-6
        print("%-34s | %s | %s | %6.1f MB -> best cold %5.2f TB/s, %.2f of 8   same=%s" % (f"{m}x{n}", " ".join("%7.2f" % t[(v, "w")] for v in VARS),
              " ".join("%7.2f" % t[(v, "c")] for v in VARS), nbytes / 1e6, tb, tb / 8.0, same), flush=True)
        del calls


if __name__ == "__main__":
    main()
