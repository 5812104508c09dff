"""backward_t_bf16 / backward_qt_bf16: the wave-owned-lines kernel (bwd_quant_tw_kernel, [r4]) against the round-3 kernel (lab option bwd_variant = 1), one box,
interleaved, GPU-only timing (HIP-graph replays), warm (one input) and cold (inputs rotated through > 256 MB so the MALL does not hold them); also checks that
both kernels return the same bytes.      python tools/ab_bwd.py > gpurun_out/ab_bwd.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us


def hadamard(n, dev):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(dev)


VARS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
NAMES = {9: "[r6] wave-owned 32-byte segments (units of 2 groups, up to 20 waves per CU)", 5: "[r5] QT: v2 with the input through a shared whole-line ring (NG 4, 3 slots)", 6: "[r5] QT: ring, NG 8, 4 slots, 16 quads per XCD", 7: "[r5] QT: ring, NG 8, 4 slots", 8: "[r5] QT: ring, NG 4, 3 slots, 16 quads per XCD", 0: "the product rule", 1: "round-3 kernel (8 waves per unit, 2 barriers per unit)", 2: "wave-owned 64-byte segments (units of 4 groups, 12-16 waves per CU)", 4: "[r5] QT only: whole-line panels, [256 n][256 m] per workgroup (T: the product rule)", 3: "wave-owned 128-byte lines (units of 8 groups, 8 waves per CU)"}


def main():
    global VARS
    if os.environ.get("AB_BWD_VARS"):
        VARS = [int(v) for v in os.environ["AB_BWD_VARS"].split(",")]
    shapes = [(4096, 4096), (8192, 8192), (2048, 14336), (8192, 1024)]
    if os.environ.get("AB_BWD_SHAPES"):
        shapes = [tuple(int(d) for d in sh.split("x")) for sh in os.environ["AB_BWD_SHAPES"].split(",")]
    ops = os.environ.get("AB_BWD_OPS", "t,qt").split(",")
    dev = torch.device("cuda:0")
    h = hadamard(32, dev)
    alpha = torch.tensor([0.75], device=dev)
    print("variants (lab option bwd_variant): " + ", ".join(f"{v}={NAMES[v]}" for v in VARS))
    print("%-30s | warm us: %s | cold us: %s |" % ("op  (N x M)", " ".join("%7s" % ("v%d" % v) for v in VARS), " ".join("%7s" % ("v%d" % v) for v in VARS)))
    for (n, m) in shapes:
        for op in ops:
            nbuf = max(2, int(300e6 / (n * m * (2 if op == "t" else 0.53))) + 1)
            if op == "t":
                xs = [torch.randn(n, m, dtype=torch.bfloat16, device=dev) * 3 for _ in range(nbuf)]
                calls = [(lambda x=x: lab.backward_t_bf16(x, h)) for x in xs]
                nbytes = n * m * 2 + n * m // 2 + n * m // 32
            else:
                g = torch.Generator(device=dev).manual_seed(1)
                qs = [torch.randint(0, 256, (n, m // 2), dtype=torch.uint8, device=dev, generator=g) for _ in range(nbuf)]
                ss = [torch.randint(120, 132, (n, m // 32), dtype=torch.uint8, device=dev, generator=g) for _ in range(nbuf)]
                calls = [(lambda a=a, b=b: lab.backward_qt_bf16(a, b, h, alpha)) for a, b in zip(qs, ss)]
                nbytes = 2 * (n * m // 2 + n * m // 32)
            outs = {}
            t = {}
            for v in VARS:
                with lab.forced(bwd_variant=v):
                    outs[v] = calls[0]()
            same = all(all(torch.equal(a, b) for a, b in zip(outs[VARS[0]], outs[v])) for v in VARS[1:])
            state = {"i": 0}
            def cold():
                state["i"] = (state["i"] + 1) % nbuf
                return calls[state["i"]]()
            for rnd in range(2):
                for v in VARS:
                    with lab.forced(bwd_variant=v):
                        t[(v, "w")] = min(t.get((v, "w"), 1e9), graph_us(calls[0], n=20))
                        t[(v, "c")] = min(t.get((v, "c"), 1e9), graph_us(cold, n=2 * nbuf))
            best = min(t[(v, "c")] for v in VARS)
            tb = nbytes / best * 1e-6
            print("%-30s | %s | %s | %6.1f MB -> best cold %5.2f TB/s, %.2f of 8   same=%s" % (f"backward_{op}_bf16 {n}x{m}", " ".join("%7.2f" % t[(v, "w")] for v in VARS),
                  " ".join("%7.2f" % t[(v, "c")] for v in VARS), nbytes / 1e6, tb, tb / 8.0, same), flush=True)
            del calls


if __name__ == "__main__":
    main()
