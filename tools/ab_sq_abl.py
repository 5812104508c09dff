"""backward_bf16_square_double_mxfp8: what its partial-line scale stores cost -- the lab library's ablation switches (option bwd_variant = 16 * mask; 1 = no row-scale
stores, 2 = no column-scale stores, 4 = no data stores), GPU-only timing, warm and cold.      python tools/ab_sq_abl.py > gpurun_out/ab_sq_abl.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us

MASKS = [0, 1, 2, 3, 4, 7]
NAMES = {0: "all", 1: "-row sf", 2: "-col sf", 3: "-both sf", 4: "-data st", 7: "loads only"}


def main():
    dev = torch.device("cuda:0")
    print("%-44s %s" % ("backward_bf16_square_double_mxfp8 (m x n)", " ".join("%11s" % NAMES[m] for m in MASKS)))
    for (m, n) in [(8192, 8192), (4096, 4096), (16384, 8192), (2048, 14336), (14336, 2048), (1024, 4096)]:
        nbuf = max(2, int(300e6 / (m * n * 2)) + 1)
        xs = [torch.randn(m, n, dtype=torch.bfloat16, device=dev) * 3 for _ in range(nbuf)]
        calls = [(lambda x=x: lab.backward_bf16_square_double_mxfp8(x)) for x in xs]
        st = {"i": 0}
        def cold():
            st["i"] = (st["i"] + 1) % nbuf
            return calls[st["i"]]()
        outs = {}
        for v in (1, 4, 8, 0):
            with lab.forced(transpose_nc=v):
                outs[v] = calls[0]()
        same = all(all(torch.equal(a, b) for a, b in zip(outs[1], outs[v])) for v in (4, 8, 0))
        for mode, fn, reps in (("warm", calls[0], 20), ("cold", cold, 2 * nbuf)):
            row = []
            for mask in MASKS:
                with lab.forced(bwd_variant=16 * mask, transpose_nc=1):
                    row.append(min(graph_us(fn, n=reps) for _ in range(2)))
            kern = []
            for v in (1, 4, 8, 0):      # 4 waves x one column tile (round 3), 16 waves x 1 / 2 column tiles (512 / 1024 columns), the product rule
                with lab.forced(transpose_nc=v):
                    kern.append(min(graph_us(fn, n=reps) for _ in range(3)))
            print("%-44s %s   | 128 cols %7.2f  512 %7.2f  1024 %7.2f  rule %7.2f  same=%s" % (f"{m}x{n} {mode}", " ".join("%11.2f" % v for v in row), *kern, same), flush=True)
        del calls, xs


if __name__ == "__main__":
    main()
