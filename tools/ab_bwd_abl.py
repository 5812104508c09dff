"""Where the time of bwd_quant_tw_kernel goes: the lab library's ablation switches (option bwd_variant = kernel + 16 * mask; mask 1 = no global loads,
2 = no unit stores, 4 = no MFMA + quantisation, 8 = no staging) on backward_qt_bf16 / backward_t_bf16, GPU-only timing, warm and cold.
    python tools/ab_bwd_abl.py > gpurun_out/ab_bwd_abl.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us
from ab_bwd import hadamard

MASKS = [0, 2, 4, 12, 14, 13, 15]
NAMES = {0: "all", 2: "-stores", 4: "-compute", 12: "loads+stores", 14: "loads only", 13: "stores only", 15: "walk only"}


def main():
    dev = torch.device("cuda:0")
    h = hadamard(32, dev)
    alpha = torch.tensor([0.75], device=dev)
    print("%-34s %s" % ("op (N x M), kernel", " ".join("%16s" % NAMES[m] for m in MASKS)))
    for (n, m) in [(8192, 8192)]:
        for op in ("qt", "t"):
            nbuf = max(2, int(300e6 / (n * m * (2 if op == "t" else 0.53))) + 1)
            if op == "t":
                xs = [torch.randn(n, m, dtype=torch.bfloat16, device=dev) * 3 for _ in range(nbuf)]
                calls = [(lambda x=x: lab.backward_t_bf16(x, h)) for x in xs]
            else:
                g = torch.Generator(device=dev).manual_seed(1)
                qs = [torch.randint(0, 256, (n, m // 2), dtype=torch.uint8, device=dev, generator=g) for _ in range(nbuf)]
                ss = [torch.randint(120, 132, (n, m // 32), dtype=torch.uint8, device=dev, generator=g) for _ in range(nbuf)]
                calls = [(lambda a=a, b=b: lab.backward_qt_bf16(a, b, h, alpha)) for a, b in zip(qs, ss)]
            state = {"i": 0}
            def cold():
                state["i"] = (state["i"] + 1) % nbuf
                return calls[state["i"]]()
            for kern in (2, 3):
                for mode, fn, reps in (("warm", calls[0], 20), ("cold", cold, 2 * nbuf)):
                    row = []
                    for mask in MASKS:
                        with lab.forced(bwd_variant=kern + 16 * mask):
                            row.append(min(graph_us(fn, n=reps) for _ in range(2)))
                    print("%-34s %s" % (f"backward_{op} {n}x{m} v{kern} {mode}", " ".join("%16.2f" % v for v in row)), flush=True)
            del calls


main()
