"""What the scale-byte stores cost in the transposing ops (lab ablation switches, option bwd_variant = 16 * mask): mxfp4_transpose_mxfp8 (mask 1 = no scale stores),
backward_t_bf16 / backward_qt_bf16 wave-owned kernels (mask 16).      python tools/ab_sf_stores.py > gpurun_out/ab_sf_stores.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from _timing import graph_us
from ab_bwd import hadamard


def main():
    dev = torch.device("cuda:0")
    h = hadamard(32, dev)
    alpha = torch.tensor([0.75], device=dev)
    print("%-44s %10s %10s %10s %10s" % ("op (rows x cols)", "warm", "warm -sf", "cold", "cold -sf"))
    for (n, m) in [(8192, 8192), (4096, 4096)]:
        g = torch.Generator(device=dev).manual_seed(1)
        nb = max(2, int(300e6 / (n * m * 0.53)) + 1)
        qs = [torch.randint(0, 256, (n, m // 2), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
        ss = [torch.randint(120, 132, (n, m // 32), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
        nbx = max(2, int(300e6 / (n * m * 2)) + 1)
        xs = [torch.randn(n, m, dtype=torch.bfloat16, device=dev) * 3 for _ in range(nbx)]
        ops = {
            "mxfp4_transpose_mxfp8": ([(lambda a=a, b=b: lab.mxfp4_transpose_mxfp8(a, b, n, m)) for a, b in zip(qs, ss)], 1, {}),
            "backward_qt_bf16 (wave-owned, 4 groups)": ([(lambda a=a, b=b: lab.backward_qt_bf16(a, b, h, alpha)) for a, b in zip(qs, ss)], 16, {"base": 2}),
            "backward_t_bf16 (wave-owned, 8 groups)": ([(lambda x=x: lab.backward_t_bf16(x, h)) for x in xs], 16, {"base": 3}),
        }
        for name, (calls, mask, extra) in ops.items():
            st = {"i": 0}
            def cold():
                st["i"] = (st["i"] + 1) % len(calls)
                return calls[st["i"]]()
            r = []
            for fn, reps in ((calls[0], 20), (cold, 2 * len(calls))):
                for mk in (0, mask):
                    with lab.forced(bwd_variant=extra.get("base", 0) + 16 * mk):
                        r.append(min(graph_us(fn, n=reps) for _ in range(3)))
            print("%-44s %10.2f %10.2f %10.2f %10.2f" % (f"{name} {n}x{m}", *r), flush=True)


if __name__ == "__main__":
    main()
