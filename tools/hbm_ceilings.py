"""What the memory system of THIS box gives trivial kernels, cold, at the read : write mixes of the streaming ops -- the practical ceilings the
roofline fractions in DESIGN.md should be read against (the fractions themselves stay priced against 8 TB/s).  torch's own elementwise kernels:
    read only   x.sum() / x.amax()      write only  y.fill_()      1 : 1   y.copy_(x)      2 : 1   torch.add(a, b, out=c)
Buffers are rotated through > 1 GB so neither the 256 MB MALL nor L2 holds them; GPU-only timing (HIP-graph replays).
    python tools/hbm_ceilings.py > gpurun_out/hbm_ceilings.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _timing import graph_us


def main():
    dev = torch.device("cuda:0")
    print("%-46s %10s %10s %10s" % ("kernel (bytes read : written per call)", "MB / call", "us", "TB/s"))
    for mb in (64, 256):
        n = mb * (1 << 20)
        nbuf = max(3, (1536 << 20) // n)
        xs8 = [torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        ys8 = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        st = {"i": 0}
        def rot(fn):
            def f():
                st["i"] = (st["i"] + 1) % nbuf
                return fn(st["i"])
            return f
        cases = [
            ("read only: fp32 view .sum()", n, rot(lambda i: xs8[i].view(torch.float32).sum())),
            ("read only: fp32 view .amax()", n, rot(lambda i: xs8[i].view(torch.float32).amax())),
            ("write only: fill_", n, rot(lambda i: ys8[i].view(torch.int32).fill_(7))),
            ("1 : 1  copy_ (int32 view)", 2 * n, rot(lambda i: ys8[i].view(torch.int32).copy_(xs8[i].view(torch.int32)))),
            ("2 : 1  add(a, b, out=c) int32", 3 * n, rot(lambda i: torch.add(xs8[i].view(torch.int32), xs8[(i + 1) % nbuf].view(torch.int32), out=ys8[i].view(torch.int32)))),
        ]
        for name, nbytes, fn in cases:
            us = min(graph_us(fn, n=2 * nbuf) for _ in range(3))
            print("%-46s %10.1f %10.2f %10.2f" % (f"{name} [{mb} MB buffers]", nbytes / 1e6, us, nbytes / us * 1e-6), flush=True)
        del xs8, ys8


if __name__ == "__main__":
    main()
