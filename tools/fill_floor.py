"""How long does it take merely to WRITE the output of the headline GEMM?  torch fill / copy kernels on a 4096 x 4096 bf16 tensor (33.5 MB), the same buffer every time (as
bench.py's D), GPU-only timing (HIP-graph replays): the floor under the store side of the GEMM's last stage.      python tools/fill_floor.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from _timing import graph_us

dev = torch.device("cuda:0")
for (m, n) in [(4096, 4096), (4096, 14336), (8192, 8192)]:
    d = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    s = torch.randn(m, n, dtype=torch.bfloat16, device=dev)
    f = min(graph_us(lambda: d.fill_(1.5), n=20) for _ in range(3))
    z = min(graph_us(lambda: d.zero_(), n=20) for _ in range(3))
    c = min(graph_us(lambda: d.copy_(s), n=20) for _ in range(3))
    mb = m * n * 2 / 1e6
    print(f"{m} x {n} bf16 = {mb:.1f} MB: fill_ {f:.2f} us ({mb / f / 1e6 * 1e6:.2f} TB/s)  zero_ {z:.2f} us  copy_ (read + write) {c:.2f} us")
