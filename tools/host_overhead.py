"""Host-side launch cost of one op through the Python surface (run on the GPU box): python tools/host_overhead.py"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, qutlass_amd as q
from qutlass_amd.utils import to_blocked
dev = torch.device("cuda", 0)
x = torch.randn(4096, 4096, dtype=torch.bfloat16, device=dev)
h = torch.eye(32, dtype=torch.bfloat16, device=dev)
for _ in range(10): q.fusedQuantizeMx(x, h, method="abs_max")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): q.fusedQuantizeMx(x, h, method="abs_max")
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host us/call (launch loop, no sync):", (t1 - t0) / 2000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): q.fusedQuantizeMx(x, h, method="abs_max")
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
