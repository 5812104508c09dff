"""How long the part takes to reach its steady clock under the headline GEMM (run on the GPU box):
per-chunk average step time of a long back-to-back launch sequence, plus rocm-smi power / sclk samples."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, qutlass_amd as q
from qutlass_amd.utils import to_blocked
dev = torch.device("cuda", 0)
torch.manual_seed(0)
M = N = K = 4096
h = torch.ones(1, 1)
while h.shape[0] < 32: h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
h = (h * 32 ** -0.5).to(torch.bfloat16).to(dev)
a = torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0
b = torch.randn(N, K, dtype=torch.bfloat16, device=dev) * 25.0
a_q, a_s = q.fusedQuantizeMx(a, h, method="abs_max"); b_q, b_s = q.fusedQuantizeMx(b, h, method="abs_max")
a_sf, b_sf = to_blocked(a_s), to_blocked(b_s); alpha = torch.tensor([1.0], device=dev)
samples, stop = [], False
def smi():
    while not stop:
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout)["card0"]
            samples.append((time.perf_counter(), d.get("Current Socket Graphics Package Power (W)"), d.get("sclk clock speed:")))
        except Exception:
            pass
        time.sleep(0.05)
th = threading.Thread(target=smi); th.start()
torch.cuda.synchronize(); time.sleep(1.0)          # idle first: the part drops to its low-power state
t0 = time.perf_counter()
chunks = [25, 200, 200, 500, 1000, 2000, 4000, 8000, 16000, 16000]
for n in chunks:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): q.matmul_mxf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha)
    e1.record(); e1.synchronize()
    print(f"t={time.perf_counter() - t0:7.3f}s  chunk of {n:6d} steps: {e0.elapsed_time(e1) * 1e3 / n:7.2f} us/step  {2.0 * M * N * K / (e0.elapsed_time(e1) * 1e-3 / n) / 1e12:7.1f} TFLOP/s", flush=True)
stop = True; th.join()
for t, p, c in samples: print(f"smi t={t - t0:7.3f}s power={p} W sclk={c}")
