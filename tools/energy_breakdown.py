"""Energy per launch of the headline GEMM and of its ablated variants (run on the GPU box): each variant runs back to
back for ~1.2 s while rocm-smi is sampled; energy = mean socket power over the last second x time per launch.
Ablation variants (bench-only instantiations of the deep schedule): 31 no epilogue, 32 no DMA + no epilogue,
33 no MFMA + no epilogue, 34 no fragment reads + no epilogue."""
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
            d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True).stdout)["card0"]
            samples.append((time.perf_counter(), float(d.get("Current Socket Graphics Package Power (W)"))))
        except Exception:
            pass
        time.sleep(0.04)
th = threading.Thread(target=smi); th.start()
time.sleep(1.0)
idle = sum(p for _, p in samples) / max(1, len(samples))
print(f"idle power {idle:.0f} W")
names = {30: "deep (default)", 31: "deep, no epilogue", 32: "deep, no DMA, no epilogue", 33: "deep, no MFMA, no epilogue", 34: "deep, no fragment reads, no epilogue",
         20: "simple (8 waves)", 5: "lockstep (8 waves)", 24: "128x128 tiles"}
for var in (30, 31, 32, 33, 34, 20, 5, 24, 30):
    q._lib.set_option("gemm_variant", var)
    n_total, t_start = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.perf_counter() - t_start < 0.3:                     # ramp
        for _ in range(500): q.matmul_mxf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha)
        torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record(); n = 0
    while time.perf_counter() - t0 < 1.0:
        for _ in range(2000): q.matmul_mxf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha)
        n += 2000
        torch.cuda.current_stream().synchronize()
    e1.record(); e1.synchronize(); t1 = time.perf_counter()
    us = e0.elapsed_time(e1) * 1e3 / n
    pw = [p for t, p in samples if t0 + 0.15 <= t <= t1]
    P = sum(pw) / max(1, len(pw))
    print(f"variant {var:3d} {names[var]:40s}: {us:7.2f} us/launch  power {P:6.0f} W (n={len(pw)})  energy {P * us * 1e-3:6.1f} mJ  above idle {(P - idle) * us * 1e-3:6.1f} mJ", flush=True)
    q._lib.set_option("gemm_variant", 0)
stop = True; th.join()
