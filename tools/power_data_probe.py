"""Is the headline GEMM bound by its schedule or by the socket's power budget?  The SAME kernel on the SAME shape with operands of different bit activity:
the bench operands (randn * 25 through fusedQuantizeMx), uniformly random code bytes, all-zero codes, all codes 0x11 (every nibble +0.5), and the bench operands
with every scale byte 127 -- GPU-only time (HIP-graph replays), then a >= 300 ms steady window with socket power and shader clock from librocm_smi64.
The instruction stream is identical in all cases; only the data the matrix pipe toggles differs.      python tools/power_data_probe.py > gpurun_out/power_data_probe.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import qutlass_amd as q
from qutlass_amd.utils import to_blocked
from _timing import graph_us
from bench import PowerSampler

DEV = torch.device("cuda:0")


def hadamard(n):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(DEV)


def window(fn, sampler, tag, us):
    n = max(50, int(400e3 / us))
    torch.cuda.synchronize()
    sampler.mark(tag + "_a")
    for _ in range(n): fn()
    torch.cuda.synchronize()
    sampler.mark(tag + "_b")
    return sampler.window(tag + "_a", tag + "_b")


def main():
    sampler = PowerSampler()
    alpha = torch.ones(1, device=DEV)
    for fmt, (m, n, k) in (("mxfp4", (4096, 4096, 4096)), ("mxfp4", (4096, 14336, 4096)), ("mxfp8", (4096, 4096, 4096)), ("nvfp4", (8192, 8192, 8192))):
        torch.manual_seed(0)
        g = torch.Generator(device=DEV).manual_seed(1)
        cases = []
        if fmt == "mxfp4":
            h = hadamard(32)
            xa = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25; xb = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25
            a, sa = q.fusedQuantizeMx(xa, h, method="abs_max"); b, sb = q.fusedQuantizeMx(xb, h, method="abs_max")
            sa_b, sb_b = to_blocked(sa), to_blocked(sb)
            one = torch.full_like(sa_b.view(torch.uint8), 127).view(torch.float8_e8m0fnu)
            one_b = torch.full_like(sb_b.view(torch.uint8), 127).view(torch.float8_e8m0fnu)
            rnd = lambda t: torch.randint(0, 256, t.shape, dtype=torch.uint8, device=DEV, generator=g)
            call = lambda A, B, SA, SB: (lambda: q.matmul_mxf4_bf16_tn(A, B, SA, SB, alpha))
            cases = [("bench operands (quantised randn)", call(a, b, sa_b, sb_b)), ("bench codes, every scale 2^0", call(a, b, one, one_b)),
                     ("uniformly random code bytes, scales 2^0", call(rnd(a), rnd(b), one, one_b)), ("all codes 0x11 (+0.5), scales 2^0", call(torch.full_like(a, 0x11), torch.full_like(b, 0x11), one, one_b)),
                     ("all-zero codes, scales 2^0", call(torch.zeros_like(a), torch.zeros_like(b), one, one_b))]
        elif fmt == "mxfp8":
            a = (torch.randn(m, k, device=DEV) * 4).to(torch.float8_e4m3fn); b = (torch.randn(n, k, device=DEV) * 4).to(torch.float8_e4m3fn)
            sa_b = to_blocked(torch.randint(120, 131, (m, k // 32), dtype=torch.uint8, device=DEV, generator=g).view(torch.float8_e8m0fnu))
            sb_b = to_blocked(torch.randint(120, 131, (n, k // 32), dtype=torch.uint8, device=DEV, generator=g).view(torch.float8_e8m0fnu))
            call = lambda A, B: (lambda: q.matmul_mxf8_bf16_tn(A, B, sa_b, sb_b, alpha))
            z = lambda t: torch.zeros_like(t.view(torch.uint8)).view(torch.float8_e4m3fn)
            cases = [("bench operands (randn * 4 as e4m3)", call(a, b)), ("all-zero operands", call(z(a), z(b)))]
        else:
            xa = torch.randn(m, k, dtype=torch.bfloat16, device=DEV) * 25; xb = torch.randn(n, k, dtype=torch.bfloat16, device=DEV) * 25
            h = hadamard(16)
            gs = torch.tensor([1.0], device=DEV)
            a, sa = q.fusedQuantizeNv(xa, h, gs); b, sb = q.fusedQuantizeNv(xb, h, gs)
            sa_b, sb_b = to_blocked(sa), to_blocked(sb)
            call = lambda A, B: (lambda: q.matmul_nvf4_bf16_tn(A, B, sa_b, sb_b, alpha))
            cases = [("bench operands (quantised randn)", call(a, b)), ("all-zero codes", call(torch.zeros_like(a), torch.zeros_like(b)))]
        flops = 2.0 * m * n * k
        print(f"\n## {fmt} {m} x {n} x {k}", flush=True)
        for rnd_ in range(2):
            for name, fn in cases:
                us = min(graph_us(fn, n=20) for _ in range(2))
                w = window(fn, sampler, f"{fmt}{m}{n}{name}{rnd_}", us)
                print(f"  {name:44s} {us:8.2f} us  {flops / us * 1e-6:7.0f} TFLOP/s   power {w.get('power_w', float('nan')):7.1f} W  sclk {w.get('sclk_mhz', float('nan')):7.1f} MHz  ({w.get('samples', 0)} samples)", flush=True)
    sampler.stop()


if __name__ == "__main__":
    main()
