"""Scan the GEMM ops for wave-quantisation / tile-rule blind spots: for every (N, K) of the reference's benchmark models and a fine grid of batch
sizes M, time the GEMM alone (C ABI entries with caller scratch, random operand bytes; GPU-only timing through HIP-graph replays) and flag every place where a LARGER batch runs FASTER, or the TFLOP/s fall by more
than 8 % ([r4]; 12 % in round 3) from one M to the next.      python tools/dip_scan.py [mxf4|nvf4|mxf8 ...] > gpurun_out/dip_scan.txt"""
import ctypes, os, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "qutlass_amd", "libqutlass_amd.so"), mode=ctypes.RTLD_LOCAL)
NK = [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (8192, 8192), (57344, 8192), (8192, 28672), (5120, 5120), (51200, 5120), (5120, 25600), (24576, 4096), (4096, 12288)]
MS = [64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192]
P = lambda t: ctypes.c_void_p(t.data_ptr()); I = ctypes.c_int64
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)   # the stream of the moment: the capturing one inside graph_us
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _timing import graph_us


def main():
    fmts = sys.argv[1:] or ["mxf4", "nvf4", "mxf8"]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    alpha = torch.ones(1, device=dev)
    flagged = 0
    for fmt in fmts:
        epb = 1 if fmt == "mxf8" else 2          # elements per byte
        gs = 16 if fmt == "nvf4" else 32
        # the entries with caller scratch (what the torch ops call): split-K where the plan wants it
        fn = {"mxf4": lib.qutlass_amd_matmul_mxf4_bf16_tn_ws, "nvf4": lib.qutlass_amd_matmul_nvf4_bf16_tn_ws, "mxf8": lib.qutlass_amd_matmul_mxf8_bf16_tn_ws}[fmt]
        lib.qutlass_amd_gemm_splitk_workspace_bytes.restype = lib.qutlass_amd_nvf4_splitk_workspace_bytes.restype = ctypes.c_int64
        ws_need = (lambda m, n, k: lib.qutlass_amd_nvf4_splitk_workspace_bytes(I(m), I(n), I(k))) if fmt == "nvf4" else \
                  (lambda m, n, k: lib.qutlass_amd_gemm_splitk_workspace_bytes(8 if fmt == "mxf8" else 4, I(m), I(n), I(k)))
        for (n, k) in NK:
            pad = lambda r: (r + 127) // 128 * 128
            b = torch.randint(0, 256, (n, k // epb), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(118, 126, (pad(n) * ((k // gs + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            prev = None
            row = []
            for m in MS:
                a = torch.randint(0, 256, (m, k // epb), dtype=torch.uint8, device=dev, generator=g)
                if fmt == "mxf8":
                    a &= 0x77; 
                sa = torch.randint(118, 126, (pad(m) * ((k // gs + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
                d = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
                wsb = ws_need(m, n, k)
                ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
                call = lambda: fn(P(a), P(b), P(sa), P(sb), P(alpha), P(d), I(m), I(n), I(k), P(ws) if wsb else ctypes.c_void_p(0), I(wsb), ST())
                assert call() == 0, (fmt, m, n, k)
                fl = 2.0 * m * n * k
                best = graph_us(call, n=max(4, min(40, int(2.5e3 / max(fl / 3.0e15 * 1e6, 5.0)))))   # GPU-only: HIP-graph replays (tools/_timing.py)
                tf = fl / best * 1e-6
                flag = ""
                if prev is not None:
                    if best < prev[0] * 0.98: flag = "  <-- FASTER than the smaller batch (%.1f us at M = %d)" % (prev[0], prev[2])
                    elif tf < prev[1] * 0.92: flag = "  <-- TFLOP/s fall %.0f %%" % (100 * (1 - tf / prev[1]))
                if flag: flagged += 1
                print("%s N=%-6d K=%-6d M=%-5d %9.2f us %8.1f TFLOP/s%s" % (fmt, n, k, m, best, tf, flag), flush=True)
                prev = (best, tf, m)
            del b, sb
    print("flagged:", flagged)


main()
