"""A/B of two builds of libqutlass_amd.so on small / mid-size GEMM shapes of all three formats (same box, interleaved, GPU-only timing through HIP-graph
replays of the C-ABI call):      python tools/ab_lib_shapes.py old.so new.so [--big | --shapes=MxNxK,...] [--fmt=mxf4 ...] > gpurun_out/ab_lib_shapes.txt"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _timing import graph_us

SHAPES = [(1, 4096, 4096), (8, 4096, 4096), (16, 4096, 14336), (32, 14336, 4096), (64, 4096, 4096), (128, 6144, 4096), (256, 4096, 4096), (512, 4096, 4096),
          (1024, 4096, 4096), (1024, 6144, 4096), (2048, 4096, 4096), (512, 28672, 4096)]


BIG = [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 28672, 4096), (8192, 4096, 14336), (2048, 28672, 4096), (5120, 4096, 4096), (3072, 6144, 4096), (4096, 11008, 4096), (4096, 14336, 4096)]


def main():
    libs = [ctypes.CDLL(p, mode=ctypes.RTLD_LOCAL) for p in [a for a in sys.argv[1:] if not a.startswith("--")][:2]]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    I = ctypes.c_int64
    print("%-6s %-22s %10s %10s %8s   same bytes" % ("fmt", "M x N x K", "old us", "new us", "new/old"))
    fmts = [a[6:] for a in sys.argv if a.startswith("--fmt=")]
    for fmt, entry, gs, fp8 in (("mxf4", "qutlass_amd_matmul_mxf4_bf16_tn", 32, False), ("mxf8", "qutlass_amd_matmul_mxf8_bf16_tn", 32, True), ("nvf4", "qutlass_amd_matmul_nvf4_bf16_tn", 16, False)):
        if fmts and fmt not in fmts: continue
        custom = [tuple(int(v) for v in x.split("x")) for a in sys.argv if a.startswith("--shapes=") for x in a[9:].split(",")]   # --shapes=4096x4096x4096,4100x4360x768
        for (m, n, k) in (custom or (BIG if "--big" in sys.argv else SHAPES)):
            kb = k if fp8 else k // 2
            if fp8:
                a = (torch.randn(m, k, device=dev, generator=g) * 2).to(torch.float8_e4m3fn).view(torch.uint8)
                b = (torch.randn(n, k, device=dev, generator=g) * 2).to(torch.float8_e4m3fn).view(torch.uint8)
            else:
                a = torch.randint(0, 256, (m, kb), dtype=torch.uint8, device=dev, generator=g)
                b = torch.randint(0, 256, (n, kb), dtype=torch.uint8, device=dev, generator=g)
            lo, hi = (124, 131) if gs == 32 else (0x30, 0x40)
            pad = lambda r: (r + 127) // 128 * 128
            sa = torch.randint(lo, hi, (pad(m) * ((k // gs + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(lo, hi, (pad(n) * ((k // gs + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            al = torch.ones(1, device=dev)
            ds = [torch.empty(m, n, dtype=torch.bfloat16, device=dev) for _ in libs]
            st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            def call(i):
                rc = getattr(libs[i], entry)(P(a), P(b), P(sa), P(sb), P(al), P(ds[i]), I(m), I(n), I(k), st())
                assert rc == 0
            t = [1e9, 1e9]
            for rep in range(3):
                for i in range(2):
                    t[i] = min(t[i], graph_us(lambda i=i: call(i), n=40))
            print("%-6s %-22s %10.2f %10.2f %8.3f   %s" % (fmt, f"{m}x{n}x{k}", t[0], t[1], t[1] / t[0], bool(torch.equal(ds[0], ds[1]))), flush=True)


if __name__ == "__main__":
    main()
