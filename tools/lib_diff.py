"""WHERE do two builds of libqutlass_amd.so disagree?  Runs one MXFP4 / MXFP8 shape through both and, if the outputs differ, prints the mismatches broken down by the
coordinates of the persistent kernels' retirement (gemm_mx_deepp.hip.h): wave quadrant of the 256 x 256 tile, accumulator pair (m, h), read-back pass, row within the
pass (lane / 8), 16-byte column group (lane % 8) -- a pattern localises a scheduling bug (one pair? one pass? the first pair after the hand-off?).

    python tools/lib_diff.py old.so new.so [--shapes=4096x4096x512,4096x4096x4096] [--fmt=mxf4] [--alpha=1.0,0.37]

Written for the round-4 `QAMD_DEEPP_RB2=1` variant (1 % slower AND different bytes, profiles/ab_lib_rb2_r4bj.txt) and kept as the first thing to run on any new retirement."""
import ctypes, os, sys
from collections import Counter
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    paths = [a for a in sys.argv[1:] if not a.startswith("--")][:2]
    libs = [ctypes.CDLL(p, mode=ctypes.RTLD_LOCAL) for p in paths]
    opt = lambda k, d: next((a[len(k) + 3:] for a in sys.argv if a.startswith("--" + k + "=")), d)
    shapes = [tuple(int(v) for v in x.split("x")) for x in opt("shapes", "4096x4096x512,4096x4096x4096,4100x4360x768").split(",")]
    fmt = opt("fmt", "mxf4")
    alphas = [float(x) for x in opt("alpha", "1.0,0.37").split(",")]
    entry, fp8 = {"mxf4": ("qutlass_amd_matmul_mxf4_bf16_tn", False), "mxf8": ("qutlass_amd_matmul_mxf8_bf16_tn", True)}[fmt]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    I = ctypes.c_int64
    for (m, n, k) in shapes:
        for alpha in alphas:
            if fp8:
                a = (torch.randn(m, k, device=dev, generator=g) * 2).to(torch.float8_e4m3fn).view(torch.uint8)
                b = (torch.randn(n, k, device=dev, generator=g) * 2).to(torch.float8_e4m3fn).view(torch.uint8)
            else:
                a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
                b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
            pad = lambda r: (r + 127) // 128 * 128
            sa = torch.randint(124, 131, (pad(m) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            sb = torch.randint(124, 131, (pad(n) * ((k // 32 + 3) // 4 * 4),), dtype=torch.uint8, device=dev, generator=g)
            al = torch.full((1,), alpha, device=dev)
            ds = [torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=dev) for _ in libs]
            for rep in range(3):   # (three runs each: is the disagreement reproducible?)
                for i, lib in enumerate(libs):
                    rc = getattr(lib, entry)(P(a), P(b), P(sa), P(sb), P(al), P(ds[i]), I(m), I(n), I(k), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                    assert rc == 0
                torch.cuda.synchronize()
                bad = (ds[0].view(torch.int16) != ds[1].view(torch.int16)).nonzero()
                print(f"{fmt} {m}x{n}x{k} alpha {alpha} run {rep}: {bad.shape[0]} of {m * n} outputs differ", flush=True)
                if bad.shape[0] == 0:
                    break
                if rep == 0:
                    r, c = bad[:, 0].cpu(), bad[:, 1].cpu()
                    rt, ct = r % 256, c % 256
                    wave = ((rt // 128) * 2 + ct // 128)
                    rw, cw = rt % 128, ct % 128
                    cnt = lambda name, v: print(f"    by {name:28s}", dict(sorted(Counter(v.tolist()).items())))
                    cnt("tile (row-major id)", (r // 256) * ((n + 255) // 256) + c // 256) if bad.shape[0] < 2000 else None
                    cnt("wave (2 wave_m + wave_n)", wave)
                    cnt("pair m = row / 32", rw // 32)
                    cnt("pair h = col / 64", cw // 64)
                    cnt("pass = (row % 32) / 8", (rw % 32) // 8)
                    cnt("row in pass = lane / 8", rw % 8)
                    cnt("16-byte group = lane % 8", (cw % 64) // 8)
                    cnt("element in the 16 bytes", cw % 8)
                    for j in range(min(8, bad.shape[0])):
                        rr, cc = int(r[j]), int(c[j])
                        print(f"    ({rr}, {cc}): old {float(ds[0][rr, cc]):.6g}  new {float(ds[1][rr, cc]):.6g}")


if __name__ == "__main__":
    main()
