"""A/B of two builds of libqutlass_amd.so on the headline GEMM (MXFP4 4096^3, pre-quantised operands), same box, interleaved:
python tools/ab_lib_gemm.py a.so b.so"""
import ctypes, os, sys, torch

def main():
    paths = [p for p in sys.argv[1:] if not p.startswith("-")]
    libs = [ctypes.CDLL(p, mode=ctypes.RTLD_LOCAL) for p in paths]
    dev = torch.device("cuda:0")
    M, N, K = (int(v) for v in os.environ.get("AB_SHAPE", "4096x4096x4096").split("x"))
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randint(0, 256, (M, K // 2), dtype=torch.uint8, device=dev, generator=g)
    b = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev, generator=g)
    sa = torch.randint(124, 131, (M * K // 32,), dtype=torch.uint8, device=dev, generator=g)
    sb = torch.randint(124, 131, (N * K // 32,), dtype=torch.uint8, device=dev, generator=g)
    if os.environ.get("AB_DATA") == "zero":   # all-zero codes under unit scales: the matrix pipe toggles nothing, the clock stays up -- the time is the SCHEDULE's (tools/power_data_probe.py)
        a.zero_(); b.zero_(); sa.fill_(127); sb.fill_(127)
    al = torch.ones(1, device=dev)
    d = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    I = ctypes.c_int64
    def run(lib, n):
        for _ in range(n):
            rc = lib.qutlass_amd_matmul_mxf4_bf16_tn(P(a), P(b), P(sa), P(sb), P(al), P(d), I(M), I(N), I(K), None)
        assert rc == 0
    def t(lib, reps=2000):
        run(lib, 3000); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(lib, reps); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    best = [1e9] * len(libs)
    for rep in range(4):
        for i, lib in enumerate(libs):
            best[i] = min(best[i], t(lib))
            print(f"rep {rep} {paths[i]}: {best[i]:.3f} us (best so far)")
    print(f"best ({os.environ.get('AB_DATA', 'random')} data, {M} x {N} x {K}):", " ".join(f"{x:.3f}" for x in best))

main()
