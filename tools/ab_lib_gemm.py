"""A/B of two builds of libqutlass_amd.so on the headline GEMM (MXFP4 4096^3, pre-quantised operands), same box, interleaved:
python tools/ab_lib_gemm.py a.so b.so"""
import ctypes, sys, torch

def main():
    libs = [ctypes.CDLL(p, mode=ctypes.RTLD_LOCAL) for p in sys.argv[1:3]]
    dev = torch.device("cuda:0")
    M = N = K = 4096
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randint(0, 256, (M, K // 2), dtype=torch.uint8, device=dev, generator=g)
    b = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev, generator=g)
    sa = torch.randint(124, 131, (M * K // 32,), dtype=torch.uint8, device=dev, generator=g)
    sb = torch.randint(124, 131, (N * K // 32,), dtype=torch.uint8, device=dev, generator=g)
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
    best = [1e9, 1e9]
    for rep in range(4):
        for i, lib in enumerate(libs):
            best[i] = min(best[i], t(lib))
            print(f"rep {rep} {sys.argv[1 + i]}: {best[i]:.3f} us (best so far)")
    print("best:", " ".join(f"{x:.3f}" for x in best))

main()
