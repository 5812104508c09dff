"""A/B of two builds of libqutlass_amd.so on the streaming ops (same box, interleaved): python tools/ab_stream_ops.py old.so new.so [n]
(n x n inputs, default 4096; at n = 8192 the cold rotation uses 12 inputs = 1.5 GiB)
Times the quantizers and the QAT-backward ops through the C ABI (torch only allocates), warm (same 32 MiB input) and cold
(40 distinct inputs rotated, 1.3 GiB > the 256 MiB MALL)."""
import ctypes, sys, torch

def load(path):
    return ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)

def main():
    old, new = load(sys.argv[1]), load(sys.argv[2])
    dev = torch.device("cuda:0")
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    NCOLD = 40 if n <= 4096 else 12
    xs = [torch.randn(n, n, device=dev, dtype=torch.bfloat16) * 25 for _ in range(NCOLD)]
    hs = {R: (torch.randn(R, R, device=dev) * 0.2).to(torch.bfloat16) for R in (16, 32, 64, 128)}
    gs = torch.full((1,), 0.37, device=dev)
    out = torch.empty(n * n // 2, device=dev, dtype=torch.uint8)
    sf = torch.empty(n * n // 16, device=dev, dtype=torch.uint8)
    mask = torch.empty(n * n // 8, device=dev, dtype=torch.uint8)
    alpha = torch.ones(1, device=dev)
    q4 = [torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8) for _ in range(NCOLD)]
    e4 = [torch.randint(118, 132, (n, n // 32), device=dev, dtype=torch.uint8) for _ in range(NCOLD)]
    y8 = torch.empty(n * n, device=dev, dtype=torch.uint8)
    rs = torch.empty(n * n // 32, device=dev, dtype=torch.uint8)
    cs = torch.empty(n * n // 32, device=dev, dtype=torch.uint8)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    I = ctypes.c_int64
    st = ctypes.c_void_p(0)
    QUEST, ABSMAX = 0, 1   # include/qutlass_amd.h

    def ops(lib):
        return {
            "fusedQuantizeMx H32 abs_max": lambda i: lib.qutlass_amd_fused_quantize_mx(P(xs[i]), P(hs[32]), 32, I(n * n), ABSMAX, P(out), P(sf), None, st),
            "fusedQuantizeMx H32 quest": lambda i: lib.qutlass_amd_fused_quantize_mx(P(xs[i]), P(hs[32]), 32, I(n * n), QUEST, P(out), P(sf), None, st),
            "fusedQuantizeMx H32 quest+mask": lambda i: lib.qutlass_amd_fused_quantize_mx(P(xs[i]), P(hs[32]), 32, I(n * n), QUEST, P(out), P(sf), P(mask), st),
            "fusedQuantizeNv H16 abs_max": lambda i: lib.qutlass_amd_fused_quantize_nv(P(xs[i]), P(hs[16]), 16, I(n * n), ABSMAX, P(gs), P(out), P(sf), st),
            "fusedQuantizeMx H64 abs_max": lambda i: lib.qutlass_amd_fused_quantize_mx(P(xs[i]), P(hs[64]), 64, I(n * n), ABSMAX, P(out), P(sf), None, st),
            "fusedQuantizeMx H128 abs_max": lambda i: lib.qutlass_amd_fused_quantize_mx(P(xs[i]), P(hs[128]), 128, I(n * n), ABSMAX, P(out), P(sf), None, st),
            "backward_t_bf16": lambda i: lib.qutlass_amd_backward_t_bf16(P(xs[i]), P(hs[32]), I(1), I(n), I(n), P(out), P(sf), st),
            "backward_qt_bf16": lambda i: lib.qutlass_amd_backward_qt_bf16(P(q4[i]), P(e4[i]), P(hs[32]), P(alpha), I(1), I(n), I(n), P(out), P(sf), st),
            "backward_bf16_square_double_mxfp8": lambda i: lib.qutlass_amd_backward_bf16_square_double_mxfp8(P(xs[i]), I(n), I(n), P(y8), P(rs), P(cs), st),
            "mxfp4_transpose_mxfp8": lambda i: lib.qutlass_amd_mxfp4_transpose_mxfp8(P(q4[i]), P(e4[i]), I(n), I(n), P(y8), P(rs), st),
        }

    def time_us(fn, cold, reps=400):
        for k in range(50):
            fn(k % NCOLD if cold else 0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(reps):
            rc = fn(k % NCOLD if cold else 0)
        b.record()
        torch.cuda.synchronize()
        assert rc == 0
        return a.elapsed_time(b) * 1e3 / reps

    oo, nn = ops(old), ops(new)
    print("%-40s %10s %10s %10s %10s" % ("op (%d x %d)" % (n, n), "old warm", "new warm", "old cold", "new cold"))
    for name in oo:
        r = []
        for cold in (False, True):
            to = tn = 1e9
            for rep in range(3):   # interleaved, best of 3
                to = min(to, time_us(oo[name], cold))
                tn = min(tn, time_us(nn[name], cold))
            r += [to, tn]
        print("%-40s %10.2f %10.2f %10.2f %10.2f" % (name, *r))

main()
