"""Several builds of libqutlass_amd.so side by side on the QAT-backward streaming ops (same box, interleaved, best of 3):
    python tools/ab_multi.py n lib1.so lib2.so ...        (a lib may carry an environment setting: path.so@QAMD_BWD_WGS=3)
Warm = one input replayed; cold = inputs rotated through > 1 GiB."""
import ctypes, os, sys, torch

def main():
    n = int(sys.argv[1])
    specs = sys.argv[2:]
    libs = []
    for sp in specs:
        path, _, env = sp.partition("@")
        libs.append((sp, ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL), dict(kv.split("=") for kv in env.split(",")) if env else {}))
    dev = torch.device("cuda:0")
    NCOLD = 40 if n <= 4096 else 12
    xs = [torch.randn(n, n, device=dev, dtype=torch.bfloat16) * 25 for _ in range(NCOLD)]
    h = (torch.randn(32, 32, device=dev) * 0.2).to(torch.bfloat16)
    out = torch.empty(n * n // 2, device=dev, dtype=torch.uint8)
    sf = torch.empty(n * n // 16, device=dev, dtype=torch.uint8)
    alpha = torch.ones(1, device=dev)
    q4 = [torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8) for _ in range(NCOLD)]
    e4 = [torch.randint(118, 132, (n, n // 32), device=dev, dtype=torch.uint8) for _ in range(NCOLD)]
    y8 = torch.empty(n * n, device=dev, dtype=torch.uint8)
    rs = torch.empty(n * n // 32, device=dev, dtype=torch.uint8)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    I = ctypes.c_int64
    st = ctypes.c_void_p(0)

    def ops(lib):
        return {
            "backward_t_bf16": lambda i: lib.qutlass_amd_backward_t_bf16(P(xs[i]), P(h), I(1), I(n), I(n), P(out), P(sf), st),
            "backward_qt_bf16": lambda i: lib.qutlass_amd_backward_qt_bf16(P(q4[i]), P(e4[i]), P(h), P(alpha), I(1), I(n), I(n), P(out), P(sf), st),
            "mxfp4_transpose_mxfp8": lambda i: lib.qutlass_amd_mxfp4_transpose_mxfp8(P(q4[i]), P(e4[i]), I(n), I(n), P(y8), P(rs), st),
        }

    def time_us(fn, cold, env, reps=300):
        for k, v in env.items():
            os.environ[k] = v
        for k in range(40):
            fn(k % NCOLD if cold else 0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(reps):
            rc = fn(k % NCOLD if cold else 0)
        b.record()
        torch.cuda.synchronize()
        for k in env:
            os.environ.pop(k, None)
        assert rc == 0
        return a.elapsed_time(b) * 1e3 / reps

    table = [(sp, ops(lib), env) for sp, lib, env in libs]
    for name in table[0][1]:
        print(f"{name} ({n} x {n})")
        best = {sp: [1e9, 1e9] for sp, _, _ in table}
        for rep in range(3):
            for sp, o, env in table:
                for ci, cold in enumerate((False, True)):
                    best[sp][ci] = min(best[sp][ci], time_us(o[name], cold, env))
        for sp, _, _ in table:
            print("   %-60s warm %7.2f us   cold %7.2f us" % (os.path.basename(sp), *best[sp]))

main()
