"""Old vs new libqutlass_amd.so on matmul_nvf4_bf16_tn (same box, interleaved, best of 3; operands quantised Gaussian):
    python tools/ab_nvf4.py old.so new.so          -- also checks that both libraries return the same bytes"""
import ctypes, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import qutlass_amd as q
from qutlass_amd.utils import to_blocked

def main():
    old, new = (ctypes.CDLL(p, mode=ctypes.RTLD_LOCAL) for p in sys.argv[1:3])
    dev = torch.device("cuda:0")
    h16 = torch.eye(16, dtype=torch.bfloat16, device=dev)
    gs = torch.tensor([1.0], device=dev)
    alpha = torch.tensor([1.0], device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr()); I = ctypes.c_int64; st = ctypes.c_void_p(0)
    shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (4096, 14336, 4096), (2048, 4096, 4096), (1024, 4096, 4096), (2048, 4096, 14336), (512, 4096, 4096), (128, 4096, 4096), (4096, 5120, 4096)]
    if len(sys.argv) > 3 and sys.argv[3] == "mid":   # the mid-batch shapes of the Llama-3-8B sweep
        shapes = [(m, n, k) for (n, k) in ((4096, 4096), (6144, 4096), (4096, 14336)) for m in (256, 512, 1024, 2048)]
    print("%-22s %10s %10s %8s  same bytes" % ("M x N x K", "old us", "new us", "ratio"))
    for (m, n, k) in shapes:
        torch.manual_seed(m + n + k)
        def mk(r):
            x = torch.randn(r, k, dtype=torch.bfloat16, device=dev)
            xq, xs = q.fusedQuantizeNv(x, h16, gs)
            return xq, to_blocked(xs)
        a, sa = mk(m); b, sb = mk(n)
        outs = []
        def run(lib):
            d = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            f = lambda: lib.qutlass_amd_matmul_nvf4_bf16_tn(P(a), P(b), P(sa), P(sb), P(alpha), P(d), I(m), I(n), I(k), st)
            assert f() == 0
            outs.append(d)
            return f
        fo, fn = run(old), run(new)
        same = torch.equal(outs[0], outs[1])
        reps = max(20, int(20000 / max(1.0, 2.0 * m * n * k / 1.3e9)))   # ~20 ms per measurement
        def t(f):
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        for _ in range(30): fn()   # clock ramp
        to = tn = 1e9
        for _ in range(3):
            to = min(to, t(fo)); tn = min(tn, t(fn))
        print("%-22s %10.2f %10.2f %8.3f  %s" % (f"{m}x{n}x{k}", to, tn, tn / to, same), flush=True)

main()
