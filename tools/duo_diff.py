"""Where does a lab variant of the MXFP4 GEMM disagree with the product's persistent kernel?  (debugging aid for csrc/lab/gemm_mx_duo.hip.h)
    python tools/duo_diff.py 88 256x256x512"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import _benchlib as lab

v = int(sys.argv[1]); m, n, k = (int(d) for d in sys.argv[2].split("x"))
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(m + n + k)
a = torch.randint(0, 256, (m, k // 2), dtype=torch.uint8, device=dev, generator=g)
b = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=dev, generator=g)
pad = lambda r: (r + 127) // 128 * 128
cb = (k // 32 + 3) // 4 * 4
sa = torch.randint(121, 130, (pad(m) * cb,), dtype=torch.uint8, device=dev, generator=g)
sb = torch.randint(121, 130, (pad(n) * cb,), dtype=torch.uint8, device=dev, generator=g)
alpha = torch.tensor([1.0], device=dev)
with lab.forced(gemm_variant=90):
    ref = lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha)
outs = []
for rep in range(3):
    with lab.forced(gemm_variant=v):
        outs.append(lab.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha))
torch.cuda.synchronize()
for rep, got in enumerate(outs):
    bad = (got.view(torch.int16) != ref.view(torch.int16))
    print(f"rep {rep}: {int(bad.sum())} of {bad.numel()} differ")
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print("  rows", rows[:64], "..." if len(rows) > 64 else "")
        print("  cols", cols[:64], "..." if len(cols) > 64 else "")
        idx = bad.nonzero()[:12].tolist()
        for r, c in idx:
            print(f"   [{r},{c}] got {float(got[r, c]):.6g} ref {float(ref[r, c]):.6g}")
print("run-to-run identical:", all(torch.equal(outs[0].view(torch.int16), o.view(torch.int16)) for o in outs[1:]))
