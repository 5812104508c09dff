"""backward_qt_bf16: the whole-line panel kernel (lab option bwd_variant = 4) must return the bytes of the round-3 kernel (bwd_variant = 1, which the GPU suite holds to the
oracle) on full, ragged and batched shapes, under special scale bytes and a rotation that is not a Hadamard matrix.     python tools/check_qt_panel.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _benchlib as lab
from ab_bwd import hadamard

VARIANTS = [int(v) for v in os.environ.get("QT_VARIANTS", "4,5,6,7,8").split(",")]
SHAPES = [(1, 256, 256), (1, 32, 128), (1, 96, 384), (2, 288, 640), (1, 4096, 4096), (3, 1056, 1152), (1, 8192, 128), (1, 64, 8192), (1, 2080, 5248), (1, 8192, 8192), (5, 32, 128), (1, 36864 // 4, 1408)]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    bad = 0
    for hname in ("hadamard", "random"):
        h = hadamard(32, dev) if hname == "hadamard" else (torch.randn(32, 32, device=dev, generator=g) * 0.2).to(torch.bfloat16)
        for alpha_v in (0.75, 3.1e-5):
            alpha = torch.tensor([alpha_v], device=dev)
            for (B, N, M) in SHAPES:
                for scales in ("usual", "wide", "special"):
                    xq = torch.randint(0, 256, (B, N, M // 2), dtype=torch.uint8, device=dev, generator=g)
                    if scales == "usual":
                        xs = torch.randint(118, 134, (B, N, M // 32), dtype=torch.uint8, device=dev, generator=g)
                    elif scales == "wide":
                        xs = torch.randint(1, 255, (B, N, M // 32), dtype=torch.uint8, device=dev, generator=g)
                    else:
                        xs = torch.randint(120, 132, (B, N, M // 32), dtype=torch.uint8, device=dev, generator=g)
                        pick = torch.rand(xs.shape, device=dev, generator=g)
                        xs = torch.where(pick < 0.02, torch.zeros_like(xs), xs)
                        xs = torch.where(pick > 0.985, torch.full_like(xs, 255), xs)
                    with lab.forced(bwd_variant=1):
                        q1, s1 = lab.backward_qt_bf16(xq, xs, h, alpha)
                    for v in VARIANTS:
                        with lab.forced(bwd_variant=v):
                            q4, s4 = lab.backward_qt_bf16(xq, xs, h, alpha)
                        torch.cuda.synchronize()
                        dq, ds = int((q1 != q4).sum()), int((s1 != s4).sum())
                        if dq or ds:
                            bad += 1
                            idx = (q1 != q4).nonzero()[:4].tolist()
                            print(f"DIFF v{v} h={hname} alpha={alpha_v} B={B} N={N} M={M} scales={scales}: {dq} code bytes, {ds} scale bytes differ; first {idx}", flush=True)
    print(f"variants {VARIANTS} vs the round-3 kernel: %s" % ("ALL EQUAL" if not bad else f"{bad} cases differ"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
