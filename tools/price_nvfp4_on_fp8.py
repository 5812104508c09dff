"""What an FP8-MFMA path for matmul_nvf4_bf16_tn would cost in accuracy (VERDICT r4 item 9 / SURVEY section 7 option ii), priced on the CPU: the only way to run NVFP4
at more than the f16 MFMA's 2.5 PFLOP/s on gfx950 is v_mfma_scale_f32_*_f8f6f4 with fp8 operands, whose block scale is e8m0 per 32 -- so every NVFP4 element
(e2m1 code x e4m3 scale of its 16-group = up to 6 significant bits) has to become ONE e4m3 (4 significant bits) under a power-of-two scale shared by 32 elements.
This script does exactly that rounding (best case: per-32 power of two chosen so that the block maximum lands in e4m3's top binade, round-to-nearest-even) on
operands produced by the NVFP4 quantizer's arithmetic (randn * 25, Hadamard 16, abs_max), and reports the norm-wise error of the product against the exact one.
    python tools/price_nvfp4_on_fp8.py          (CPU only, ~20 s)"""
import math
import torch

E2M1 = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def hadamard(n):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return h * n ** -0.5


def quant_nvfp4(x):
    """rows of 16-groups -> dequantised NVFP4 values (fp32), abs_max scale in e4m3 (global scale 1), RNE to e2m1: the arithmetic of fusedQuantizeNv's abs_max arm"""
    m, k = x.shape
    g = (x.view(m, k // 16, 16) @ hadamard(16)).float()
    amax = g.abs().amax(-1, keepdim=True)
    sc = (amax / 6.0).to(torch.float8_e4m3fn).float()
    y = torch.where(sc > 0, g / sc, torch.zeros_like(g))
    idx = (y.abs().unsqueeze(-1) - E2M1).abs().argmin(-1)          # nearest code (ties do not matter for a pricing run)
    return (E2M1[idx] * y.sign() * sc).view(m, k)


def to_fp8_blocks(v):
    """v: dequantised NVFP4 (exact in fp32) -> one e4m3 per element under an e8m0 scale per 32, dequantised again"""
    m, k = v.shape
    b = v.view(m, k // 32, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(amax)) - 8.0                         # block maximum in [256, 512): e4m3's top binade (max finite 448 -> saturates above)
    s = torch.exp2(e)
    q = (b / s).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    return (q * s).view(m, k)


def main():
    torch.manual_seed(0)
    for (m, n, k) in [(1024, 1024, 2048), (512, 512, 8192)]:
        a = quant_nvfp4(torch.randn(m, k) * 25.0)
        b = quant_nvfp4(torch.randn(n, k) * 25.0)
        exact = a.double() @ b.double().t()
        a8, b8 = to_fp8_blocks(a), to_fp8_blocks(b)
        ea = ((a8 - a).norm() / a.norm()).item()
        approx = a8.double() @ b8.double().t()
        err = ((approx - exact).norm() / exact.norm()).item()
        one = ((a8.double() @ b.double().t() - exact).norm() / exact.norm()).item()
        nz = (a8 != a).float().mean().item()
        print(f"{m} x {n} x {k}: {100 * nz:.1f} % of the elements are not representable; operand error {ea:.4f} (norm-wise); product error {err:.4f} with both operands rounded, "
              f"{one:.4f} with one -- the bar is 1e-2")


main()
