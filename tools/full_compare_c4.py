#!/usr/bin/env python3
"""BASELINE.json configs[3] compared IN FULL: matmul_nvf4_bf16_tn 8192 x 8192 x 8192 on the GPU
against the reference's own test method restated in oracle/dequant_matmul.py (tests/nvfp4_test.py:80-110, :214-224: dequantise both
operands, a_dq @ b_dq.T in fp64, cast to bf16, `out.equal(ref)`) on the host cores -- all 67 108 864 outputs, not sampled rows.
Also cross-checks 64 rows of that torch oracle against the pinned C oracle.  Test infrastructure; prints one JSON line.

    python tools/full_compare_c4.py > gpurun_out/full_compare_c4.json       (about 1-2 minutes of host time on the GPU box)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def hadamard(n, device):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(device)


def main():
    import oracle
    import qutlass_amd as q
    from oracle import dequant_matmul as dm
    from qutlass_amd.utils import to_blocked

    dev = torch.device("cuda", 0)
    m = n = k = 8192
    torch.manual_seed(4)
    gs = torch.tensor([1.0], device=dev)
    h16 = hadamard(16, dev)
    a = torch.randn(m, k, dtype=torch.bfloat16, device=dev) * 25.0
    b = torch.randn(n, k, dtype=torch.bfloat16, device=dev) * 25.0
    a_q, a_s = q.fusedQuantizeNv(a, h16, gs)
    b_q, b_s = q.fusedQuantizeNv(b, h16, gs)
    del a, b
    out = q.matmul_nvf4_bf16_tn(a_q, b_q, to_blocked(a_s), to_blocked(b_s), torch.tensor([1.0], device=dev))
    torch.cuda.synchronize()
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.perf_counter()
    ref = dm.dequant_matmul_nvfp4(a_q.cpu(), a_s.cpu(), b_q.cpu(), b_s.cpu(), 1.0, torch.float64)
    t1 = time.perf_counter()
    got = out.cpu()
    neq = int((got.view(torch.int16) != ref.view(torch.int16)).sum())
    # the torch oracle itself against the pinned C oracle on 64 rows
    rows = sorted(set([0, 1, 255, 256, 4095, 4096, 8191] + list(np.random.default_rng(0).integers(0, m, 57))))
    u8 = lambda t: t.cpu().contiguous().view(torch.uint8).numpy()
    cref = oracle.gemm_blockscaled(oracle.KIND_NVFP4, u8(a_q)[rows], u8(b_q), oracle.to_blocked(u8(a_s)[rows, : k // 16]), oracle.to_blocked(u8(b_s)[:n, : k // 16]), 1.0, len(rows), n, k)
    c_vs_torch = int((cref != ref[rows].view(torch.uint16).numpy()).sum())
    print(json.dumps({"config": "C4 matmul_nvf4_bf16_tn 8192x8192x8192, operands = fusedQuantizeNv(H16, abs_max, global_scale 1) of randn*25 (seed 4)",
                      "outputs_compared": m * n, "bit_mismatches_vs_fp64_dequant_matmul_oracle": neq, "equal": neq == 0,
                      "oracle": "oracle/dequant_matmul.py dequant_matmul_nvfp4 (fp64, host cores)", "oracle_seconds": round(t1 - t0, 1), "host_threads": torch.get_num_threads(),
                      "c_oracle_rows_checked": len(rows), "c_oracle_vs_torch_oracle_mismatches": c_vs_torch}))
    return 0 if neq == 0 and c_vs_torch == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
