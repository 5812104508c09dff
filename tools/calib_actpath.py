"""The activation path y = Q(x h) W^T of one linear layer at decode batches: ONE launch (csrc/gemm_mx_fusedq.hip.h: the GEMM rotates and quantises its own A operand) against
TWO (fusedQuantizeMxBlocked + matmul_mxf4_bf16_tn) against the reference's THREE (fusedQuantizeMx + to_blocked + GEMM), GPU-only timing (HIP-graph replays) through the product
library; the last column is what qutlass_amd_activation_path_launches picks.  The GEMM behind the two-launch path got faster in round 6 (decode forms): this is the re-take the
rule's thresholds are read from.    python tools/calib_actpath.py > gpurun_out/calib_actpath.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _benchlib as lab   # (only for the lab column: the hand-off form of the one-launch layer, gemm_variant 580)
import qutlass_amd as q
from qutlass_amd import _lib
from qutlass_amd.utils import to_blocked
from _timing import graph_us

dev = torch.device("cuda:0")
torch.manual_seed(0)
h = torch.ones(1, 1)
while h.shape[0] < 32:
    h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
h = (h * 32 ** -0.5).to(torch.bfloat16).to(dev)
alpha = torch.ones(1, device=dev)
ms = [int(v) for v in os.environ.get("ACT_M", "1,4,8,16,32").split(",")]
nks = [tuple(int(d) for d in s.split("x")) for s in os.environ.get("ACT_NK", "4096x4096,6144x4096,14336x4096,4096x14336,2048x2048,8192x8192,4096x8192,4096x6144").split(",")]
print("# us per layer call (HIP-graph replays): one launch | two launches | three launches | GEMM alone || the rule's choice (launches)")
for (n, k) in nks:
    w = torch.randn(n, k, dtype=torch.bfloat16, device=dev) * 25
    wq, wsf = q.fusedQuantizeMx(w, h, method="abs_max")
    wsf = to_blocked(wsf)
    for m in ms:
        x = torch.randn(m, k, dtype=torch.bfloat16, device=dev) * 25
        t = {}
        for name, fn in (("one", lambda: q.fused_quantize_matmul_mxf4_bf16_tn(x, h, wq, wsf, alpha, method="abs_max", single_launch=True)),
                         ("two", lambda: q.fused_quantize_matmul_mxf4_bf16_tn(x, h, wq, wsf, alpha, method="abs_max", single_launch=False)),
                         ("three", lambda: q.matmul_mxf4_bf16_tn(*(lambda aq, asf: (aq, wq, to_blocked(asf), wsf, alpha))(*q.fusedQuantizeMx(x, h, method="abs_max"))))):
            try:
                t[name] = min(graph_us(fn, n=40) for _ in range(3))
            except Exception as e:   # the one-launch kernel has shape limits
                t[name] = float("nan")
        t["handoff"], eq = float("nan"), ""
        if os.environ.get("ACT_HANDOFF", "0") == "1" and m <= 16:
            try:
                two = q.fused_quantize_matmul_mxf4_bf16_tn(x, h, wq, wsf, alpha, method="abs_max", single_launch=False)
                with lab.forced(gemm_variant=580):
                    got = lab.fused_quantize_matmul_mxf4_bf16_tn(x, h, wq, wsf.view(torch.uint8), alpha, method="abs_max")
                    torch.cuda.synchronize()
                    eq = "equal" if torch.equal(got.view(torch.int16), two.view(torch.int16)) else "DIFFER (%d)" % int((got.view(torch.int16) != two.view(torch.int16)).sum())
                    t["handoff"] = min(graph_us(lambda: lab.fused_quantize_matmul_mxf4_bf16_tn(x, h, wq, wsf.view(torch.uint8), alpha, method="abs_max"), n=40) for _ in range(3))
            except Exception as e:
                eq = "error: %s" % str(e)[:60]
        aq, asf = q.fusedQuantizeMxBlocked(x, h, method="abs_max")
        t["gemm"] = min(graph_us(lambda: q.matmul_mxf4_bf16_tn(aq, wq, asf, wsf, alpha), n=40) for _ in range(3))
        rule = _lib.load().qutlass_amd_activation_path_launches(m, n, k, 32)
        print("N=%-6d K=%-6d M=%-3d | %6.2f | %6.2f | %6.2f | %6.2f || %d%s" % (n, k, m, t["one"], t["two"], t["three"], t["gemm"], rule,
              (" || hand-off form (lab) %6.2f %s" % (t["handoff"], eq)) if os.environ.get("ACT_HANDOFF", "0") == "1" else ""), flush=True)
