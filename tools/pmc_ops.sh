#!/bin/bash
# HBM-side traffic of the HBM-bound ops (quantizers, to_blocked, QAT-backward data prep) from rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE in separate runs), next to their algorithmic bytes.  Run on the GPU box:
#     tools/pmc_ops.sh [outdir]   -> <outdir>/summary.txt
OUT=${1:-gpurun_out/pmc_ops}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/pmc_ops_driver.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, qutlass_amd as q
from qutlass_amd.utils import to_blocked
dev = torch.device("cuda", 0); torch.manual_seed(0)
def had(n):
    h = torch.ones(1, 1)
    while h.shape[0] < n: h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(dev)
x = torch.randn(4096, 4096, dtype=torch.bfloat16, device=dev) * 25.0
h32, h128, h16 = had(32), had(128), had(16)
gs = torch.tensor([1.0], device=dev); al3 = torch.tensor([3.0], device=dev)
for _ in range(10):
    xq, xs = q.fusedQuantizeMx(x, h32, method="abs_max")
    q.fusedQuantizeMx(x, h32, method="quest", return_mask=True)
    q.fusedQuantizeMx(x, h128, method="abs_max")
    q.fusedQuantizeNv(x, h16, gs)
    to_blocked(xs)
    xs2 = xs.view(torch.uint8).reshape(-1)[: 4096 * 128].reshape(4096, 128).contiguous().view(torch.float8_e8m0fnu)
    q.backward_t_bf16(x, h32)
    q.backward_qt_bf16(xq, xs2, h32, al3)
    q.backward_bf16_square_double_mxfp8(x)
    q.mxfp4_transpose_mxfp8(xq, xs2)
torch.cuda.synchronize()
PY
rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/fetch -o p -- python /tmp/pmc_ops_driver.py > $R/$OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/write -o p -- python /tmp/pmc_ops_driver.py > $R/$OUT/write.log 2>&1
cd $R
python tools/rocprof_summary.py $OUT/fetch/p_results.db $OUT/write/p_results.db 2>&1 | grep -E "qamd|calls|dispatches|==" > $OUT/summary.txt
cat $OUT/summary.txt | cut -c1-170
