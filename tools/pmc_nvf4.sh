#!/bin/bash
# SQ / LDS counters of the persistent NVFP4 kernel (4096^3, 8 launches): separate rocprofv3 --pmc passes over a small torch driver (run on the GPU box):
#     tools/pmc_nvf4.sh [outdir]
OUT=${1:-gpurun_out/pmc_nvf4}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/pmc_nvf4_driver.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, qutlass_amd as q
from qutlass_amd.utils import to_blocked
dev = torch.device("cuda", 0); torch.manual_seed(0)
h = torch.ones(1, 1)
while h.shape[0] < 16: h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
h = (h * 16 ** -0.5).to(torch.bfloat16).to(dev)
a = torch.randn(4096, 4096, dtype=torch.bfloat16, device=dev) * 25.0
b = torch.randn(4096, 4096, dtype=torch.bfloat16, device=dev) * 25.0
gs = torch.tensor([1.0], device=dev)
aq, asf = q.fusedQuantizeNv(a, h, gs); bq, bsf = q.fusedQuantizeNv(b, h, gs)
asf, bsf = to_blocked(asf), to_blocked(bsf)
for _ in range(8): q.matmul_nvf4_bf16_tn(aq, bq, asf, bsf, gs)
torch.cuda.synchronize()
PY
run() { rocprofv3 --pmc $2 -d $R/$OUT/$1 -o p -- python /tmp/pmc_nvf4_driver.py > $R/$OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"
cd $R; python tools/rocprof_summary.py $OUT/*/p_results.db > $OUT/summary.txt 2>&1; grep -v "^$\|copyBuffer\|at::native\|fused_quantize\|to_blocked" $OUT/summary.txt | cut -c1-170
