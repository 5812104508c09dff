#!/bin/bash
# What bounds the QAT-backward streaming ops at 8192^2?  rocprofv3 --pmc passes (separate runs, counters only) over a driver that calls the
# ops through the C ABI of ONE library:   tools/pmc_stream_ops.sh <lib.so> <outdir> [n]
LIB=${1:-qutlass_amd/libqutlass_amd.so}; OUT=${2:-gpurun_out/pmc_stream}; N=${3:-8192}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/pmc_stream_driver.py <<PY
import ctypes, torch
lib = ctypes.CDLL("$R/$LIB", mode=ctypes.RTLD_LOCAL)
dev = torch.device("cuda:0"); n = $N; torch.manual_seed(0)
x = torch.randn(n, n, device=dev, dtype=torch.bfloat16) * 25
h = (torch.randn(32, 32, device=dev) * 0.2).to(torch.bfloat16)
out = torch.empty(n * n // 2, device=dev, dtype=torch.uint8); sf = torch.empty(n * n // 16, device=dev, dtype=torch.uint8)
alpha = torch.ones(1, device=dev)
q4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8); e4 = torch.randint(118, 132, (n, n // 32), device=dev, dtype=torch.uint8)
y8 = torch.empty(n * n, device=dev, dtype=torch.uint8); rs = torch.empty(n * n // 32, device=dev, dtype=torch.uint8)
P = lambda t: ctypes.c_void_p(t.data_ptr()); I = ctypes.c_int64; st = ctypes.c_void_p(0)
for _ in range(6):
    assert lib.qutlass_amd_fused_quantize_mx(P(x), P(h), 32, I(n * n), 1, P(out), P(sf), None, st) == 0
    assert lib.qutlass_amd_backward_t_bf16(P(x), P(h), I(1), I(n), I(n), P(out), P(sf), st) == 0
    assert lib.qutlass_amd_backward_qt_bf16(P(q4), P(e4), P(h), P(alpha), I(1), I(n), I(n), P(out), P(sf), st) == 0
    assert lib.qutlass_amd_mxfp4_transpose_mxfp8(P(q4), P(e4), I(n), I(n), P(y8), P(rs), st) == 0
torch.cuda.synchronize()
PY
run() { rocprofv3 --pmc $2 -d $R/$OUT/$1 -o p -- python /tmp/pmc_stream_driver.py > $R/$OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
run sq3 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
run tcc1 "FETCH_SIZE"
run tcc2 "WRITE_SIZE"
run tcc3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run tcp1 "TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"
cd $R; python tools/rocprof_summary.py $OUT/*/p_results.db 2>&1 | grep -E "qamd|calls|dispatches|==" | grep -v "vectorized\|distribution\|rocclr" > $OUT/summary.txt; cut -c1-160 $OUT/summary.txt
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/tcc1 $OUT/tcc2 $OUT/tcc3 $OUT/tcp1   # the result databases are ~10 MB each: keep the summary and the logs only
