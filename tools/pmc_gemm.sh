#!/bin/bash
# PMC passes for one GEMM variant (run on the GPU box): tools/pmc_gemm.sh <variant> <outdir>
# Counters go in separate rocprofv3 runs (8 SQ slots per pass; FETCH_SIZE / WRITE_SIZE do not fit one TCC pass).
V=${1:-6}; OUT=${2:-gpurun_out/pmc}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 -d $R/$OUT/$1 -o p -- $R/tests/native/qamd_check one $V > $R/$OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
run sq3 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES"
run tcc1 "FETCH_SIZE"
run tcc2 "WRITE_SIZE"
run tcc3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run tcp1 "TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"
cd $R; python tools/rocprof_summary.py $OUT/*/p_results.db > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | grep -v "^$" | cut -c1-150 | head -80
