#!/bin/bash
# SQ counters of the persistent MXFP8 kernel, TN and the (K, M) operand (ds_read_b64_tr_b8 fragment reads): separate rocprofv3 --pmc
# passes over `qamd_check onefp8 [nn]` (run on the GPU box): tools/pmc_fp8.sh [outdir]
OUT=${1:-gpurun_out/pmc_fp8}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
run() { rocprofv3 --pmc $3 -d $R/$OUT/$1 -o p -- $R/tests/native/qamd_check onefp8 $2 > $R/$OUT/$1.log 2>&1; }
for op in tn nn; do
  run ${op}_sq1 $op "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8"
  run ${op}_sq2 $op "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"
done
cd $R; python tools/rocprof_summary.py $OUT/*/p_results.db > $OUT/summary.txt 2>&1; grep -v "^$\|copyBuffer\|transpose_u8\|gemm_mx_kernel" $OUT/summary.txt | cut -c1-170
