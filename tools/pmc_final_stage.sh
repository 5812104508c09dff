#!/bin/bash
# [r5] SQ / SQC counter passes of the persistent MXFP4 kernel on a SHORT-K shape (4096 x 4096 x 512: stage 0 + the last stage per tile), where the last stage dominates
OUT=${1:-gpurun_out/pmc_fs}; R=${GRAFT_REPO_ROOT:-$(pwd)}; K=${2:-512}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*\(ICACHE\|IFETCH\|INST_LEVEL\)[A-Z_0-9]*" | sort -u > $R/$OUT/counters_icache.txt
run() { rocprofv3 --pmc $2 -d $R/$OUT/$1 -o p -- $R/tests/native/qamd_check one 90 4096 4096 $K > $R/$OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run sq2 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
run sq3 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM"
run sq4 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVES"
cd $R; python tools/rocprof_summary.py $OUT/*/p_results.db > $OUT/summary.txt 2>&1; cat $OUT/counters_icache.txt | tr '\n' ' '; echo; grep -v "copyBuffer\|^$\|calls\|dispatches" $OUT/summary.txt | cut -c1-110
