#!/bin/bash
# Round-2 GPU session E: full GPU suite on the current build, mid-batch ring depth, bench_configs (cold, large), rocprofv3 evidence.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); O=gpurun_out/r2e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_gpu.log
timeout 600 tests/native/qamd_check splitk > $O/splitk.log 2>&1; echo "splitk rc=$?"
tail -1 $O/splitk.log; grep BENCH $O/splitk.log | grep "auto\|depth" | grep "N=4096 K=14336\|N=8192 K=8192"
timeout 900 python bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "bench_configs rc=$?"
grep "cold" $O/bench_configs.jsonl | cut -c1-170
# rocprofv3: kernel trace + HBM-side traffic of bench.py, then bench.py with the fresh counters attached
bash tools/pmc_bench.sh $O/pmc_bench > $O/pmc_bench.log 2>&1; echo "pmc_bench rc=$?"
tail -3 $O/pmc_bench.log | cut -c1-600
# SQ counters of the persistent deep kernel (separate passes, counters only)
cd /tmp; export TMPDIR=/tmp
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set -d $R/$O/sq_$n -o p -- $R/tests/native/qamd_check one 90 > $R/$O/sq_$n.log 2>&1
done
cd $R
python tools/rocprof_summary.py $O/sq_*/p_results.db > $O/sq_counters_deepp.txt 2>&1
cat $O/sq_counters_deepp.txt | grep -v "^$" | head -40
