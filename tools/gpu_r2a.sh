#!/bin/bash
# Round-2 GPU session A: persistent deep kernel (parity + timing), power traces, full GPU test suite, bench lines.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2a; mkdir -p $O
rocm-smi --showpower --showclocks > $O/smi_idle.txt 2>&1
timeout 240 tests/native/qamd_check deepp > $O/deepp_check.log 2>&1; echo "deepp rc=$?" | tee $O/rc.txt
tail -4 $O/deepp_check.log
timeout 200 tests/native/qamd_check deeppbench > $O/deepp_bench.log 2>&1; echo "deeppbench rc=$?" | tee -a $O/rc.txt
cat $O/deepp_bench.log | grep BENCH
timeout 150 tests/native/qamd_check power > $O/power_ubench.log 2>&1; echo "power rc=$?" | tee -a $O/rc.txt
timeout 150 tests/native/qamd_check gpower > $O/power_gemm.log 2>&1; echo "gpower rc=$?" | tee -a $O/rc.txt
grep "POWER" $O/power_ubench.log $O/power_gemm.log
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -5 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cat $O/bench.json
timeout 900 python bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "bench_configs rc=$?" | tee -a $O/rc.txt
grep -c config $O/bench_configs.jsonl
