#!/bin/bash
# Round-2 GPU session B: persistent deep kernel (parity + timing + phase trace), GEMM power trace, full GPU test suite, bench lines.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2b; mkdir -p $O
timeout 300 tests/native/qamd_check deepp > $O/deepp_check.log 2>&1; echo "deepp rc=$?" | tee $O/rc.txt
grep -c "^PASS\|^ok\|PASS" $O/deepp_check.log; grep -i "fail\|mismatch" $O/deepp_check.log | head -20; tail -3 $O/deepp_check.log
timeout 200 tests/native/qamd_check deeppbench > $O/deepp_bench.log 2>&1; echo "deeppbench rc=$?" | tee -a $O/rc.txt
grep BENCH $O/deepp_bench.log
timeout 100 tests/native/qamd_check deepptrace > $O/deepp_trace.log 2>&1; echo "deepptrace rc=$?" | tee -a $O/rc.txt
grep -v "^DEVICE" $O/deepp_trace.log
timeout 150 tests/native/qamd_check gpower > $O/power_gemm.log 2>&1; echo "gpower rc=$?" | tee -a $O/rc.txt
grep "GPOWER" $O/power_gemm.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -15 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
cat $O/bench.json
timeout 900 python bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "bench_configs rc=$?" | tee -a $O/rc.txt
tail -3 $O/bench_configs.err; grep -c config $O/bench_configs.jsonl
