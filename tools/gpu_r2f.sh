#!/bin/bash
# Round-2 GPU session F: whole-line (pair-wise) epilogue of the persistent deep kernel: parity, timing, phase trace.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2f; mkdir -p $O
timeout 300 tests/native/qamd_check deepp > $O/deepp_check.log 2>&1; echo "deepp rc=$?"
grep -c " OK " $O/deepp_check.log; grep "FAIL\|mismatch row" $O/deepp_check.log | head; tail -1 $O/deepp_check.log
timeout 200 tests/native/qamd_check deeppbench > $O/deepp_bench.log 2>&1; echo "deeppbench rc=$?"
grep BENCH $O/deepp_bench.log
timeout 100 tests/native/qamd_check deepptrace > $O/deepp_trace.log 2>&1
grep -v "^DEVICE" $O/deepp_trace.log | head -14
timeout 300 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_launch_us'])"
