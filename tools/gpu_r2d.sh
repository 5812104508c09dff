#!/bin/bash
# Round-2 GPU session D: fused split-K (parity + timing vs two-launch + tile x split sweep), e5m2 operand tests, full GPU suite.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2d; mkdir -p $O
timeout 600 tests/native/qamd_check splitk > $O/splitk.log 2>&1; echo "splitk rc=$?"
grep -c " OK " $O/splitk.log; grep "FAIL\|mismatch row" $O/splitk.log | head -10; tail -1 $O/splitk.log
grep BENCH $O/splitk.log | grep "auto" 
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest_gpu.log
timeout 900 python bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "bench_configs rc=$?"
grep "C5 matmul" $O/bench_configs.jsonl | cut -c1-200
