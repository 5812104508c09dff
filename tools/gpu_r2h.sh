#!/bin/bash
# Round-2 GPU session H: persistent deep schedule for MXFP8 -- parity and timing; MXFP8 GPU tests.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2h; mkdir -p $O
timeout 400 tests/native/qamd_check deepp8 > $O/deepp8.log 2>&1; echo "deepp8 rc=$?"
grep "CHECK\|BENCH\|SUMMARY" $O/deepp8.log | cut -c1-220
timeout 900 python -m pytest tests -m gpu -q -k "mxf8 or mxfp8 or c5 or fuzz" > $O/pytest_fp8.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_fp8.log
