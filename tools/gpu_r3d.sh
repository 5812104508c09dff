#!/bin/bash
# Round-3 GPU session D: timing-only ablations of the persistent kernel (alpha = 1, stores spread over two stages).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3d; mkdir -p $O
QAMD_STEADY_MS=30 timeout 600 tests/native/qamd_check spread > $O/native_spread.log 2>&1; echo "spread rc=$?"
grep BENCH $O/native_spread.log | awk '{printf "%-45s %s us %s TF\n", $2" "$3" "$4" "$5, $(NF-3), $(NF-1)}'
