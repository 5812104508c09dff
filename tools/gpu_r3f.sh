#!/bin/bash
# Round-3 GPU session F: pipelined-ring variants for one-workgroup-per-CU tiles (reads-first order, 8-wave 128x128) -- parity + steady-state A/B.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3f; mkdir -p $O
QAMD_STEADY_MS=30 timeout 600 tests/native/qamd_check readsfirst > $O/native_readsfirst.log 2>&1; echo "readsfirst rc=$?"
grep -E "BENCH|CHECK" $O/native_readsfirst.log | awk '/BENCH/ {printf "%-45s %s us %s TF\n", $2" "$3" "$4" "$5, $(NF-3), $(NF-1)} /CHECK/ {print}'
