#!/bin/bash
# Round-3 GPU session I: 256x128 / 128x256 tiles on four waves for half-chip outputs.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3i; mkdir -p $O
QAMD_STEADY_MS=30 timeout 600 tests/native/qamd_check halftile > $O/native_halftile.log 2>&1; echo "halftile rc=$?"
grep -E "BENCH|CHECK" $O/native_halftile.log | awk '/BENCH/ {printf "%-45s %s us %s TF\n", $2" "$3" "$4" "$5, $(NF-3), $(NF-1)} /CHECK/ {print}'
