#!/bin/bash
# Round-3 GPU session J: NVFP4 256x128 tile on four waves for half-chip outputs.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3j; mkdir -p $O
QAMD_STEADY_MS=30 timeout 600 tests/native/qamd_check nvhalf > $O/native_nvhalf.log 2>&1; echo "nvhalf rc=$?"
grep -E "BENCH|CHECK" $O/native_nvhalf.log | awk '/BENCH/ {printf "%-45s %s us %s TF\n", $2" "$3" "$4" "$5, $(NF-3), $(NF-1)} /CHECK/ {print}'
