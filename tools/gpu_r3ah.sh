#!/bin/bash
# Round-3 GPU session AH: in-stream steady-state timing (C++ harness, no Python in the loop) of the ring (73) against the pipelined (24) 128x128 schedule for MXFP4 on
# grids of 192 ... 256 tiles, K = 2048 ... 8192 -- the calibration's Python-timed columns disagree with the burst-timed sweep there.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3ah; mkdir -p $O
export QAMD_STEADY_MS=60
for shp in "1024 4096 4096" "896 4096 4096" "1024 3584 4096" "512 8192 4096" "2048 2048 4096" "768 4096 4096" "1024 4096 2048" "1024 4096 6144" "1024 4096 8192" "1024 4096 14336" "512 6144 4096" "640 6144 4096"; do
  for v in 73 24 0; do echo "== $v $shp"; timeout 60 tests/native/qamd_check one $v $shp 2>&1 | tail -1; done
done > $O/ring_vs_pipelined.txt 2>&1
cat $O/ring_vs_pipelined.txt
