#!/bin/bash
# Build the library with different cache-policy bits on the operand LDS-DMA loads and time the headline GEMM in the
# steady state (run on the GPU box).   tools/dma_aux_sweep.sh "0 2 1 16 17 18"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for aux in ${1:-0 2 1 17}; do
  mkdir -p /tmp/aux$aux
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DQAMD_DMA_AUX=$aux $R/qutlass_amd/csrc/capi.hip -o /tmp/aux$aux/libqutlass_amd.so 2>&1 | grep -E "error" | head -3
  echo "== QAMD_DMA_AUX=$aux"
  LD_LIBRARY_PATH=/tmp/aux$aux QAMD_STEADY_MS=80 $R/tests/native/qamd_check one 30 4096 4096 4096 | grep BENCH | cut -c1-150
  LD_LIBRARY_PATH=/tmp/aux$aux QAMD_STEADY_MS=80 $R/tests/native/qamd_check one 30 8192 8192 8192 | grep BENCH | cut -c1-150
done
