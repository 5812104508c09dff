#!/bin/bash
# Builds the native on-device checker (test infrastructure).  Needs qutlass_amd/libqutlass_amd_bench.so (the LAB build: schedule variants, ablations) and
# oracle/libqutlass_oracle.so (built by __graft_entry__.build()).
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O2 -w -std=c++17 -x hip qamd_check.cpp probe.hip ubench.hip -o qamd_check -lrocm_smi64 \
  -L../../qutlass_amd -lqutlass_amd_bench -L../../oracle -lqutlass_oracle \
  -Wl,-rpath,'$ORIGIN/../../qutlass_amd:$ORIGIN/../../oracle'
echo built tests/native/qamd_check
