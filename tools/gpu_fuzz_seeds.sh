#!/bin/bash
# extra fuzz sweeps of the GPU parity tests (tests/test_gpu_fuzz.py, QAMD_FUZZ_SEED = 3 ... 14) on the current build
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/fuzz_seeds; mkdir -p $O; : > $O/summary.txt
for s in ${FUZZ_SEEDS:-3 4 5 6 7 8 9 10 11 12 13 14}; do
  QAMD_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $O/seed_$s.log 2>&1; echo "seed $s rc=$? $(tail -1 $O/seed_$s.log)" | tee -a $O/summary.txt
done
