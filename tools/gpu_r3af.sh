#!/bin/bash
# Round-3 GPU session AF: MX small-output plan with the fitted-model correction + the wide-weight rule -- GPU suite, fuzz seeds, calibration after.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3af; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for s in 1 2; do QAMD_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $O/fuzz_$s.log 2>&1; echo "fuzz $s rc=$?"; tail -1 $O/fuzz_$s.log; done
timeout 900 python tools/calib_mx_small.py > $O/calib_mx_small_after.txt 2> $O/calib.err; echo "calib rc=$?"; wc -l $O/calib_mx_small_after.txt
