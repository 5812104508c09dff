#!/bin/bash
# Round-3 GPU session AA: NVFP4 split-K -- calibration of tile x split candidates for M = 16 ... 1024 (tools/calib_nv_small.py) and a parity spot check.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3aa; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -x -k "nvf4" > $O/pytest_nv.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_nv.log
timeout 900 python tools/calib_nv_small.py > $O/calib_nv_small.txt 2> $O/calib.err; echo "calib rc=$?"; wc -l $O/calib_nv_small.txt; tail -3 $O/calib.err
