#!/bin/bash
# Round-3 GPU session AE: calibration of the MX GEMMs' small / mid batch dispatch (tools/calib_mx_small.py).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3ae; mkdir -p $O
timeout 900 python tools/calib_mx_small.py > $O/calib_mx_small.txt 2> $O/calib.err; echo "calib rc=$?"; wc -l $O/calib_mx_small.txt; tail -3 $O/calib.err
