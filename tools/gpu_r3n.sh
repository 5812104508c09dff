#!/bin/bash
# Round-3 GPU session N: the VALU diet of the streaming kernels (VGPR-form rotation MFMAs, packed multiplies, block scale inside the convert) --
# full GPU suite on the new build, then old vs new library on one box (tools/ab_stream_ops.py) at 4096^2 and 8192^2.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r3n}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python tools/ab_stream_ops.py build/ab/libqutlass_amd_old.so qutlass_amd/libqutlass_amd.so 4096 > $O/ab_stream_ops_4096.txt 2>&1; echo "ab4096 rc=$?"; cat $O/ab_stream_ops_4096.txt
timeout 600 python tools/ab_stream_ops.py build/ab/libqutlass_amd_old.so qutlass_amd/libqutlass_amd.so 8192 > $O/ab_stream_ops_8192.txt 2>&1; echo "ab8192 rc=$?"; cat $O/ab_stream_ops_8192.txt
