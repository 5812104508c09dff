#!/bin/bash
# Round-3 GPU session T: NVFP4 GEMM with the MFMA / dequantisation interleave -- parity tests, then old vs new library (tools/ab_nvf4.py)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r3t}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "nvf4 or nvfp4 or Nv" > $O/pytest_nv.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_nv.log
timeout 900 python tools/ab_nvf4.py build/ab/libqutlass_amd_old.so qutlass_amd/libqutlass_amd.so > $O/ab_nvf4.txt 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $O/ab_nvf4.txt
