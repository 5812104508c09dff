#!/bin/bash
# Round-3 GPU session Q: tile order / resident workgroups of the rewritten bwd_quant_t_kernel side by side (tools/ab_multi.py)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r3q}; mkdir -p $O
L=build/ab
for n in 4096 8192; do
timeout 900 python tools/ab_multi.py $n $L/libqutlass_amd_old.so $L/lib_d1.so $L/lib_d1.so@QAMD_BWD_TMB=2 $L/lib_d1.so@QAMD_BWD_TMB=4 $L/lib_d1.so@QAMD_BWD_TMB=8 $L/lib_d1.so@QAMD_BWD_TMB=16 $L/lib_d1.so@QAMD_BWD_TMB=1000 $L/lib_d1.so@QAMD_BWD_TMB=4,QAMD_BWD_WGS=3 $L/lib_d1.so@QAMD_BWD_TMB=16,QAMD_BWD_WGS=3 > $O/ab_multi_$n.txt 2>&1; echo "rc=$?"; grep -v "amdgpu.ids\|mxfp4_transpose\|^   lib.*warm   2[5].4\|warm    8.6[0-9] us   cold   10.0" $O/ab_multi_$n.txt
done
