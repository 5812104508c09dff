#!/bin/bash
# Round-3 GPU session Q: lab variants of bwd_quant_t_kernel side by side (tools/ab_multi.py)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r3q}; mkdir -p $O
L=build/ab
for n in 4096 8192; do
timeout 900 python tools/ab_multi.py $n $L/libqutlass_amd_old.so $L/lib_cur.so $L/lib_noscale.so > $O/ab_multi_$n.txt 2>&1; echo "rc=$?"; grep -v "amdgpu.ids" $O/ab_multi_$n.txt
done
