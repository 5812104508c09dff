#!/bin/bash
# Round-3 GPU session H: rocprofv3 --kernel-trace --stats of bench_configs.py (every product kernel with call count and traced duration) + of the fused-provider sweep (decode kernels).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3h; mkdir -p $O; R=$(pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/bc -o p -- python $R/bench_configs.py > $R/$O/bench_configs.jsonl 2> $R/$O/bench_configs.err; echo "bench_configs rc=$?"
rocprofv3 --kernel-trace --stats -d $R/$O/ab -o p -- python $R/tools/ab_blocked_quant.py > $R/$O/ab.txt 2> $R/$O/ab.err; echo "ab rc=$?"
cd $R
python tools/rocprof_summary.py $O/bc/p_results.db > $O/bc_kernel_stats.txt 2>&1; head -60 $O/bc_kernel_stats.txt | cut -c1-200
python tools/rocprof_summary.py $O/ab/p_results.db > $O/ab_kernel_stats.txt 2>&1; grep -E "fusedq|fused_quantize|hetero" $O/ab_kernel_stats.txt | cut -c1-200 | head -20
rm -rf $O/bc $O/ab
