#!/bin/bash
# Round-3 final GPU session: tools/gpu_r3g.sh (suite, smoke, bench.py, rocprofv3 kernel-trace + PMC passes, Llama-3-8B MXFP4 sweep, dip scan) on the last commit,
# plus the reference-shaped batch sweeps for the other two model families and for NVFP4.
cd ${GRAFT_REPO_ROOT:-.}
export QAMD_SESSION=${QAMD_SESSION:-r3z_final}
bash tools/gpu_r3g.sh > gpurun_out/${QAMD_SESSION}_main.log 2>&1; echo "main rc=$?"
O=gpurun_out/$QAMD_SESSION
grep -E "rc=|passed|failed|value |smoke ok|flagged" gpurun_out/${QAMD_SESSION}_main.log | head -20
for model in Llama-3.1-70B Qwen3-32B; do
  timeout 600 python benchmarks/bench_mxfp4_mi355x.py --model $model --fused --max-batch 8192 --reps 30 > $O/bench_sweep_mxfp4_$model.txt 2>> $O/sweeps.err; echo "mxfp4 $model rc=$?"
done
for model in Llama-3-8B Llama-3.1-70B; do
  timeout 600 python benchmarks/bench_mxfp4_mi355x.py --format nvfp4 --had 16 --model $model --max-batch 8192 --reps 30 > $O/bench_sweep_nvfp4_$model.txt 2>> $O/sweeps.err; echo "nvfp4 $model rc=$?"
done
find $O -name "*.db" -size +8M -delete
