#!/bin/bash
# Round-3 GPU session AB: NVFP4 plan (small-output cost model + split-K) -- GPU suite, calibration after (auto column), dip scan of nvf4, NVFP4 batch sweep.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3ab; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for s in 1 2 3; do QAMD_FUZZ_SEED=$s timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k nvf4 > $O/fuzz_$s.log 2>&1; echo "fuzz $s rc=$?"; tail -1 $O/fuzz_$s.log; done
timeout 600 python tools/calib_nv_small.py > $O/calib_nv_small_after.txt 2> $O/calib.err; echo "calib rc=$?"
timeout 600 python tools/dip_scan.py nvf4 > $O/dip_scan_nvf4_after.txt 2> $O/dip.err; echo "dip rc=$?"; tail -1 $O/dip_scan_nvf4_after.txt
timeout 600 python benchmarks/bench_mxfp4_mi355x.py --format nvfp4 --had 16 --model Llama-3-8B --max-batch 8192 --reps 30 > $O/bench_sweep_nvfp4_llama3_8b.txt 2> $O/sweep.err; echo "sweep rc=$?"
