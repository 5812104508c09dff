#!/bin/bash
# ONE parametrised GPU session script (replaces the per-session tools/gpu_r*.sh of rounds 2-3):
#     gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <name> <step> [<step> ...]'
# writes everything under gpurun_out/<name>/ and prints a one-line verdict per step.  Steps:
#   testsk    the stream-K parity tests only, short timeout (a wrong flag protocol would hang)
#   tests4    tests/test_gpu_round4.py (fail-fast)            suite     the whole -m gpu suite
#   smoke     __graft_entry__.smoke()                         bench     bench.py (driver protocol) + a summary of its JSON line
#   pmc       tools/pmc_bench.sh (kernel-trace + FETCH_SIZE / WRITE_SIZE passes of the bench command + a fresh-traffic bench line)
#   abnv      tools/ab_nvpk.py --trace (persistent NVFP4 kernel vs per-tile kernel vs torch bf16; stage trace)
#   vendor    tools/vendor_ab.py (hipBLASLt block-scaled GEMM through torch beside ours, power / clock)
#   profnv    rocprofv3 --kernel-trace --stats + SQ busy counters of the NVFP4 GEMM at 8192^3 and 4096 x 28672 x 4096
#   dip       tools/dip_scan.py (8 % threshold)               sweeps    benchmarks/bench_mxfp4_mi355x.py for MXFP4 (fused) and NVFP4, Llama-3-8B
#   configs   bench_configs.py                                abmx      tools/ab_mxsk.py (MX persistent kernels: balanced / heterogeneous / stream-K)
#   calibnv   tools/calib_tiles.py nvf4 (forced tile candidates on the training grid of the NVFP4 tile rule)
#   abbwd     tools/ab_bwd.py (backward_t / backward_qt: wave-owned-lines kernel vs the round-3 kernel)     testbwd   the GPU tests of the QAT-backward ops
#   ablbwd    tools/ab_bwd_abl.py (where the time of the wave-owned backward kernel goes: loads / stores / arithmetic left out in turn)
#   abtr      tools/ab_transpose.py (mxfp4_transpose_mxfp8: one-shot kernel vs the wave-owned-lines kernels)
#   pmcstream tools/pmc_stream_ops.sh (SQ / TCC counters of the streaming ops at 8192^2)            hbm   tools/hbm_ceilings.py (trivial kernels at the ops' read : write mixes)
#   fullcmp   tools/full_compare.py (configs C2, C3, C5: every output against the fp64 dequant-matmul oracle)
#   absq / absf   tools/ab_sq_abl.py / ab_sf_stores.py (what the scale-byte stores of the transposing ops cost; square_double workgroup shapes)
#   testq     the GPU tests of the quantizers            tracenv   stage + hand-off trace of the persistent NVFP4 kernel (lab variant 44)
#   contention   tools/final_stage_contention.py (stage trace of the persistent MXFP4 kernel at 8 / 64 / 256 workgroups, same work per workgroup: is the final stage's cost the chip-wide store burst?)
#   ablib / ablibnv / ablibmx / ablibmid   two BUILDS of the library side by side (tools/ab_lib_shapes.py, ab_lib_gemm.py): copy the old one to
#             build/exp/libqamd_base.so first (or set AB_OLD / AB_NEW / AB_FMT); small + mid-size shapes, NVFP4 large, MXFP4 / MXFP8 large + headline
cd ${GRAFT_REPO_ROOT:-.}
NAME=${1:?session name}; shift
O=gpurun_out/$NAME; mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  t0=$(date +%s)
  case $step in
    testsk) timeout 240 python -m pytest tests/test_gpu_round4.py -x -q -k stream_k > $O/pytest_streamk.log 2>&1; echo "testsk rc=$?"; tail -5 $O/pytest_streamk.log ;;
    tests4) timeout 900 python -m pytest tests/test_gpu_round4.py -x -q > $O/pytest_round4.log 2>&1; echo "tests4 rc=$?"; tail -3 $O/pytest_round4.log ;;
    suite)  timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_gpu.log ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log ;;
    bench)  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
            python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'kernel_us', r.get('kernel_us'), 'cpu', d['cpu_baseline']['value'])
print('power', d.get('power', {}).get('timed_region'))
for k, v in (d.get('configs') or {}).items(): print(' ', k, v if 'us' not in v else (v['us'], v['roofline']['frac']))
PY
            ;;
    pmc)    bash tools/pmc_bench.sh $O/pmc_bench > $O/pmc_bench.log 2>&1; echo "pmc rc=$?"; head -20 $O/pmc_bench/summary.txt; cat $O/pmc_bench/traffic.json ;;
    abnv)   timeout 900 python tools/ab_nvpk.py --trace --shapes all > $O/ab_nvpk.txt 2> $O/ab_nvpk.err; echo "abnv rc=$?"; cat $O/ab_nvpk.txt; tail -3 $O/ab_nvpk.err ;;
    vendor) timeout 600 python tools/vendor_ab.py > $O/vendor_ab.txt 2> $O/vendor_ab.err; echo "vendor rc=$?"; cat $O/vendor_ab.txt; tail -3 $O/vendor_ab.err ;;
    profnv) R=$(pwd); (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_nv_trace -o p -- python $R/tools/ab_nvpk.py --shapes prof > $R/$O/prof_nv_trace.log 2>&1
              rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d $R/$O/prof_nv_pmc -o p -- python $R/tools/ab_nvpk.py --shapes prof > $R/$O/prof_nv_pmc.log 2>&1)
            python tools/rocprof_summary.py $O/prof_nv_trace/p_results.db $O/prof_nv_pmc/p_results.db > $O/prof_nv_summary.txt 2>&1; echo "profnv rc=$?"; head -40 $O/prof_nv_summary.txt ;;
    dip)    timeout 900 python tools/dip_scan.py > $O/dip_scan.txt 2> $O/dip_scan.err; echo "dip rc=$?"; tail -2 $O/dip_scan.txt ;;
    sweeps) timeout 900 python benchmarks/bench_mxfp4_mi355x.py --model Llama-3-8B --fused --vendor --max-batch 8192 --reps 30 > $O/bench_sweep_mxfp4_Llama-3-8B.txt 2> $O/sweeps.err; echo "sweep mxfp4 rc=$?"
            timeout 900 python benchmarks/bench_mxfp4_mi355x.py --format nvfp4 --had 16 --model Llama-3-8B --max-batch 8192 --reps 30 > $O/bench_sweep_nvfp4_Llama-3-8B.txt 2>> $O/sweeps.err; echo "sweep nvfp4 rc=$?"
            cat $O/bench_sweep_nvfp4_Llama-3-8B.txt ;;
    ablibmid) timeout 900 python tools/ab_lib_shapes.py ${AB_OLD:-build/exp/libqamd_base.so} ${AB_NEW:-qutlass_amd/libqutlass_amd.so} --fmt=mxf4 --fmt=mxf8 > $O/ab_lib_mid.txt 2> $O/ab_lib_mid.err; echo "ablibmid rc=$?"; cat $O/ab_lib_mid.txt; tail -2 $O/ab_lib_mid.err ;;
    ablibmx) timeout 900 python tools/ab_lib_gemm.py ${AB_OLD:-build/exp/libqamd_base.so} ${AB_NEW:-qutlass_amd/libqutlass_amd.so} > $O/ab_lib_gemm.txt 2> $O/ab_lib_gemm.err; echo "ablibmx rc=$?"; cat $O/ab_lib_gemm.txt; timeout 900 python tools/ab_lib_shapes.py ${AB_OLD:-build/exp/libqamd_base.so} ${AB_NEW:-qutlass_amd/libqutlass_amd.so} --big --fmt=${AB_FMT:-mxf4} > $O/ab_lib_mx.txt 2>> $O/ab_lib_gemm.err; cat $O/ab_lib_mx.txt; tail -3 $O/ab_lib_gemm.err ;;
    ablibnv) timeout 900 python tools/ab_lib_shapes.py ${AB_OLD:-build/exp/libqamd_base.so} ${AB_NEW:-qutlass_amd/libqutlass_amd.so} --big --fmt=nvf4 > $O/ab_lib_nv.txt 2> $O/ab_lib_nv.err; timeout 900 python tools/ab_lib_shapes.py ${AB_OLD:-build/exp/libqamd_base.so} ${AB_NEW:-qutlass_amd/libqutlass_amd.so} --fmt=nvf4 >> $O/ab_lib_nv.txt 2>> $O/ab_lib_nv.err; echo "ablibnv rc=$?"; cat $O/ab_lib_nv.txt; tail -3 $O/ab_lib_nv.err ;;
    tracenv) timeout 600 python tools/ab_nvpk.py --trace --shapes none > $O/trace_nvpk.txt 2> $O/trace_nvpk.err; echo "tracenv rc=$?"; cat $O/trace_nvpk.txt; tail -3 $O/trace_nvpk.err ;;
    testq)  timeout 900 python -m pytest tests -m gpu -q -k "quantize or fuzz or Quantize or quest or golden" > $O/pytest_quant.log 2>&1; echo "testq rc=$?"; tail -4 $O/pytest_quant.log ;;
    absf)   timeout 600 python tools/ab_sf_stores.py > $O/ab_sf_stores.txt 2> $O/ab_sf_stores.err; echo "absf rc=$?"; cat $O/ab_sf_stores.txt; tail -3 $O/ab_sf_stores.err ;;
    absq)   timeout 600 python tools/ab_sq_abl.py > $O/ab_sq_abl.txt 2> $O/ab_sq_abl.err; echo "absq rc=$?"; cat $O/ab_sq_abl.txt; tail -3 $O/ab_sq_abl.err ;;
    ablib)  timeout 900 python tools/ab_lib_shapes.py ${AB_OLD:-build/exp/libqamd_base.so} ${AB_NEW:-qutlass_amd/libqutlass_amd.so} > $O/ab_lib_shapes.txt 2> $O/ab_lib_shapes.err; echo "ablib rc=$?"; cat $O/ab_lib_shapes.txt; tail -3 $O/ab_lib_shapes.err ;;
    configs) timeout 900 python bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "configs rc=$?"; tail -30 $O/bench_configs.jsonl | cut -c1-240 ;;
    calibnv) timeout 900 python tools/calib_tiles.py nvf4 > $O/calib_tiles_nvf4.txt 2> $O/calib_tiles_nvf4.err; echo "calibnv rc=$?"; tail -5 $O/calib_tiles_nvf4.txt ;;
    abbwd)  timeout 600 python tools/ab_bwd.py > $O/ab_bwd.txt 2> $O/ab_bwd.err; echo "abbwd rc=$?"; cat $O/ab_bwd.txt; tail -3 $O/ab_bwd.err ;;
    fullcmp) timeout 900 python tools/full_compare.py > $O/full_compare.jsonl 2> $O/full_compare.err; echo "fullcmp rc=$?"; cat $O/full_compare.jsonl; tail -3 $O/full_compare.err ;;
    pmcstream) timeout 900 bash tools/pmc_stream_ops.sh qutlass_amd/libqutlass_amd.so $O/pmc_stream 8192 > $O/pmc_stream.log 2>&1; echo "pmcstream rc=$?"; tail -120 $O/pmc_stream.log ;;
    hbm)    timeout 600 python tools/hbm_ceilings.py > $O/hbm_ceilings.txt 2> $O/hbm_ceilings.err; echo "hbm rc=$?"; cat $O/hbm_ceilings.txt; tail -3 $O/hbm_ceilings.err ;;
    abtr)   timeout 600 python tools/ab_transpose.py > $O/ab_transpose.txt 2> $O/ab_transpose.err; echo "abtr rc=$?"; cat $O/ab_transpose.txt; tail -3 $O/ab_transpose.err ;;
    ablbwd) timeout 600 python tools/ab_bwd_abl.py > $O/ab_bwd_abl.txt 2> $O/ab_bwd_abl.err; echo "ablbwd rc=$?"; cat $O/ab_bwd_abl.txt; tail -3 $O/ab_bwd_abl.err ;;
    testbwd) timeout 600 python -m pytest tests -m gpu -q -k "backward or quartet or bwd or transpos or square" > $O/pytest_bwd.log 2>&1; echo "testbwd rc=$?"; tail -4 $O/pytest_bwd.log ;;
    contention) timeout 300 python tools/final_stage_contention.py > $O/final_stage_contention.txt 2> $O/final_stage_contention.err; echo "contention rc=$?"; cat $O/final_stage_contention.txt; tail -3 $O/final_stage_contention.err ;;
    abmx)   timeout 900 python tools/ab_mxsk.py > $O/ab_mxsk.txt 2> $O/ab_mxsk.err; echo "abmx rc=$?"; cat $O/ab_mxsk.txt; tail -3 $O/ab_mxsk.err ;;
    hazard) (cd tests/native && hipcc --offload-arch=gfx950 -O2 store_hazard_probe.hip -o store_hazard_probe 2>/dev/null); timeout 120 tests/native/store_hazard_probe > $O/store_hazard_probe.txt 2>&1; echo "hazard rc=$?"; cat $O/store_hazard_probe.txt ;;
    xposeub) (cd tests/native && hipcc --offload-arch=gfx950 -O3 -w xpose_traffic_ubench.hip -o xpose_traffic_ubench); timeout 300 tests/native/xpose_traffic_ubench > $O/xpose_traffic_ubench.txt 2>&1; echo "xposeub rc=$?"; cat $O/xpose_traffic_ubench.txt ;;
    qtpanel) timeout 600 python tools/check_qt_panel.py > $O/check_qt_panel.txt 2>&1; echo "qtpanel rc=$?"; tail -15 $O/check_qt_panel.txt ;;
    powerdata) timeout 600 python tools/power_data_probe.py > $O/power_data_probe.txt 2> $O/power_data_probe.err; echo "powerdata rc=$?"; cat $O/power_data_probe.txt; tail -3 $O/power_data_probe.err ;;
    abvar) timeout 600 python tools/ab_gemm_variants.py > $O/ab_gemm_variants.txt 2> $O/ab_gemm_variants.err; echo "abvar rc=$?"; cat $O/ab_gemm_variants.txt; tail -3 $O/ab_gemm_variants.err ;;
    libdiff) # [r5] where do two builds disagree: LD_PAIRS="old.so:new.so:fmt ..." (tools/lib_diff.py)
            for pr in ${LD_PAIRS:-build/exp/libqamd_base.so:qutlass_amd/libqutlass_amd.so:mxf4}; do IFS=: read a b f <<< "$pr"
              echo "== $a vs $b ($f)" >> $O/lib_diff.txt; timeout 300 python tools/lib_diff.py $a $b --fmt=$f >> $O/lib_diff.txt 2>> $O/lib_diff.err; done
            echo "libdiff rc=$?"; cat $O/lib_diff.txt; tail -3 $O/lib_diff.err ;;
    abpairs) # [r5] timing of pairs of builds on chosen shapes: AB_PAIRS="old.so:new.so:fmt:shapes ..." (tools/ab_lib_shapes.py)
            for pr in $AB_PAIRS; do IFS=: read a b f sh <<< "$pr"
              echo "== $a vs $b" >> $O/ab_pairs.txt; timeout 600 python tools/ab_lib_shapes.py $a $b --fmt=$f --shapes=$sh >> $O/ab_pairs.txt 2>> $O/ab_pairs.err; done
            echo "abpairs rc=$?"; cat $O/ab_pairs.txt; tail -3 $O/ab_pairs.err ;;
    pmcgemm) # [r5] SQ / TCC / GRBM counter passes of ONE lab variant of the MXFP4 GEMM at 4096^3 (PMC_VARIANT, default 90 = the product's persistent kernel)
            (cd tests/native && bash build.sh > /dev/null 2>&1); timeout 900 bash tools/pmc_gemm.sh ${PMC_VARIANT:-90} $O/pmc_gemm > $O/pmc_gemm.log 2>&1; echo "pmcgemm rc=$?"; cat $O/pmc_gemm/summary.txt | cut -c1-200 | head -120 ;;
    contabl) # [r5] the stage trace under side builds of the lab library (CONT_LIBS="a.so b.so ..."; tools/build_variant.py --lab -DQAMD_FS_ABL=...)
            for l in $CONT_LIBS; do echo "== $l" >> $O/contention_abl.txt; QAMD_LAB_LIB=$l timeout 300 python tools/final_stage_contention.py 2>> $O/contention_abl.err | grep -A2 "^grid" | grep -v "^--" >> $O/contention_abl.txt; done
            echo "contabl rc=$?"; cat $O/contention_abl.txt; tail -3 $O/contention_abl.err ;;
    *) echo "unknown step $step" ;;
  esac
  echo "[$step: $(( $(date +%s) - t0 )) s]"
done
find $O -name "*.db" -size +8M -delete
