#!/bin/bash
# HBM-side traffic of the bench's dominant kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one TCC pass; MI355X_MICROARCH.md "rocprofv3 PMC slots") over the SAME command
# bench.py times, plus a --kernel-trace --stats pass.  Run on the GPU box:
#     tools/pmc_bench.sh [outdir]        -> <outdir>/summary.txt, <outdir>/traffic.json
# traffic.json is what bench.py reports as roofline.traffic (copy it to profiles/pmc_bench_<round>.json); the last step
# re-runs bench.py un-profiled with these fresh counters attached (traffic_stale: false).
OUT=${1:-gpurun_out/pmc_bench}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-configs --no-pmc"   # the headline hot loop of the default command (durations must agree with the bench line);
# [r3] the side configs and the nested rocprofv3 --pmc child runs of a plain `bench.py` stay out of the profiled command
rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o p -- $CMD > $R/$OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/fetch -o p -- $CMD > $R/$OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/write -o p -- $CMD > $R/$OUT/write.log 2>&1
cd $R
python tools/rocprof_summary.py $OUT/*/p_results.db > $OUT/summary.txt 2>&1
python tools/rocprof_summary.py --traffic-json gemm_mx_ $OUT/fetch/p_results.db $OUT/write/p_results.db > $OUT/traffic.json
cat $OUT/traffic.json
# final, un-profiled run of the same command on the same box: its JSON line carries THIS session's counters (traffic_stale: false)
QAMD_PMC_TRAFFIC_JSON=$R/$OUT/traffic.json python $R/bench.py --no-configs > $R/$OUT/bench_with_fresh_traffic.json 2> $R/$OUT/bench_with_fresh_traffic.err
tail -1 $R/$OUT/bench_with_fresh_traffic.json
