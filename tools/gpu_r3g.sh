#!/bin/bash
# Round-3 GPU session G: validation of the final build -- full GPU suite, smoke, bench.py (driver protocol), rocprofv3 kernel-trace + PMC passes of the same command,
# the batch sweep in the shape of the reference's benchmark script with the fused provider.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${QAMD_SESSION:-r3g}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
import os; d=json.loads(open('gpurun_out/%s/bench.json' % os.environ.get('QAMD_SESSION', 'r3g')).read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'kernel_us', r['kernel_us'], 'parity', d['config'].get('parity_vs_cpu_oracle_slab'), 'cpu', d['cpu_baseline']['value'])
print('power', d.get('power', {}).get('timed_region'))
print('traffic', r.get('traffic'), r.get('traffic_over_algorithmic'), r.get('traffic_source'))
for k,v in (d.get('configs') or {}).items(): print(k, v if 'us' not in v else (v['us'], v['roofline']['frac']))
PY
bash tools/pmc_bench.sh gpurun_out/${QAMD_SESSION:-r3g}/pmc_bench > $O/pmc_bench.log 2>&1; echo "pmc rc=$?"; head -30 $O/pmc_bench/summary.txt; cat $O/pmc_bench/traffic.json
timeout 900 python benchmarks/bench_mxfp4_mi355x.py --model Llama-3-8B --fused --max-batch 8192 --reps 30 > $O/bench_sweep_mxfp4_llama3_8b.txt 2> $O/bench_sweep.err; echo "sweep rc=$?"; cat $O/bench_sweep_mxfp4_llama3_8b.txt
timeout 600 python tools/dip_scan.py > $O/dip_scan.txt 2> $O/dip_scan.err; echo "dip rc=$?"; tail -1 $O/dip_scan.txt
find $O -name "*.db" -size +8M -delete
