#!/bin/bash
# Round-3 GPU session A: full GPU suite (incl. tests/test_gpu_round3.py), heterogeneous-launch parity + steady-state A/B, bench.py with
# power sampling / fresh PMC traffic / side configs.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest_gpu.log; grep QUEST_BINADE $O/pytest_gpu.log
timeout 300 tests/native/qamd_check hetero > $O/native_hetero.log 2>&1; echo "hetero rc=$?"; grep -c "^PASS\|^ok\|PASS" $O/native_hetero.log; grep -i "fail" $O/native_hetero.log | head -20
QAMD_STEADY_MS=30 timeout 600 tests/native/qamd_check heterobench > $O/native_heterobench.log 2>&1; echo "heterobench rc=$?"
grep BENCH $O/native_heterobench.log | awk '{printf "%-55s %s us %s TF\n", $2" "$3" "$4, $(NF-3), $(NF-1)}'
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3a/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'median', r['per_launch_us']['median'], 'parity', d['config'].get('parity_vs_cpu_oracle_slab'))
print('power', d.get('power'))
print('traffic', r.get('traffic'), r.get('traffic_source'), r.get('traffic_over_algorithmic'))
for k,v in (d.get('configs') or {}).items(): print(k, v if 'us' not in v else (v['us'], v['roofline']['frac']))
PY
