#!/bin/bash
# Round-3 GPU session E: 4-deep ring A/B for mid-size outputs, decode A/B with K = 8192, bench_configs with the 8192^2 rows, full GPU suite.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
QAMD_STEADY_MS=30 timeout 600 tests/native/qamd_check ring4 > $O/native_ring4.log 2>&1; echo "ring4 rc=$?"
grep -E "BENCH|CHECK" $O/native_ring4.log | awk '/BENCH/ {printf "%-45s %s us %s TF\n", $2" "$3" "$4" "$5, $(NF-3), $(NF-1)} /CHECK/ {print}'
timeout 600 python tools/ab_blocked_quant.py > $O/ab_blocked_quant.txt 2> $O/ab_blocked_quant.err; echo "ab rc=$?"; sed -n '/linear layer/,$p' $O/ab_blocked_quant.txt
timeout 900 python bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "bench_configs rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r3e/bench_configs.jsonl'):
    d=json.loads(l); r=d.get('roofline',{})
    if '8192' in d['config'] or d['config'].startswith('C') or 'backward' in d['config'] or 'transpose' in d['config']:
        print(f"{d['config'][:100]:100s} {d['us']:9.2f} us frac={r.get('frac','')}")
PY
