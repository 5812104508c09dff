#!/bin/bash
# Round-3 GPU session B: full GPU suite, decode-path / blocked-quantizer A/B, full-matrix C4 compare, bench_configs, hetero spot checks.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $O/pytest_gpu.log; grep QUEST_BINADE $O/pytest_gpu.log; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
timeout 600 python tools/ab_blocked_quant.py > $O/ab_blocked_quant.txt 2> $O/ab_blocked_quant.err; echo "ab rc=$?"; cat $O/ab_blocked_quant.txt; tail -3 $O/ab_blocked_quant.err
timeout 600 python tools/full_compare_c4.py > $O/full_compare_c4.json 2> $O/full_compare_c4.err; echo "c4 full rc=$?"; cat $O/full_compare_c4.json; tail -2 $O/full_compare_c4.err
QAMD_STEADY_MS=30 timeout 300 tests/native/qamd_check heterobench2 > $O/native_heterobench2.log 2>&1; echo "heterobench2 rc=$?"
grep BENCH $O/native_heterobench2.log | awk '{printf "%-55s %s us %s TF\n", $2" "$3" "$4, $(NF-3), $(NF-1)}'
timeout 900 python bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "bench_configs rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r3b/bench_configs.jsonl'):
    d=json.loads(l); r=d.get('roofline',{})
    print(f"{d['config'][:100]:100s} {d['us']:9.2f} us frac={r.get('frac','')}")
PY
