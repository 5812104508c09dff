#!/bin/bash
# Round-2 GPU session P: full GPU suite + smoke + bench + bench_configs on the current build.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2p; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['per_launch_us']['median'], d['config']['parity_vs_cpu_oracle_slab'], d['cpu_baseline']['value'])"
timeout 900 python bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; echo "bench_configs rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r2p/bench_configs.jsonl'):
    d=json.loads(l); r=d.get('roofline',{})
    if d['config'].startswith('C') : print(f"{d['config'][:80]:80s} {d['us']:9.2f} us frac={r.get('frac','')}")
PY
