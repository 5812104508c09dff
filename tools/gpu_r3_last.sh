#!/bin/bash
# Round-3 last GPU session: the GPU suite, smoke, bench.py and the GPU-only plan checks on the final commit.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3_last; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_last/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'kernel_us', r['kernel_us'], 'cpu', d['cpu_baseline']['value'])
for k,v in (d.get('configs') or {}).items(): print(k, v.get('us'), v.get('roofline',{}).get('frac'))
PY
CALIB_MS=1,8,16,32 timeout 300 python tools/calib_mx_small.py mxf4 > $O/calib_mx_decode_graph_after.txt 2> $O/mx.err; echo "mx rc=$?"; grep -E "^mxf4 +(16|32) +(4096|5120|6144|8192) +(4096|5120|8192) " $O/calib_mx_decode_graph_after.txt | cut -c1-60; grep -E "^mxf4 +(1|8) +8192 +8192 " $O/calib_mx_decode_graph_after.txt | cut -c1-60
