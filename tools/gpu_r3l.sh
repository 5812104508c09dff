#!/bin/bash
# Round-3 GPU session L: what sits between two launches of the headline kernel -- in-stream vs HIP-graph replays, kernel arguments in host vs device memory
# (HIP_FORCE_DEV_KERNARG), tools/ab_launch_floor.py; then bench.py under the better setting.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3l; mkdir -p $O
for v in unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  timeout 300 python tools/ab_launch_floor.py >> $O/ab_launch_floor.jsonl 2>> $O/ab_launch_floor.err; echo "ab $v rc=$?"
done
unset HIP_FORCE_DEV_KERNARG
cat $O/ab_launch_floor.jsonl
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 600 python bench.py --no-configs --no-pmc --no-cpu-baseline > $O/bench_kernarg$v.json 2>> $O/bench.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.loads(open('$O/bench_kernarg$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('kernarg $v: value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_us', r['kernel_us'], 'per_launch', r['per_launch_us']['median'], d.get('power',{}).get('timed_region'))
PY
done
