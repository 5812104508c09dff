#!/bin/bash
# Round-2 GPU session C: where the ~10 us outside the K loop go -- store patterns / policies, workgroup entry-exit skew, launch floor.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r2c; mkdir -p $O
timeout 120 tests/native/qamd_check stores > $O/stores.log 2>&1; echo "stores rc=$?"
grep UBENCH $O/stores.log
timeout 200 tests/native/qamd_check staux > $O/staux.log 2>&1; echo "staux rc=$?"
grep BENCH $O/staux.log
timeout 100 tests/native/qamd_check deepptrace > $O/deepp_trace.log 2>&1; echo "deepptrace rc=$?"
grep -v "^DEVICE" $O/deepp_trace.log
for kv in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "== env: $kv"; env $kv timeout 300 python bench.py --no-cpu-baseline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['per_launch_us'])"
done
