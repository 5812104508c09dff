#!/bin/bash
# Round-3 GPU session Z: the cost-model tile rules (nvf4_auto_cfg, the lowered big-tile threshold and the wider 256x128 rule of the MX GEMMs) --
# full GPU suite, smoke, the dip scan and the forced-variant calibration after the change, bench.py.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3z; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python tools/dip_scan.py > $O/dip_scan_after.txt 2> $O/dip_scan.err; echo "dip rc=$?"; tail -3 $O/dip_scan_after.txt
timeout 600 python tools/calib_tiles.py > $O/calib_tiles_after.txt 2> $O/calib.err; echo "calib rc=$?"; wc -l $O/calib_tiles_after.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3z/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'kernel_us', r['kernel_us'])
PY
