#!/bin/bash
# Round-3 GPU session V: the deep kernel with one fragment read behind each MFMA -- native parity (16 shapes vs the oracle), GPU test subset, bench.py, and the
# old library's timing on the same box (tools/ab_lib_gemm.py)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r3v}; mkdir -p $O
timeout 600 tests/native/qamd_check deepp > $O/native_deepp.log 2>&1; echo "deepp rc=$?"; grep -c "OK\|PASS\|exact" $O/native_deepp.log; grep -i "FAIL\|mismatch" $O/native_deepp.log | head -5; tail -3 $O/native_deepp.log
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 600 python bench.py --no-pmc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", r["frac"], "kernel_us", r["kernel_us"], d.get("power",{}).get("timed_region"), "parity", d["config"].get("parity_vs_cpu_oracle_slab"))
for k,v in (d.get("configs") or {}).items(): print(k, v.get("us"), v.get("roofline",{}).get("frac"), v.get("power_w"), v.get("sclk_mhz"))
PY
timeout 600 python tools/ab_lib_gemm.py build/ab/libqutlass_amd_old.so qutlass_amd/libqutlass_amd.so > $O/ab_lib_gemm.txt 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $O/ab_lib_gemm.txt | tail -30
