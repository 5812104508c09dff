#!/bin/bash
# Round-3 GPU session C: full GPU suite after the fixes + the decode-path / blocked-quantizer A/B again.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_gpu.log; grep QUEST_BINADE $O/pytest_gpu.log; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
timeout 600 python tools/ab_blocked_quant.py > $O/ab_blocked_quant.txt 2> $O/ab_blocked_quant.err; echo "ab rc=$?"; cat $O/ab_blocked_quant.txt; tail -3 $O/ab_blocked_quant.err
