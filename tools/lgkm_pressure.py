#!/usr/bin/env python3
"""Static upper bound of the LDS / scalar-memory operations a wave can have outstanding, per basic block of a kernel (hipcc -S; CPU only): every ds_* / s_load / s_buffer_load
counts + 1, an s_waitcnt lgkmcnt(n) clamps the count to n, a block starts from 0 (so the figure is a lower bound of the true worst case across block boundaries).

Why it exists: the lgkmcnt field of gfx9-family parts has 4 bits (15); LLVM clamps the waits it inserts to lgkmcnt(14) when more are pending and relies on the hardware to
hold back the 16th operation.  When the one schedule of round 4 that produced wrong results with a correct-looking ISA (QAMD_DEEPP_RB2 = 1) turned out to reach 24 by this
count, overflow was the first suspect -- but the validated product kernels reach 22 (the K loop of the persistent MXFP4 kernel) and 28 (256 x 128 three-stage ring) by the
same count and are bit-exact over hundreds of tests, so a high count alone is not the explanation.  Kept as a measuring stick for that investigation.

    python tools/lgkm_pressure.py <translation unit> <substring of the mangled kernel name> [--lab] [-DFLAG ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qutlass_amd.build import TU_FLAGS  # noqa: E402


def pressure(tu, pat, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DQAMD_TU={tu}", "--cuda-device-only", "-S", "-o", out,
               os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")] + list(extra) + TU_FLAGS.get(tu, [])
        subprocess.run(cmd, check=True, capture_output=True)
        lines = open(out).read().split("\n")
    res = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w*" + re.escape(pat) + r"\w*):", l)
        if not m:
            continue
        label, c, worst = "entry", 0, {}
        for x in lines[i + 1:]:
            t = x.strip().split(";")[0].strip()
            if t.startswith("s_endpgm"):
                break
            if t.startswith(".LBB"):
                label, c = t.split(":")[0], 0
                continue
            if not t or t.startswith("."):
                continue
            op = t.split()[0]
            if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                c += 1
                worst[label] = max(worst.get(label, 0), c)
            w = re.search(r"lgkmcnt\((\d+)\)", t)
            if op == "s_waitcnt" and w:
                c = min(c, int(w.group(1)))
        res[m.group(1)] = worst
    return res


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("-")]
    extra = [x for x in sys.argv[1:] if x.startswith("-D")] + (["-DQAMD_BENCH=1"] if "--lab" in sys.argv else [])
    for name, worst in pressure(int(a[0]), a[1], extra).items():
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
        print(name[:110], "max", max(worst.values()) if worst else 0, " worst blocks:", top)
