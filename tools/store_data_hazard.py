#!/usr/bin/env python3
"""Scan the ISA of the product kernels for the one pattern that (by the evidence of round 4) the compiler does not guard on gfx950:

    buffer_store_dwordx3/x4 v[a:b], voff, rsrc, sN offen        <- soffset in an SGPR
    v_... v[a..b] ...                                           <- a VALU write of the store's DATA registers one instruction later

LLVM's hazard recogniser inserts wait states between a > 64-bit VMEM store and a VALU write of its data registers (2 on gfx940-family parts) -- but only when the
store's soffset is not a register (GCNHazardRecognizer::createsVALUHazard).  [r5] SETTLED ON THE DEVICE (tests/native/store_hazard_probe.hip, profiles/store_hazard_probe_r5.txt,
2.1e9 stored words per case): with an SGPR soffset a VALU write DIRECTLY behind the store corrupts it (v_mov_b32: 3.3e6 wrong words, v_pk_mul_f32: 1.2e8), ONE wait state
(any instruction in between) is enough; without an SGPR soffset two are needed (what the compiler inserts); an LDS read returning into the data registers is harmless.
The round-4 QAMD_DEEPP_RB2 = 1 variant had exactly this pattern on one store per pair, and its wrong outputs are exactly that store's elements (tools/lib_diff.py,
profiles/lib_diff_r5a_rb2_and_bf16_first.txt: always pair 6, pass 1, elements 2-3 of the lane's 16 bytes).  So: distance 1 (directly behind) is a bug, distance >= 2 is safe.

    python tools/store_data_hazard.py [--lab]       # exit status 1 if any kernel of the build has a distance-1 case; prints the closest case per kernel otherwise
CPU only (hipcc -S of every translation unit of the build, in parallel)."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qutlass_amd.build import TU_FLAGS, UNITS, UNITS_BENCH  # noqa: E402

REG = re.compile(r"\bv(?:(\d+)|\[(\d+):(\d+)\])")


def vregs(tok):
    out = []
    for m in REG.finditer(tok):
        out += [int(m.group(1))] if m.group(1) is not None else list(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan_unit(tu, extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DQAMD_TU={tu}", "--cuda-device-only", "-S", "-o", out,
               os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")] + list(extra) + TU_FLAGS.get(tu, [])
        subprocess.run(cmd, check=True, capture_output=True)
        lines = open(out).read().split("\n")
    res, name, body = {}, None, []
    def flush():
        if name is None:
            return
        best = None
        for i, x in enumerate(body):
            m = re.match(r"buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0)\b", x)
            if not m:
                continue
            data = set(vregs(m.group(1)))
            for j in range(i + 1, min(i + 8, len(body))):
                y = body[j]
                op = y.split()[0]
                if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_readlane", "v_readfirstlane")):
                    dst = y.split(None, 1)[1].split(",")[0]
                    if set(vregs(dst)) & data:
                        if best is None or j - i < best[0]:
                            best = (j - i, x[:70], y[:50])
                        break
        res[name] = best
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        t = l.strip().split(";")[0].strip()
        if m:
            flush(); name, body = m.group(1), []
        elif t.startswith("s_endpgm"):
            flush(); name = None
        elif name and t and not t.startswith("."):
            body.append(t)
    return res


def main():
    lab = "--lab" in sys.argv
    extra = ["-DQAMD_BENCH=1"] if lab else []
    extra += [a for a in sys.argv[1:] if a.startswith("-D")]
    units = UNITS_BENCH if lab else UNITS
    with ThreadPoolExecutor(max_workers=len(units)) as ex:
        parts = list(ex.map(lambda t: scan_unit(t, extra), units))
    bad = 0
    for p in parts:
        for k, v in p.items():
            if v is not None:
                print(f"distance {v[0]}: {k[:90]}\n      {v[1]}\n      {v[2]}")
                bad += v[0] <= 1
    print(f"{sum(len(p) for p in parts)} kernels scanned; {bad} with a VALU write of store data directly behind a scalar-offset store")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
