#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of the PRODUCT library, from clang's -Rpass-analysis=kernel-resource-usage remarks
(device pass only, one run per translation unit of csrc/capi.hip, in parallel).  Exit status 1 if any kernel uses scratch memory:
a spill in one of the hand-scheduled kernels is silent and slow (a scratch_load forces s_waitcnt vmcnt(0)), so the build
(`__graft_entry__.build()`) and tests/test_cabi_and_host.py run this as a gate.

    python tools/kernel_resources.py [--lab] [--all]      # --lab: the -DQAMD_BENCH=1 build; --all: print every kernel
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")
sys.path.insert(0, ROOT)
from qutlass_amd.build import TU_FLAGS, UNITS, UNITS_BENCH  # noqa: E402  (translation units and per-unit compiler flags of the build)
FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
          "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}


def demangle(names):
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def unit(tu: int, lab: bool):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"-DQAMD_TU={tu}", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", os.devnull] + (["-DQAMD_BENCH=1"] if lab else []) + TU_FLAGS.get(tu, [])
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    kernels, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            v = m.group(2)
            cur[FIELDS[m.group(1).strip()]] = int(v) if v.isdigit() else v
    return kernels


def collect(lab: bool = False):
    units = UNITS_BENCH if lab else UNITS          # the translation units of the build itself (qutlass_amd/build.py): unit 8 = the persistent NVFP4 kernel
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as ex:
        parts = list(ex.map(lambda t: unit(t, lab), units))
    allk = {}
    for p in parts:
        allk.update(p)
    return allk


def main():
    lab, show_all = "--lab" in sys.argv, "--all" in sys.argv
    ks = collect(lab)
    names = demangle(list(ks))
    bad = 0
    for mangled, r in sorted(ks.items(), key=lambda kv: names[kv[0]]):
        spilled = r.get("scratch", 0) != 0 or r.get("vgpr_spill", 0) != 0
        bad += spilled
        if show_all or spilled:
            nm = names[mangled]
            print(f"{'SCRATCH ' if spilled else '        '}{nm[:150]:150s} vgpr {r.get('vgpr'):>3} agpr {r.get('agpr'):>3} sgpr {r.get('sgpr'):>3} "
                  f"(spilled {r.get('sgpr_spill')}) scratch {r.get('scratch')} lds {r.get('lds')} occ {r.get('occ')}")
    print(f"{len(ks)} kernels, {bad} with scratch / VGPR spills ({'lab' if lab else 'product'} build)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
