#!/usr/bin/env python3
"""How much code a persistent GEMM kernel runs ahead of its first MFMA, read off the ISA (hipcc -S of one translation unit; CPU only).

Round 4 measured that these instructions are exposed one for one at 4096^3 (the operands are hot in the memory-side cache, so the first stage lands quickly and
nothing hides behind it: DESIGN.md 7): 624 -> 555 instructions were worth 1.1 %.  tests/test_cabi_and_host.py pins the counts so that a change elsewhere does not
quietly put them back.

    python tools/isa_prologue.py <translation unit> <substring of the mangled kernel name> [--lab]
prints {first_dma, waits_before_dma, barrier, first_mfma, arch_vgprs, sgpr_spills, vgpr_spills} -- positions are instruction counts in listing order from the
kernel's entry (labels and directives not counted); for the product kernels the listing order up to the first MFMA is the executed order of a first tile."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qutlass_amd.build import TU_FLAGS  # noqa: E402


def prologue(tu, pat, lab=False):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DQAMD_TU={tu}", "--cuda-device-only", "-S", "-o", out,
               os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")] + (["-DQAMD_BENCH=1"] if lab else []) + TU_FLAGS.get(tu, [])
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read()
    lines = text.split("\n")
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l)), None)
    if start is None:
        raise SystemExit(f"no kernel label matching {pat!r} in translation unit {tu}")
    name = lines[start].split(":")[0]
    body = [l.strip().split(";")[0].strip() for l in lines[start + 1:]]
    body = [l for l in body if l and not l.startswith(".")]
    first_dma = next(i for i, l in enumerate(body) if l.startswith("buffer_load") and l.endswith(" lds"))
    first_mfma = next(i for i, l in enumerate(body) if l.startswith("v_mfma"))
    barrier = next(i for i, l in enumerate(body) if l.startswith("s_barrier"))
    entry = next(e for e in text[text.index("amdhsa.kernels:"):].split("\n  - ") if re.search(r"\.name:\s+" + re.escape(name) + r"\n", e))
    def field(k):
        return int(re.search(r"\." + k + r":\s+(\d+)", entry).group(1))
    return {"kernel": name, "first_dma": first_dma, "waits_before_dma": sum(1 for l in body[:first_dma] if l.startswith("s_waitcnt")), "barrier": barrier,
            "first_mfma": first_mfma, "arch_vgprs": field("vgpr_count") - field("agpr_count"), "sgpr_spills": field("sgpr_spill_count"),
            "vgpr_spills": field("vgpr_spill_count")}


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    print(json.dumps(prologue(int(a[0]), a[1], "--lab" in sys.argv), indent=1))
