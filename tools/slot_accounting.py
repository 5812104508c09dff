#!/usr/bin/env python3
"""Static "slot accounting" of a hand-scheduled kernel: how many other instructions sit between two consecutive MFMAs in the ISA.

With ONE wave per SIMD (the 256 x 256 GEMM tiles) a wave issues at most one instruction every 4 cycles, whatever its type, so a 32-cycle MFMA
(8 passes: fp4 32x32x64, f16 / bf16 32x32x16) covers itself + 7 more instructions; a slot with more leaves the matrix pipe idle.  This is how the
clustered LDS-DMA items of round 4 were found (17 of them behind the first 16 MFMAs after the hand-off, 21 scalar selects between two stages).

    python tools/slot_accounting.py <translation unit 1..8> <substring of the mangled kernel name> [--budget 7] [--lab]
    python tools/slot_accounting.py 2 gemm_mx_deepp_kernelINS_7GemmCfgILi256ELi256ELi2ELi2ELi4ELb0ELi0ELi2ELi0EEELi17E
    python tools/slot_accounting.py 8 gemm_nvf4_pk_kernelILb0ELb0

Prints, per basic-block run of MFMAs, the slots over budget and the excess in issue slots (x 4 = cycles).  CPU only (hipcc -S)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qutlass_amd.build import TU_FLAGS  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tu, pat = int(args[0]), args[1]
    budget = int(sys.argv[sys.argv.index("--budget") + 1]) if "--budget" in sys.argv else 7
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DQAMD_TU={tu}", "--cuda-device-only", "-S", "-o", out,
               os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")] + (["-DQAMD_BENCH=1"] if "--lab" in sys.argv else []) + TU_FLAGS.get(tu, [])
        subprocess.run(cmd, check=True, capture_output=True)
        lines = open(out).read().split("\n")
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l)), None)
    if start is None:
        sys.exit(f"no kernel label matching {pat!r} in translation unit {tu}")
    print(lines[start].split(":")[0])
    slots, cur, label, total_mfma, total_excess = [], [], "entry", 0, 0
    def flush_run():
        nonlocal slots, total_excess
        if len(slots) >= 8:
            ex = sum(max(0, n - budget) for n, _ in slots)
            total_excess += ex
            print(f"  block {label}: {len(slots)} MFMAs, {sum(n for n, _ in slots) / len(slots):.1f} other instructions per slot on average, "
                  f"{ex} issue slots over the budget of {budget} (~{4 * ex} cycles of {32 * len(slots)})")
            for i, (n, ops) in enumerate(slots):
                if n > budget:
                    print(f"      slot {i:3d}: {n:2d}  {' '.join(ops)[:180]}")
        slots = []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t.startswith(";"):
            continue
        if t.startswith("."):
            if t.startswith(".LBB"):
                flush_run(); cur = []; label = t.split(":")[0]
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            slots.append((sum(n for _, n in cur), [o for o, _ in cur])); cur = []; total_mfma += 1
        else:
            n = 1 + int(t.split()[1]) // 4 if op == "s_nop" and len(t.split()) > 1 else 1
            cur.append((op, n))
    flush_run()
    print(f"  {total_mfma} MFMAs in the kernel; {total_excess} issue slots over budget in its MFMA runs")


if __name__ == "__main__":
    main()
