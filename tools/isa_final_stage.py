#!/usr/bin/env python3
"""ISA of one kernel of a translation unit, cut at its MFMAs: prints the instructions between consecutive MFMAs of the block that holds the last-stage retirement
(the block with the most ds_write_b64 / buffer_store_dwordx4), one line per slot, so that a hand-made schedule can be read off the compiler's output.

    python tools/isa_final_stage.py <unit> <kernel-name substring> [extra hipcc flags ...]      (CPU only: hipcc -S)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qutlass_amd.build import TU_FLAGS  # noqa: E402

SHORT = {"v_accvgpr_read_b32": "ar", "v_mul_f32_e32": "mul", "v_mul_f32": "mul", "v_cvt_pk_bf16_f32": "cvt", "ds_write_b64": "W64", "ds_write_b128": "W128", "ds_read_b128": "R128",
         "ds_read_b64": "R64", "ds_read2_b64": "R2x64", "buffer_store_dwordx4": "ST", "buffer_load_dwordx4": "DMA", "s_mov_b32": "smov", "v_add_u32_e32": "vadd", "v_pk_mul_f32": "pkmul",
         "v_cndmask_b32_e32": "cnd", "v_cmp_lt_i32_e32": "cmp", "s_waitcnt": "WAIT", "s_nop": "nop", "ds_read_b32": "R32", "v_mov_b32_e32": "vmov"}


def kernel_body(tu, pat, extra):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DQAMD_TU={tu}", "--cuda-device-only", "-S", "-o", out,
               os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")] + extra + TU_FLAGS.get(tu, [])
        subprocess.run(cmd, check=True, capture_output=True)
        lines = open(out).read().split("\n")
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l)), None)
    if start is None:
        sys.exit(f"no kernel label matching {pat!r} in unit {tu}")
    body = []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if t and not t.startswith(";") and not (t.startswith(".") and not t.startswith(".LBB")):
            body.append(t)
    return lines[start].split(":")[0], body


def main():
    tu, pat, extra = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
    name, body = kernel_body(tu, pat, extra)
    print(name, len(body), "instructions")
    # basic blocks
    blocks, cur = [], []
    for t in body:
        if t.startswith(".LBB"):
            blocks.append(cur); cur = []
        else:
            cur.append(t)
    blocks.append(cur)
    score = lambda b: sum(t.startswith(("ds_write_b64", "buffer_store_dwordx4")) for t in b)
    for bi, b in enumerate(blocks):
        if score(b) < 16:
            continue
        print(f"--- block {bi}: {len(b)} instructions, {sum(t.startswith('v_mfma') for t in b)} MFMAs, {score(b)} LDS-write / store instructions")
        slot, n, tot = [], 0, 0
        for t in b + ["v_mfma_END"]:
            op = t.split()[0]
            if op.startswith("v_mfma"):
                issue = sum(1 + (int(x.split()[1]) if x.split()[0] == "s_nop" else 0) for x in slot)
                txt = " ".join(SHORT.get(x.split()[0], x.split()[0]) + (("(" + x.split(None, 1)[1].replace(" ", "") + ")") if x.split()[0] == "s_waitcnt" else "") for x in slot)
                print(f"  slot {n:3d} [{issue:2d}] {txt[:230]}")
                tot += max(8, issue + 1)
                slot, n = [], n + 1
            else:
                slot.append(t)
        print(f"  issue-slot model (4 cycles per instruction, MFMA = 8 slots): ~{4 * tot} cycles")


if __name__ == "__main__":
    main()
