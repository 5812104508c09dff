#!/usr/bin/env python3
"""Register map of a hand-scheduled persistent GEMM kernel, read off its ISA: which of the 256 architectural VGPRs (+ 256 accumulation VGPRs) hold what
inside the K loop, and which are only parked across it for the final stage / the next tile.

    python tools/register_map.py <translation unit> <substring of the mangled kernel name> [--lab]
    python tools/register_map.py 2 gemm_mx_deepp_kernelINS_7GemmCfgILi256ELi256ELi2ELi2ELi4ELb0ELi0ELi2ELi0EEELi17E

The K loop is taken to be the basic block with the most MFMAs that ends in a backward branch to itself.  Classes (a register counts once, first match):
  accumulator   destination / C operand of an MFMA                       fragment      A / B operand of an MFMA (written by ds_read_b128)
  scale         scale operand of a scaled MFMA                            LDS address   address operand of a ds_read / ds_write inside the loop
  DMA offset    vaddr of a buffer_load ... lds inside the loop            loop other    anything else the loop touches
  parked        never touched by the loop but live into it (block-level liveness over the kernel's control-flow graph): final-stage / epilogue / next-tile state
  outside only  temporaries of the other blocks                           unused
CPU only (hipcc -S).  This is the evidence behind "the kernel has no register left for a second live window" (VERDICT r3 item 5)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qutlass_amd.build import TU_FLAGS  # noqa: E402

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        k = m.group(1)
        if m.group(2) is not None:
            out.append((k, int(m.group(2))))
        else:
            out += [(k, i) for i in range(int(m.group(3)), int(m.group(4)) + 1)]
    return out


def operands(line):
    t = line.split(";")[0].strip()
    parts = t.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops, depth, cur = [], 0, ""
    for ch in parts[1]:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip()); cur = ""
        else:
            cur += ch
    ops.append(cur.strip())
    return parts[0], ops


def fmt(rs):
    rs = sorted(rs)
    out, i = [], 0
    while i < len(rs):
        j = i
        while j + 1 < len(rs) and rs[j + 1] == rs[j] + 1:
            j += 1
        out.append(f"{rs[i]}" if i == j else f"{rs[i]}-{rs[j]}")
        i = j + 1
    return " ".join(out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tu, pat = int(args[0]), args[1]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DQAMD_TU={tu}", "--cuda-device-only", "-S", "-o", out,
               os.path.join(ROOT, "qutlass_amd", "csrc", "capi.hip")] + (["-DQAMD_BENCH=1"] if "--lab" in sys.argv else []) + TU_FLAGS.get(tu, [])
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read()
    lines = text.split("\n")
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l)), None)
    if start is None:
        sys.exit(f"no kernel label matching {pat!r} in translation unit {tu}")
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # basic blocks
    blocks, cur, label = [], [], "entry"
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith(".LBB"):
            blocks.append((label, cur)); cur = []; label = t.split(":")[0]
            continue
        if t.startswith("."):
            continue
        cur.append(t)
    blocks.append((label, cur))
    def n_mfma(b):
        return sum(1 for t in b[1] if t.startswith("v_mfma"))
    loops = [i for i, b in enumerate(blocks) if any(re.match(r"s_cbranch\w*\s+" + re.escape(b[0]) + r"\b", t) for t in b[1])]
    if not loops:
        sys.exit("no self-looping block found")
    kl = max(loops, key=lambda i: n_mfma(blocks[i]))
    label, body = blocks[kl]
    cls = {}
    def put(r, c):
        cls.setdefault(r, c)
    counts = {}
    for t in body:
        op, ops = operands(t)
        counts[op] = counts.get(op, 0) + 1
        if op.startswith("v_mfma"):
            for r in regs(ops[0]) + regs(ops[3]):
                put(r, "accumulator")
            for r in regs(ops[1]) + regs(ops[2]):
                put(r, "fragment")
            for o in ops[4:6]:
                for r in regs(o.split(" op_sel")[0]):
                    put(r, "scale")
    for t in body:
        op, ops = operands(t)
        if op.startswith("ds_read") or op.startswith("ds_write"):
            ai = 1 if op.startswith("ds_read") else 0
            for r in regs(ops[ai].split(" offset")[0]):
                put(r, "LDS address")
        elif op.startswith("buffer_load") and "lds" in t:
            for r in regs(ops[0]):
                put(r, "DMA offset")
    for t in body:
        _, ops = operands(t)
        for o in ops:
            for r in regs(o):
                put(r, "loop other")
    # liveness over the kernel's control-flow graph (VGPRs / AGPRs only): what is live into the K loop and not touched by it is parked across it
    def def_use(t):
        op, ops = operands(t)
        allr = [regs(o) for o in ops]
        flat = [r for rs_ in allr for r in rs_]
        if not ops:
            return [], []
        if op.startswith(("ds_write", "buffer_store", "scratch_store", "global_store", "flat_store")) or (op.startswith("buffer_load") and " lds" in t):
            return [], flat
        if op.startswith("v_writelane"):
            return allr[0], flat
        if "mac" in op:
            return allr[0], flat
        if op.startswith(("v_", "ds_read", "buffer_load", "scratch_load", "global_load", "flat_load")):
            return allr[0], [r for rs_ in allr[1:] for r in rs_]
        return [], flat
    idx = {b[0]: i for i, b in enumerate(blocks)}
    succ, use, dfn = [], [], []
    for i, (lab, b) in enumerate(blocks):
        sc, u, d = set(), set(), set()
        for t in b:
            dd, uu = def_use(t)
            u.update(r for r in uu if r not in d)
            d.update(dd)
            mm = re.match(r"s_(?:c)?branch\w*\s+(\.LBB\w+)", t)
            if mm and mm.group(1) in idx:
                sc.add(idx[mm.group(1)])
        last = b[-1] if b else ""
        if not (last.startswith("s_branch") or last.startswith("s_endpgm")) and i + 1 < len(blocks):
            sc.add(i + 1)
        succ.append(sc); use.append(u); dfn.append(d)
    live_in = [set() for _ in blocks]
    changed = True
    while changed:
        changed = False
        for i in reversed(range(len(blocks))):
            out_ = set()
            for j in succ[i]:
                out_ |= live_in[j]
            new_in = use[i] | (out_ - dfn[i])
            if new_in != live_in[i]:
                live_in[i] = new_in; changed = True
    anywhere = set()
    for _, b in blocks:
        for t in b:
            _, ops = operands(t)
            for o in ops:
                anywhere.update(regs(o))
    m = re.search(r"\.name:\s+" + re.escape(name) + r"\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text)
    for i in range(256):
        r = ("v", i)
        if r in cls:
            continue
        if r in live_in[kl]:
            cls[r] = "parked"
        elif r in anywhere:
            cls[r] = "outside only"
        else:
            cls[r] = "unused"
    print(name)
    print(f"K loop = block {label}: {n_mfma((label, body))} MFMAs, " + ", ".join(f"{v} {k}" for k, v in sorted(counts.items(), key=lambda kv: -kv[1]) if not k.startswith("v_mfma"))[:400])
    print(f"(.vgpr_count of the kernel: {m.group(1) if m else '?'} = architectural + accumulation registers)")
    for kind in ("v", "a"):
        rs = {}
        for (k, i), c in cls.items():
            if k == kind:
                rs.setdefault(c, []).append(i)
        if not rs:
            continue
        print(f"\n{'architectural VGPRs v0-v255' if kind == 'v' else 'accumulation VGPRs a0-a255'}:")
        for c in ("accumulator", "fragment", "scale", "LDS address", "DMA offset", "loop other", "parked", "outside only", "unused"):
            if c in rs:
                print(f"  {c:13s} {len(rs[c]):3d}   {fmt(rs[c])}")


if __name__ == "__main__":
    main()
