#!/bin/bash
# Sample rocm-smi (power / sclk) while a command runs on the GPU box: evidence for the power-limited clock.
#     tools/power_trace.sh <outfile> -- <command...>
OUT=$1; shift; shift
( while true; do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | head -c 1500; echo; sleep 0.2; done ) > $OUT.samples &
SP=$!
"$@" > $OUT.cmd.log 2>&1
kill $SP 2>/dev/null
python3 - "$OUT" <<'PY'
import json, sys, re
out = sys.argv[1]
pw, sclk = [], []
for line in open(out + ".samples"):
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    c = d.get("card0", {})
    for k, v in c.items():
        kl = k.lower()
        if "power" in kl and "(w)" in kl:
            try: pw.append(float(v))
            except Exception: pass
        if kl.startswith("sclk clock speed"):
            m = re.search(r"(\d+)\s*mhz", str(v).lower())
            if m: sclk.append(int(m.group(1)))
print(f"samples={len(pw)} power W: min={min(pw) if pw else None} max={max(pw) if pw else None} mean={sum(pw)/len(pw) if pw else None}")
print(f"sclk MHz: min={min(sclk) if sclk else None} max={max(sclk) if sclk else None} mean={sum(sclk)/len(sclk) if sclk else None}")
PY
