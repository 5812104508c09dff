#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) result database as text: per-kernel calls / total / average
duration (the `--kernel-trace --stats` view) and, when the run collected PMC counters, their per-dispatch averages.

    python tools/rocprof_summary.py gpurun_out/prof_r1/bench_results.db [more.db ...] > profiles/xxx.txt
"""
import os
import sqlite3
import sys


def short(name: str, n: int = 110) -> str:
    return name if len(name) <= n else name[: n - 3] + "..."


def pmc_rows(c):
    try:
        q = ("select k.name, p.name, count(*), avg(e.value) from pmc_events e "
             "join pmc_info p on e.pmc_id = p.id join kernels k on k.dispatch_id = e.event_id group by k.name, p.name")
        rows = list(c.execute(q))
    except sqlite3.Error:
        rows = []
    if not rows:
        try:
            cur = c.execute("select * from counters_collection limit 1")
            cols = [d[0] for d in cur.description]
            kn = "kernel_name" if "kernel_name" in cols else "name"
            q = f"select {kn}, counter_name, count(*), avg(value) from counters_collection group by {kn}, counter_name"
            rows = list(c.execute(q))
        except sqlite3.Error:
            rows = []
    return rows


def traffic_json(kernel_substr, paths):
    """FETCH_SIZE / WRITE_SIZE (KB per dispatch) of the kernel whose name contains `kernel_substr`, corrected as
    MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE tallies 128-B read requests at 64 B -> x2 for wide
    coalesced reads (our 16 B/lane buffer_load...lds); WRITE_SIZE taken as is (it reproduces the 32 MiB output
    of the 4096^3 GEMM exactly, i.e. it is calibrated for our full-line bf16 stores)."""
    import json

    vals = {}
    for path in paths:
        for kname, cname, n, avg in pmc_rows(sqlite3.connect(path)):
            if kernel_substr in kname and cname in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[cname] = (avg, n, kname)
    out = {"kernel": None, "unit": "bytes per launch"}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        f, w = vals["FETCH_SIZE"][0] * 1024.0, vals["WRITE_SIZE"][0] * 1024.0
        out.update({"kernel": vals["FETCH_SIZE"][2], "dispatches": vals["FETCH_SIZE"][1],
                    "fetch_size_raw_bytes": f, "fetch_bytes_corrected_x2": 2 * f, "write_bytes": w,
                    "traffic_bytes": 2 * f + w})
    print(json.dumps(out))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--traffic-json":
        return traffic_json(sys.argv[2], sys.argv[3:])
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help"):
        print(__doc__ or "usage: rocprof_summary.py [--traffic-json <kernel substring>] <rocprofv3 results .db> ...")
        return
    for path in sys.argv[1:]:
        if not os.path.exists(path):   # sqlite3.connect would silently create an empty database
            print(f"== {path}: no such file")
            continue
        c = sqlite3.connect(path)
        print(f"== {path}")
        rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
            for name, calls, tot, avg, pct in rows:
                print(f"{calls:6d} {tot:12.3f} {avg:10.3f} {pct:6.2f}  {short(name)}")
        pmc = pmc_rows(c)
        if pmc:
            print(f"{'dispatches':>10} {'avg_value':>18}  counter  kernel")
            for kname, cname, n, avg in pmc:
                print(f"{n:10d} {avg:18.1f}  {cname}  {short(kname, 80)}")
        print()


if __name__ == "__main__":
    main()
