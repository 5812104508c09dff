#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) result database as text: per-kernel calls / total / average
duration (the `--kernel-trace --stats` view) and, when the run collected PMC counters, their per-dispatch averages.

    python tools/rocprof_summary.py gpurun_out/prof_r1/bench_results.db [more.db ...] > profiles/xxx.txt
"""
import sqlite3
import sys


def short(name: str, n: int = 110) -> str:
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    for path in sys.argv[1:]:
        c = sqlite3.connect(path)
        print(f"== {path}")
        rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
            for name, calls, tot, avg, pct in rows:
                print(f"{calls:6d} {tot:12.3f} {avg:10.3f} {pct:6.2f}  {short(name)}")
        try:
            q = ("select k.name, p.name, count(*), avg(e.value) from pmc_events e "
                 "join pmc_info p on e.pmc_id = p.id join kernels k on k.dispatch_id = e.event_id group by k.name, p.name")
            pmc = list(c.execute(q))
        except sqlite3.Error:
            pmc = []
        if not pmc:
            try:
                cur = c.execute("select * from counters_collection limit 1")
                cols = [d[0] for d in cur.description]
                kn = "kernel_name" if "kernel_name" in cols else "name"
                q = f"select {kn}, counter_name, count(*), avg(value) from counters_collection group by {kn}, counter_name"
                pmc = list(c.execute(q))
            except sqlite3.Error:
                pmc = []
        if pmc:
            print(f"{'dispatches':>10} {'avg_value':>18}  counter  kernel")
            for kname, cname, n, avg in pmc:
                print(f"{n:10d} {avg:18.1f}  {cname}  {short(kname, 80)}")
        print()


if __name__ == "__main__":
    main()
