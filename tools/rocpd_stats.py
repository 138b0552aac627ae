#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max microseconds, share) of a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db on this ROCm): the table rocprofv3 --stats would print,
written as CSV so that it can be kept under profiles/.   usage: rocpd_stats.py DB [OUT.csv] [--skip-first-ms MS]"""
import csv
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        name = re.sub(r"\[clone .*\]", "", name).strip()
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*\)$", "", name)
        name = name.replace("seld::", "")
        d = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        dt = (e - s) / 1e3
        d[0] += 1
        d[1] += dt
        d[2] = min(d[2], dt)
        d[3] = max(d[3], dt)
    tot = sum(v[1] for v in agg.values())
    table = sorted(agg.items(), key=lambda kv: -kv[1][1])
    w = csv.writer(open(out, "w", newline="")) if out else None
    hdr = ["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage", "MinUs", "MaxUs"]
    if w:
        w.writerow(hdr)
    print(f"{'kernel':90s} {'calls':>6s} {'total us':>11s} {'avg us':>9s} {'%':>6s}")
    for name, (n, t, mn, mx) in table:
        if w:
            w.writerow([name, n, round(t, 2), round(t / n, 2), round(100 * t / tot, 2), round(mn, 2), round(mx, 2)])
        print(f"{name[:90]:90s} {n:6d} {t:11.1f} {t / n:9.1f} {100 * t / tot:6.2f}")
    print(f"total kernel time {tot / 1e3:.2f} ms over {sum(v[0] for v in agg.values())} launches")


if __name__ == "__main__":
    main()
