#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 --pmc rocpd database.   usage: rocpd_pmc.py DB [kernel-substring]"""
import re
import sqlite3
import sys

db = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
kn = "kernel_name" if "kernel_name" in ix else "name"
agg = {}
for r in rows:
    name = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", r[ix[kn]])).replace("seld::", "")
    if filt not in name:
        continue
    d = agg.setdefault(name, {})
    key = r[ix["counter_name"]]
    v = d.setdefault(key, [0, 0.0])
    v[0] += 1
    v[1] += float(r[ix["value"]])
    if "start" in ix and "end" in ix:
        t = d.setdefault("_dur_us", [0, 0.0])
        t[0] += 1
        t[1] += (r[ix["end"]] - r[ix["start"]]) / 1e3
for name, d in agg.items():
    print(name)
    for k, (n, s) in sorted(d.items()):
        print(f"    {k:36s} {s / n:18.1f}   (n={n})")
