#!/usr/bin/env python3
"""The launches of ONE training step, in start order, from a rocprofv3 rocpd database (`--kernel-trace -d DIR -o NAME`):
everything after the second-to-last `adam_kernel` up to and including the last one.  Prints per launch the start offset,
duration, the gap to the end of the latest earlier launch, the queue, and at the end a per-name count -- the launch diet
works from this list.   usage: rocpd_step_sequence.py DB [--which K]   (K: step counted from the end, default 1)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\[clone .*\]", "", name).strip()
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("seld::", "")


def main():
    db = sys.argv[1]
    which = int(sys.argv[sys.argv.index("--which") + 1]) if "--which" in sys.argv else 1
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    q_col = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    sel = f"select {name_col}, start, end" + (f", {q_col}" if q_col else ", 0") + " from kernels order by start"
    rows = [(short(n), s, e, q) for n, s, e, q in c.execute(sel)]
    marks = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel")]
    if len(marks) < which + 1:
        sys.exit("fewer than two adam_kernel launches in this trace")
    a, b = marks[-which - 1], marks[-which]
    step = rows[a + 1:b + 1]
    t0, latest_end = step[0][1], rows[a][2]
    counts, busy = {}, 0.0
    for n, s, e, q in step:
        gap = (s - latest_end) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  q{q}  {n[:110]}")
        latest_end = max(latest_end, e)
        d = counts.setdefault(n, [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        busy += (e - s) / 1e3
    print(f"\n{len(step)} launches, {busy / 1e3:.3f} ms of kernel time, {(step[-1][2] - rows[a][2]) / 1e6:.3f} ms wall (adam end to adam end)")
    for n, (k, t) in sorted(counts.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:4d}  {t:9.1f} us  {n[:120]}")


if __name__ == "__main__":
    main()
