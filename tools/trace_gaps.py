#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of `bench.py --mode eager|graph`: per training step (delimited by adam kernels),
wall time, time with at least one kernel running, and the idle gaps.   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
ends = [e for s, e, n in rows if "adam" in n]
steps = list(zip(ends[:-1], ends[1:]))
sel = steps[len(steps) // 2: len(steps) // 2 + 20]
tot_wall = tot_busy = 0
gaps = []
for a, b in sel:
    ks = [(s, e) for s, e, n in rows if s >= a and e <= b]
    busy, cur_s, cur_e = 0, None, None
    for s, e in ks:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot_wall += b - a
    tot_busy += busy
n = len(sel)
print(f"steps {n}: wall {tot_wall / n / 1e6:.3f} ms, >=1 kernel running {tot_busy / n / 1e6:.3f} ms, idle {(tot_wall - tot_busy) / n / 1e6:.3f} ms")
gaps.sort(reverse=True)
print("gaps per step > 20 us:", sum(1 for g in gaps if g > 20000) / n, " > 5 us:", sum(1 for g in gaps if g > 5000) / n,
      " largest (us):", [round(g / 1e3) for g in gaps[:8]])
