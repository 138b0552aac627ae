#!/usr/bin/env python3
"""Mean counter values per kernel from rocprofv3 --pmc csv output directories: python tools/pmc_summ.py DIR [substr]"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, d in acc.items():
    if sub in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "launches", max(len(v) for v in d.values()))
