#!/usr/bin/env python3
"""HBM-side traffic per launch of every kernel, from two rocprofv3 counter passes of the SAME command:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Both counters are printed in KiB (checked: the first conv's 1.6106e9-byte output reads WRITE_SIZE = 1.574e6).  On gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced reads
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): it is doubled here.  bench.py puts the value of the dominant
kernel into roofline.traffic."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(directory, counter):
    files = glob.glob(f"{directory}/*/*counter_collection.csv")
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def label(name):
    """rocprofv3 kernel name -> the label bench.py uses (template arguments kept, namespace / parameters dropped)."""
    m = re.match(r"(?:void )?seld::([\w]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, _ = per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if "seld::" not in k:
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        res[label(k)] = dict(fetch_bytes=2.0 * f * 1024.0, write_bytes=w * 1024.0,
                             traffic_bytes=2.0 * f * 1024.0 + w * 1024.0, launches_sampled=nf.get(k, 0))
    json.dump(dict(note="bytes per launch (mean over the sampled launches); the counters print KiB, FETCH_SIZE is doubled (gfx950 correction)", kernels=res), open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes"])[:12]:
        print(f"{v['traffic_bytes'] / 1e6:10.1f} MB  {k}")


if __name__ == "__main__":
    main()
