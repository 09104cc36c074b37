#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) into per-kernel HBM bytes per launch.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
bytes = FETCH_SIZE*1024*2 (gfx950 under-count of coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024."""
import csv, glob, json, os, sys, collections


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"_method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes over `python bench.py --steps 1 "
       "--warmup 1 --no-cpu-baseline`; bytes = FETCH_SIZE*1024*2 (gfx950 under-count of coalesced reads, "
       "MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024", "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    n = max(fe.get(k, [0])[0], wr.get(k, [0])[0]) or 1
    rb = fe[k][1] * 1024 * 2 / max(fe[k][0], 1) if k in fe else 0.0
    wb = wr[k][1] * 1024 / max(wr[k][0], 1) if k in wr else 0.0
    out["kernels"][k] = {"launches": n, "read_bytes_per_launch": rb, "write_bytes_per_launch": wb,
                         "hbm_bytes_per_launch": rb + wb}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("kernels:", len(out["kernels"]))
