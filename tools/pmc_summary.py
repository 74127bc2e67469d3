#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, the mean counter value per dispatch.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, i.e. half the bytes (MI355X_MICROARCH.md, HBM section):
the 'corrected' column doubles it.

  pmc_summary.py DIR [DIR ...] [--json OUT]

With --json, writes {bench.py kernel key: HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE} for the
kernels bench.py reports (the file bench.py reads as profiles/traffic_bytes_per_launch.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

BENCH_KEYS = {"gray_kernel": "gray_kernel<", "gauss_sep_kernel<2>": "gauss_sep_kernel<2,",
              "median3_kernel": "median3_kernel<", "sobel_kernel": "sobel_kernel<",
              "chain_kernel": "chain_kernel<"}

args = sys.argv[1:]
json_out = None
if "--json" in args:
    i = args.index("--json")
    json_out = args[i + 1]
    del args[i:i + 2]

total = defaultdict(dict)
for d in args:
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, counters in acc.items():
        for c, vals in counters.items():
            mean_kib = sum(vals) / len(vals)
            corr = mean_kib * 2 if c == "FETCH_SIZE" else mean_kib
            total[k][c] = corr * 1024.0
            print(f"{k[:70]:70s} {c:11s} n={len(vals):3d} mean={mean_kib / 1024:10.1f} MiB  corrected={corr / 1024:10.1f} MiB")

if json_out:
    out = {}
    for key, sub in BENCH_KEYS.items():
        for k, c in total.items():
            if sub in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                out[key] = c["FETCH_SIZE"] + c["WRITE_SIZE"]
    json.dump(out, open(json_out, "w"), indent=1, sort_keys=True)
    print("wrote", json_out, out)
