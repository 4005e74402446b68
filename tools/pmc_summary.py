#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel: tools/pmc_summary.py DIR... > summary.csv"""
import collections
import csv
import glob
import os
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
names = sorted({c for v in rows.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + names)
for k, v in sorted(rows.items()):
    nd = max(len(calls[(k, c)]) for c in v)
    w.writerow([k, nd] + ["%.6g" % v.get(c, float("nan")) for c in names])
