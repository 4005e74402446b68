#!/usr/bin/env python3
"""Sums the counters of a rocprofv3 --pmc csv run per kernel name prefix.
Usage: tools/pmc_kernel.py DIR [name-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if sub not in name:
            continue
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
for name, c in tot.items():
    print(name, "dispatches", len(calls[name]))
    for k, v in sorted(c.items()):
        print("   %-28s %.4g" % (k, v))
