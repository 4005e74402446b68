#!/usr/bin/env python3
"""Writes tests/golden/dubrovnik_queries.json: image size and known focal length (0 = unknown)
of the 800 Dubrovnik6K queries, taken line by line from the reference's
data/Dubrovnik6K/list.query.dimensions.txt and list.query.txt.  Only these numbers are kept
(no image names); they shape the synthetic query batch of the benchmark."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/Dubrovnik6K"

dims = [l.split() for l in open(os.path.join(REF, "list.query.dimensions.txt")) if l.strip()]
foc = [l.split() for l in open(os.path.join(REF, "list.query.txt")) if l.strip()]
assert len(dims) == len(foc) == 800
out = []
for d, f in zip(dims, foc):
    assert d[0] == f[0]
    out.append([int(d[1]), int(d[2]), float(f[2]) if len(f) >= 3 else 0.0])
with open(os.path.join(ROOT, "tests", "golden", "dubrovnik_queries.json"), "w") as fh:
    json.dump({"columns": ["width", "height", "focal_or_0"], "queries": out}, fh)
print(len(out), "queries,", sum(1 for q in out if q[2] > 0), "with known focal length")
