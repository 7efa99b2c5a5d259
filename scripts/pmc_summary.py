#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output: mean counter value per dispatch, per kernel (substring filter).
usage: pmc_summary.py <dir> [kernel-substring ...]"""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if subs and not any(s in k for s in subs):
            continue
        acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out))
