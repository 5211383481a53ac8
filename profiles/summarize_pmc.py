#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel name.  usage: summarize_pmc.py <dir with pass subdirs>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {k: {c: {"sum": v, "dispatches": len(calls[k][c])} for c, v in cs.items()} for k, cs in tot.items()}
print(json.dumps(out, indent=1, sort_keys=True))
