#!/usr/bin/env python3
"""Per-bounce kernel durations of the last frame from a rocprofv3 kernel trace CSV.  usage: depth_times.py <trace.csv> [max_depth]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 64
b = [r for r in rows if "bounce" in r["Kernel_Name"]][-depth:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in b]
print("per-bounce us:", [round(x) for x in d])
print("sum ms %.2f  span ms %.2f" % (sum(d) / 1e3, (int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])) / 1e6))
acc = [r for r in rows if "accumulate" in r["Kernel_Name"]]
if acc:
    print("accumulate us:", round((int(acc[-1]["End_Timestamp"]) - int(acc[-1]["Start_Timestamp"])) / 1e3))
