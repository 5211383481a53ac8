#!/usr/bin/env python3
"""Per-bounce kernel durations of the last frame from a rocprofv3 kernel trace CSV.

usage: depth_times.py <kernel_trace.csv> <launches per frame> [kernel name part, default extend_kernel]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per_frame = int(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "extend_kernel"
b = sorted((r for r in rows if name in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))[-per_frame:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in b]
print("per-bounce us:", [round(x) for x in d])
print("sum ms %.2f  span ms %.2f" % (sum(d) / 1e3, (int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])) / 1e6))
