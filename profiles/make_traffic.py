#!/usr/bin/env python3
"""Turns the output of profiles/collect.sh into the committed evidence of a round.

    python profiles/make_traffic.py gpurun_out/<tag> <round-name> [workload-key]

Writes profiles/<round-name>_kernel_stats.csv (the rocprofv3 --kernel-trace --stats summary of the bench command),
profiles/<round-name>_traffic_summary.json (per-kernel FETCH_SIZE / WRITE_SIZE sums of one frame) and updates
profiles/traffic.json, which bench.py reads for roofline.traffic.

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE tallies the
128-byte requests of wide loads at 64 bytes.  The kernels here gather 64-byte nodes and records with four 16-byte loads
per lane, a pattern that correction has not been calibrated on, so the figure is an upper estimate of the read side.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, name = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else "dragon_1920x1080_100spp"
here = os.path.dirname(os.path.abspath(__file__))


def short(kernel):
    return kernel.split("(")[0].replace("void ", "")


def per_kernel(counter):
    tot, calls = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(src, counter, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            calls[k] += 1
    return tot, calls


fetch, calls = per_kernel("FETCH_SIZE")
write, _ = per_kernel("WRITE_SIZE")
summary = {}
for k in sorted(fetch):
    if not k.startswith("rtk::"):
        continue
    summary[k] = {
        "launches_per_frame": calls[k],
        "fetch_size_kib": fetch[k],
        "write_size_kib": write.get(k, 0.0),
        "hbm_bytes_per_frame": (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0,
    }
ext = [k for k in summary if "extend_kernel" in k or "extend_queue_kernel" in k]  # the traversal launch of every bounce
launches = sum(summary[k]["launches_per_frame"] for k in ext)
total = sum(summary[k]["hbm_bytes_per_frame"] for k in ext)
stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(here, name + "_kernel_stats.csv"))
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, one frame of `bench.py --steps 1 --warmup 0`",
           "workload": key, "kernels": summary}, open(os.path.join(here, name + "_traffic_summary.json"), "w"), indent=1)
tfile = os.path.join(here, "traffic.json")
tj = json.load(open(tfile)) if os.path.exists(tfile) else {}
tj[key] = {"kernel": "extend_kernel + extend_queue_kernel", "launches": launches, "hbm_bytes_per_launch": total / max(launches, 1),
           "hbm_bytes_per_frame": total, "from": name + "_traffic_summary.json"}
json.dump(tj, open(tfile, "w"), indent=1)
print(json.dumps(tj[key]))
