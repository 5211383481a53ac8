#!/usr/bin/env python3
"""Per-launch counter values of profiles/experiments/fetch_calib.sh against the byte counts the kernels are known to move."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
RECORDS = 1 << 26
recs = 256 * 6 * 256 * 256
known = {  # kernel -> (unit count, unit name, useful bytes)
    "stream_read": (RECORDS * 64 / 128, "128-B lines", RECORDS * 64),
    "stream_write": (RECORDS * 16 / 128, "128-B lines", RECORDS * 16),
    "gather<64>": (recs, "64-B records", recs * 64),
    "gather<128>": (recs, "128-B records", recs * 128),
    "gather_pair": (recs, "128-B lines", recs * 128),
}
vals = defaultdict(dict)
for f in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    rows = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, counter), v in rows.items():
        vals[name][counter] = v[-1]  # the timed launch (the first is the warm-up)
for name, (units, unit_name, useful) in known.items():
    v = vals.get(name, {})
    print("%-13s %.4g %s, %.4g useful bytes" % (name, units, unit_name, useful))
    for c in sorted(v):
        x = v[c]
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            print("    %-22s %.6g KiB = %.4g bytes = %.3f x useful bytes, %.1f bytes per %s" %
                  (c, x, x * 1024, x * 1024 / useful, x * 1024 / units, unit_name[:-1]))
        else:
            print("    %-22s %.6g = %.3f per %s" % (c, x, x / units, unit_name[:-1]))
