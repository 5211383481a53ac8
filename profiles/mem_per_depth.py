#!/usr/bin/env python3
"""Per-launch memory-side counters of one kernel from the output of profiles/pmc_passes.sh.

usage: mem_per_depth.py <dir under gpurun_out> [kernel name part, default extend_kernel]
l2_hit = TCC_HIT / TCC_REQ; ea_rd = TCC_EA0_RDREQ (requests L2 sent to the fabric); ta_busy = TA_TA_BUSY summed over
the TAs / (256 CUs x kernel cycles); fetch / write = FETCH_SIZE / WRITE_SIZE in GB (KiB counters, uncorrected)."""
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else 'extend_kernel'
D = {}
for f in glob.glob(root + '/*/*/*counter_collection.csv'):
    rows = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            rows[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for c, v in rows.items():
        v.sort(); D[c] = [x for _, x in v]
print('depth  Mcyc  l2_req(M) l2_hit  ea_rd(M)  ta_busy  l1_acc(G)  fetch(GB) write(GB)')
n = min(21, len(D['GRBM_GUI_ACTIVE']))
for i in range(n):
    cyc = D['GRBM_GUI_ACTIVE'][i] / 8
    print(f"d{i:2d} {cyc/1e6:7.2f} {D['TCC_REQ_sum'][i]/1e6:9.0f} {D['TCC_HIT_sum'][i]/max(D['TCC_REQ_sum'][i],1):6.2f} "
          f"{D['TCC_EA0_RDREQ_sum'][i]/1e6:9.0f} {D['TA_TA_BUSY_sum'][i]/(256*cyc):8.2f} {D['TCP_TOTAL_CACHE_ACCESSES_sum'][i]/1e9:9.2f} "
          f"{D['FETCH_SIZE'][i]*1024/1e9:9.2f} {D['WRITE_SIZE'][i]*1024/1e9:9.2f}")
print('frame: fetch %.1f GB, write %.1f GB, l2 requests %.2f G, fabric reads %.2f G' % (
    sum(D['FETCH_SIZE']) * 1024 / 1e9, sum(D['WRITE_SIZE']) * 1024 / 1e9, sum(D['TCC_REQ_sum']) / 1e9, sum(D['TCC_EA0_RDREQ_sum']) / 1e9))
