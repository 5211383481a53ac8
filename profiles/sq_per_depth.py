#!/usr/bin/env python3
"""Per-launch SQ counters of one kernel from the output of profiles/pmc_sq.sh or pmc_passes.sh.

usage: sq_per_depth.py <dir under gpurun_out> [kernel name part, default extend_kernel]
valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); lane_util = SQ_THREAD_CYCLES_VALU /
(SQ_ACTIVE_INST_VALU x 64)."""
import csv,glob,sys
from collections import defaultdict
root=sys.argv[1]; kern=sys.argv[2] if len(sys.argv)>2 else 'extend_kernel'
data={}
for f in glob.glob(root+'/*/*/*counter_collection.csv'):
    rows=defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            rows[r['Counter_Name']].append((int(r['Dispatch_Id']),float(r['Counter_Value'])))
    for c,v in rows.items():
        v.sort(); data[c]=[x for _,x in v]
D=data
print('depth  Mcyc  valu_busy  lane_util valu_insts(G) salu(G) wait_any wait_inst')
for i in range(min(21, len(D["GRBM_GUI_ACTIVE"]))):
    cyc=D['GRBM_GUI_ACTIVE'][i]/8
    busy=D['SQ_ACTIVE_INST_VALU'][i]*4/(cyc*1024)
    util=D['SQ_THREAD_CYCLES_VALU'][i]/(D['SQ_ACTIVE_INST_VALU'][i]*64)
    print(f"d{i:2d} {cyc/1e6:7.2f} {busy:8.2f} {util:9.2f} {D['SQ_INSTS_VALU'][i]/1e9:10.2f} {D['SQ_INSTS_SALU'][i]/1e9:7.2f} {D['SQ_WAIT_ANY'][i]/D['SQ_WAVE_CYCLES'][i]:8.2f} {D['SQ_WAIT_INST_ANY'][i]/D['SQ_WAVE_CYCLES'][i]:8.2f}")
print('total VALU insts G', sum(D['SQ_INSTS_VALU'])/1e9, 'total Mcyc', sum(D['GRBM_GUI_ACTIVE'])/8e6)
