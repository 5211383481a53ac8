#!/bin/bash
# Kernel trace of one rank's share (RTMI_BENCH_SHARE, default 8) of the headline frame.
export TMPDIR=/tmp
export RTMI_BENCH_SHARE=${RTMI_BENCH_SHARE:-8}
rm -rf gpurun_out/share_trace; mkdir -p gpurun_out/share_trace
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/share_trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-visits > gpurun_out/share_trace/bench.json 2> gpurun_out/share_trace/log.txt
cut -c1-120 gpurun_out/share_trace/*/*kernel_stats.csv | head -8
python3 - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/share_trace/*/*kernel_trace.csv')[0])))
rows=[r for r in rows if 'rtk::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last frame: from last raygen
idx=[i for i,r in enumerate(rows) if 'raygen' in r['Kernel_Name']][-1]
fr=rows[idx:]
t0=int(fr[0]['Start_Timestamp'])
for r in fr:
    print('%-18s start %8.1f us dur %8.1f us' % (r['Kernel_Name'].split('(')[0][5:], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
