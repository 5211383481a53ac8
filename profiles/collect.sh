#!/bin/bash
# Collects the judged profiling evidence for one round on the GPU box.
#   usage: profiles/collect.sh <tag>      (writes gpurun_out/<tag>/...)
# 1. rocprofv3 --kernel-trace --stats over the default bench command (no counters in this run)
# 2. FETCH_SIZE and WRITE_SIZE, one --pmc pass each, over one frame of the same workload
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
OUT=$R/gpurun_out/$TAG
rm -rf $OUT
mkdir -p $OUT
cd $R
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.log || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
echo "trace done: $(cat $OUT/trace_bench.json | head -c 300)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 1 --warmup 0 --no-visits --no-cpu-baseline > $OUT/$c.json 2> $OUT/$c.log || { echo "pass $c failed"; tail -5 $OUT/$c.log; exit 1; }
  echo "pass $c done"
done
find $OUT -name '*agent_info.csv' -delete
ls -R $OUT | head -40
