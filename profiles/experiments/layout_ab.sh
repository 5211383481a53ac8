#!/bin/bash
# Fabric read requests (128-byte lines) per launch of one frame of the headline workload under the two item layouts.
#   usage (GPU box, repo root): profiles/experiments/layout_ab.sh   -> gpurun_out/layout_ab/{bfs,pairs}/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for layout in bfs pairs; do  # bfs = the default
  OUT=$R/gpurun_out/layout_ab/$layout
  rm -rf $OUT; mkdir -p $OUT
  export RTMI_LAYOUT=$layout
  timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc -- python3 bench.py --steps 1 --warmup 0 --no-visits --no-cpu-baseline > $OUT/bench.json 2> $OUT/log.txt || { echo "pass $layout failed"; tail -3 $OUT/log.txt; exit 1; }
  echo "pass $layout done"
done
find $R/gpurun_out/layout_ab -name '*agent_info.csv' -delete
