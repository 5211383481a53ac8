#!/bin/bash
# Calibration of the memory-side counters on known byte counts (fetch_calib.hip), one --pmc pass per counter set.
#   usage (on the GPU box, from the repo root): profiles/experiments/fetch_calib.sh   -> gpurun_out/fetch_calib/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetch_calib
BIN=$R/profiles/experiments/_build/fetch_calib
rm -rf $OUT
mkdir -p $OUT
cd /tmp
timeout -k 10 120 $BIN 1 > $OUT/plain.txt 2>&1 || { echo "plain run failed"; tail -5 $OUT/plain.txt; exit 1; }
cat $OUT/plain.txt
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $BIN > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
  echo "pass $name done"
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum || exit 1
find $OUT -name '*agent_info.csv' -delete
python3 $R/profiles/experiments/fetch_calib_report.py $OUT | tee $OUT/report.txt
