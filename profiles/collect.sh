#!/bin/bash
# Collects the judged profiling evidence for one round on the GPU box.
#   usage: profiles/collect.sh <tag> [extra bench.py args]     (writes gpurun_out/<tag>/...)
# 1. rocprofv3 --kernel-trace --stats over the default bench command (no counters in this run)
# 2. one --pmc pass per counter set over one frame of the same workload (no tracing domains combined with --pmc)
# 3. the hash of the sources the counters were taken on (bench.py pairs counters only with timings of the same build)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
shift
EXTRA="$*"
OUT=$R/gpurun_out/$TAG
rm -rf $OUT
mkdir -p $OUT
cd $R
python3 -c "import bench; print(bench.csrc_hash())" > $OUT/csrc_hash.txt || exit 1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $EXTRA > $OUT/trace_bench.json 2> $OUT/trace.log || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
echo "trace done: $(head -c 300 $OUT/trace_bench.json)"
pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-visits --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.log || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  echo "pass $name done"
}
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
pass write WRITE_SIZE &&
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum &&
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU || exit 1
find $OUT -name '*agent_info.csv' -delete
ls $OUT
