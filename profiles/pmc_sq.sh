#!/bin/bash
# one SQ counter pass over a frame of the headline workload: usage profiles/pmc_sq.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- python3 bench.py --steps 1 --warmup 0 --no-visits --no-cpu-baseline > $OUT/sq1.log 2>&1 || { echo "pass failed"; tail -3 $OUT/sq1.log; exit 1; }
echo done
