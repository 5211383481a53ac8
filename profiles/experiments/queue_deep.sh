#!/bin/bash
# What holds extend_queue_kernel back on deep bounces?  One frame with the queue kernel on every wavefront bounce
# (RTMI_QUEUE_DEPTHS=7) under two SQ counter passes.   usage: profiles/experiments/queue_deep.sh -> gpurun_out/queue_deep/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/queue_deep
rm -rf $OUT; mkdir -p $OUT
export RTMI_QUEUE_DEPTHS=7
pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-visits --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.log || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  echo "pass $name done"
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE &&
pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT &&
pass tcc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-visits --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.log
find $OUT -name '*agent_info.csv' -delete
