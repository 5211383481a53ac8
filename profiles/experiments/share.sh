#!/bin/bash
# Strong-scaling rehearsal on one GPU: the frame time of one rank's interleaved share of the tiles (no gather).
# usage (on the GPU box): bash profiles/experiments/share.sh
cd ${GRAFT_REPO_ROOT:-.}
for n in 1 2 4 8; do
  r=$(RTMI_BENCH_SHARE=$n timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-visits 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['value'],1))") || exit 1
  echo "share 1/$n: ms/frame, Mrays/s = $r"
done
