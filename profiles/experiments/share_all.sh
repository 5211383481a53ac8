#!/bin/bash
# Load balance of the interleaved tile partition: every one of the 8 shares of the headline frame, one after the other
# on one GPU (no gather between GPUs: what each rank of an 8-GPU job would spend rendering).
#   usage (GPU box, repo root): profiles/experiments/share_all.sh [N=8]  -> gpurun_out/share_all.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-8}
cd $R
OUT=$R/gpurun_out/share_all.txt
: > $OUT
for r in $(seq 0 $((N - 1))); do
  RTMI_BENCH_SHARE=$N RTMI_BENCH_SHARE_RANK=$r timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-visits --no-cpu-baseline > $R/gpurun_out/share_$r.json 2> $R/gpurun_out/share_$r.log || { echo "share $r failed" >> $OUT; tail -3 $R/gpurun_out/share_$r.log; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$R/gpurun_out/share_$r.json')); print('share %d of %d: %.2f ms/frame, %.1f Mrays/frame, %.2f rays/path' % ($r, $N, d['ms_per_step'], d['value']*d['ms_per_step']/1e3, d['rays_per_path']))" >> $OUT
done
python3 - >> $OUT <<PY
import json
ms=[json.load(open('$R/gpurun_out/share_%d.json' % r))['ms_per_step'] for r in range($N)]
print('max %.2f ms, mean %.2f ms, max/mean %.3f' % (max(ms), sum(ms)/len(ms), max(ms)/(sum(ms)/len(ms))))
PY
cat $OUT
