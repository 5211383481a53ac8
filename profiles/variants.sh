#!/bin/bash
# A/B of library builds on the headline workload: usage profiles/variants.sh <lib> [<lib> ...] (paths under raytracing_amd/;
# a library may be named several times: boxes differ by ~1 %, runs on one box by ~0.3 %)
mkdir -p gpurun_out
i=0
for v in "$@"; do
  i=$((i + 1))
  RTMI_LIB=$PWD/raytracing_amd/$v timeout -k 10 170 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/variant_${i}_$v.json 2> gpurun_out/variant_${i}_$v.err || { echo "$v failed"; tail -3 gpurun_out/variant_${i}_$v.err; exit 1; }
  echo "$v $(python3 -c "
import json
d = json.load(open('gpurun_out/variant_${i}_$v.json'))
r = d.get('roofline') or {}
print(round(d['ms_per_step'], 2), round(d['value']), [round(b['ms'], 2) for b in r.get('per_bounce', [])], 'tail', round((r.get('tail') or {}).get('ms', 0.0), 2), 'nodes/ray', round(r.get('nodes_per_ray', 0.0), 2), 'records/ray', round(r.get('records_per_ray', 0.0), 2))")"
done
