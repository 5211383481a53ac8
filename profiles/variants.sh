#!/bin/bash
# A/B of library builds on the headline workload: usage profiles/variants.sh <lib> [<lib> ...] (paths under raytracing_amd/)
mkdir -p gpurun_out
for v in "$@"; do
  RTMI_LIB=$PWD/raytracing_amd/$v timeout -k 10 170 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-visits > gpurun_out/variant_$v.json 2> gpurun_out/variant_$v.err || { echo "$v failed"; tail -3 gpurun_out/variant_$v.err; exit 1; }
  echo "$v $(python3 -c "import json,sys; d=json.load(open('gpurun_out/variant_$v.json')); print(d['ms_per_step'], d['value'])")"
done
