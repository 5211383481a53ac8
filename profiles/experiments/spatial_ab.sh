#!/bin/bash
# Headline workload under builder settings that cut more than the cost estimate justifies (profiles/experiments/spatial_splits.py
# shows what they do to the tree): does the frame gain from 1.5-4 % fewer visits per ray at 1.1-1.7 times the items?
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/spatial_$tag.json 2> gpurun_out/spatial_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/spatial_$tag.err; return 1; }
  echo "$tag $* $(python3 -c "
import json
d = json.load(open('gpurun_out/spatial_$tag.json'))
r = d['roofline']
print(round(d['ms_per_step'], 2), 'ms; build', round(d['scene_build_ms']), 'ms; nodes', d['config']['bvh_nodes'], [round(b['ms'], 2) for b in r['per_bounce']], 'tail', round(r['tail']['ms'], 2), 'nodes/ray', round(r['nodes_per_ray'], 2), 'records/ray', round(r['records_per_ray'], 2))")"
}
run off RTMI_SPATIAL=0 &&
run default RTMI_SPATIAL=1 &&
run gain11 RTMI_SPATIAL_GAIN=1.1 &&
run gain11_alpha RTMI_SPATIAL_GAIN=1.1 RTMI_SPATIAL_ALPHA=1e-7 &&
run off2 RTMI_SPATIAL=0
