#!/bin/bash
# The other benchmark configurations (Cornell, bunny, diffuse dragon, 4K) on one GPU (one frame each, after one warm-up where it is cheap).
mkdir -p gpurun_out/configs
run() { name=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/configs/$name.json 2> gpurun_out/configs/$name.err || { echo "$name failed"; tail -3 gpurun_out/configs/$name.err; exit 1; }; echo "$name $(python3 -c "import json; d=json.load(open('gpurun_out/configs/$name.json')); r=d['roofline']; print('%.1f ms/frame %.0f Mrays/s rays/path %.2f nodes/ray %.1f records/ray %.2f algorithmic bytes / HBM peak %.2f' % (d['ms_per_step'], d['value'], d['rays_per_path'], r['nodes_per_ray'], r['records_per_ray'], r['frac']))")"; }
run cornell_1080p_5spp --scene cornell-box --spp 5 --steps 3 --warmup 1
run bunny_1080p_20spp --scene bunny --spp 20 --steps 3 --warmup 1
run dragon_diffuse_1080p_100spp --scene dragon-diffuse --spp 100 --steps 2 --warmup 1
run dragon_4k_100spp --scene dragon --width 3840 --height 2160 --spp 100 --steps 1 --warmup 1
run cornell_4k_1000spp --scene cornell-box --width 3840 --height 2160 --spp 1000 --steps 1 --warmup 0
