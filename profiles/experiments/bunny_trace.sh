export TMPDIR=/tmp
rm -rf gpurun_out/bunny_trace; mkdir -p gpurun_out/bunny_trace
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bunny_trace -- python3 bench.py --scene bunny --spp 20 --steps 5 --warmup 2 --no-cpu-baseline --no-visits > gpurun_out/bunny_trace/bench.json 2> gpurun_out/bunny_trace/log.txt
cat gpurun_out/bunny_trace/bench.json | head -c 400; echo
cut -c1-140 gpurun_out/bunny_trace/*/*kernel_stats.csv | head -10
