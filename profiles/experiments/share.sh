#!/bin/bash
# Time of one rank's share of the headline frame for 1, 2, 4, 8 ranks, on one GPU (no gather): how much of the frame
# does not shrink with the number of GPUs.
for n in 1 2 4 8; do echo "share=1/$n $(RTMI_BENCH_SHARE=$n timeout -k 5 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-visits 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"gpu_ms_per_step_rank0": [0-9.]*' | tr '\n' ' ')"; done
