#!/bin/bash
# Sweeps the live-ray count below which the rest of a pass is handed to tail_kernel (0 = never).
for t in 0 16384 65536 131072 262144 393216; do echo "tail_rays=$t $(RTMI_TAIL_RAYS=$t timeout -k 5 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-visits 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
