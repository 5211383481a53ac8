#!/bin/bash
# Sweeps the wave scheduling thresholds of extend_kernel on the headline workload.
for rb in 32 40 48 52 56 60; do echo "refill_below=$rb $(RTMI_REFILL_BELOW=$rb timeout -k 5 120 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-visits 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
for lb in -16 -8 8 16; do echo "refill 48 leaf_bias=$lb $(RTMI_REFILL_BELOW=48 RTMI_LEAF_BIAS=$lb timeout -k 5 120 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-visits 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
