#!/bin/bash
# Batch-size sweep of the fp32 forward on one GPU: default kernel choice and ADN_WINO_TILE=2.
# Usage (GPU box): bash tools/batch_sweep.sh > gpurun_out/batch_sweep.txt
for mode in auto 2; do
    for b in 1 2 8 16 64 128; do
        if [ "$mode" = "2" ]; then export ADN_WINO_TILE=2; else unset ADN_WINO_TILE; fi
        python bench.py --batch-per-gpu $b --steps 50 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('wino_tile=%-4s batch %3d  %8.3f ms/step  %9.1f frames/s' % ('$mode', $b, d['ms_per_step'], d['value']))"
    done
done
