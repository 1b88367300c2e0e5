#!/bin/bash
# fp16 forward at small batches under the two 3x3 kernel families (ADN_F16_CONV=32 / 16 / default rule) -> gpurun_out/f16_small_batch.txt
mkdir -p gpurun_out
out=gpurun_out/f16_small_batch.txt; : > $out
for b in 1 4 16; do
  for mode in ADN_F16_CONV=32 ADN_F16_CONV=16 none; do
    envs=""; [ "$mode" != none ] && envs=$mode
    env $envs timeout -k 10 200 python bench.py --dtype f16 --batch-per-gpu $b --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>>gpurun_out/f16_small_batch.err | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('batch %-3d %-18s %8.3f ms/step' % ($b, '$mode', d['ms_per_step']))
" >> $out || { tail -5 gpurun_out/f16_small_batch.err; exit 1; }
  done
done
cat $out
