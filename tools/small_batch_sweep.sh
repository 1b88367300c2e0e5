#!/bin/bash
# Small-batch sweep of the forward step (513x256) under the automatic kernel choice and under ADN_BATCH_INVARIANT=1 (one kernel per layer:
# no split-K slices anywhere), fp32 and fp16 -> gpurun_out/small_batch_sweep.txt.  "No batch size slower than the pinned form" is the claim.
mkdir -p gpurun_out
out=gpurun_out/small_batch_sweep.txt; : > $out
for dt in f32 f16; do
  for b in 1 2 3 4 6 8 12 16 24 32; do
    for mode in none ADN_BATCH_INVARIANT=1; do
      envs=""; [ "$mode" != none ] && envs=$mode
      env $envs timeout -k 10 200 python bench.py --dtype $dt --batch-per-gpu $b --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>>gpurun_out/small_batch_sweep.err | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%s batch %-3d %-22s %8.3f ms/step' % ('$dt', $b, '$mode', d['ms_per_step']))
" >> $out || { tail -5 gpurun_out/small_batch_sweep.err; exit 1; }
    done
  done
done
cat $out
