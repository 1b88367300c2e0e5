#!/bin/bash
# fp16 transposed convolutions at small batches: convt16_f16 (default) against conv_dma<_Float16, ..., CONVT2X2> (ADN_F16_CONVT=dma); per-launch ms
mkdir -p gpurun_out
out=gpurun_out/f16_convt_small_batch.txt; : > $out
for b in 1 2 4 8 16 64; do
  for mode in none ADN_F16_CONVT=dma; do
    envs=""; [ "$mode" != none ] && envs=$mode
    env $envs timeout -k 10 200 python bench.py --dtype f16 --batch-per-gpu $b --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>>gpurun_out/f16_convt_small_batch.err | python -c "
import json, sys
d = json.loads(sys.stdin.read()); t = d['forward']['per_launch_ms']
print('batch %-3d %-18s %8.3f ms/step | convT up1..4 %s  sum %.4f' % ($b, '$mode', d['ms_per_step'], ' '.join('%.4f' % t['up%d.convT' % i] for i in (1, 2, 3, 4)), sum(t['up%d.convT' % i] for i in (1, 2, 3, 4))))
" >> $out || { tail -5 gpurun_out/f16_convt_small_batch.err; exit 1; }
  done
done
cat $out
