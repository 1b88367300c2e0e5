#!/bin/bash
# Same-box A/B of the fp16 3x3 kernel families / settings at BASELINE configs[4] (batch 256).  Arguments: "VAR=value,VAR2=value"
# sets (default: ADN_F16_CONV=32, ADN_F16_CONV=16, none).  -> gpurun_out/f16_ab.txt
mkdir -p gpurun_out
out=gpurun_out/f16_ab.txt
: > $out
[ $# -eq 0 ] && set -- ADN_F16_CONV=32 ADN_F16_CONV=16 none
for mode in "$@"; do
    envs=""; [ "$mode" != none ] && envs=$(echo $mode | tr ',' ' ')
    env $envs timeout -k 10 300 python bench.py --dtype f16 --batch-per-gpu 256 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>>gpurun_out/f16_ab.err | python -c "
import json, sys
d = json.loads(sys.stdin.read())
t = d['forward']['per_launch_ms']
print('%-40s %8.3f ms/step %10.1f frames/s' % ('$mode', d['ms_per_step'], d['value']))
for k, v in t.items(): print('   %-22s %.4f' % (k, v))
" >> $out || { tail -5 gpurun_out/f16_ab.err; exit 1; }
done
cat $out
