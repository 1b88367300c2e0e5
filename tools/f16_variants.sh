#!/bin/bash
# Build-time variants of the fp16 path on the GPU box (experiments build): each argument is a set of -D switches.
# Runs the fp16 parity tests and the configs[4] bench (batch 256) for each.  -> gpurun_out/f16_variants.txt
export ADN_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out
out=gpurun_out/f16_variants.txt
: > $out
for defs in "$@"; do
    export ADN_BUILD_DEFINES="$defs"
    python -m audiodenoiser_amd.build > /dev/null 2>gpurun_out/f16_variants.err || { tail -5 gpurun_out/f16_variants.err; exit 1; }
    t=$(timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "fp16 or f16" 2>&1 | tail -1)
    python bench.py --dtype f16 --batch-per-gpu 256 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>>gpurun_out/f16_variants.err | python -c "
import json, sys
d = json.loads(sys.stdin.read())
t = d['forward']['per_launch_ms']
print('%-24s %8.3f ms/step %10.1f frames/s frac %.4f | %s | tests: $t' % ('$defs', d['ms_per_step'], d['value'], d['roofline']['frac'], ' '.join('%s %.2f' % (k.split('.')[0][:3] + k.split('.')[1][:6], v) for k, v in t.items() if 'conv1' in k or 'conv2' in k)))
" >> $out || exit 1
done
cat $out
