#!/bin/bash
# Timing ablations of the F(4x4,3x3) kernel (experiments build only: ADN_BUILD_EXPERIMENTS=1 python -m audiodenoiser_amd.build).
# Usage (GPU box): bash tools/wino4_experiments.sh [ablate values...]  -> gpurun_out/wino4_ablate.txt
# ADN_WINO4_ABLATE bits: 1 no copies after the first chunk, 2 no patch reads / transform, 4 no transform, 8 no barrier,
# 16 no B-fragment reads (results are wrong by design; only the launch times are read).
export ADN_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out
out=gpurun_out/wino4_ablate.txt
: > $out
for ab in ${@:-0 1 2 4 8 16 9 18 19 27}; do
    ADN_WINO4_ABLATE=$ab python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-finite-check 2>gpurun_out/wino4_ablate.err | python -c "
import json, sys
d = json.loads(sys.stdin.read())
t = d['forward']['per_launch_ms']
keys = ['down2.conv2+pool', 'down4.conv2+pool', 'up1.conv1(cat)', 'up1.conv2', 'up3.conv1(cat)', 'up4.conv1(cat)']
print('ablate %3s  %8.1f frames/s  %7.3f ms/step  ' % ('$ab', d['value'], d['ms_per_step']) + '  '.join('%s %.3f' % (k, t[k]) for k in keys))
" >> $out || exit 1
done
cat $out
