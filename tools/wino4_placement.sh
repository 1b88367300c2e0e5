#!/bin/bash
# Copy-piece placement sweep of the F(4x4,3x3) kernel (W4_PLACE: which MFMA groups of a pass are followed by one of its five copy pieces) (experiments build: ADN_BUILD_EXPERIMENTS=1 python -m audiodenoiser_amd.build).
# Usage (GPU box): bash tools/wino4_placement.sh [placements...]  -> gpurun_out/wino4_placement.txt
# Every placement is arithmetically identical; the table shows, per epilogue variant, the summed time of its launches.
export ADN_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out
out=gpurun_out/wino4_placement.txt
: > $out
for pl in ${@:-0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15}; do
    ADN_W4_PLACE=$pl python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>gpurun_out/wino4_placement.err | python -c "
import json, sys
d = json.loads(sys.stdin.read())
t = d['forward']['per_launch_ms']
pool = sum(v for k, v in t.items() if k.endswith('+pool'))
dot = t['up4.conv2']
plain = sum(v for k, v in t.items() if ('conv1' in k or 'conv2' in k) and not k.endswith('+pool') and k not in ('up4.conv2', 'down1.conv1', 'out.conv1x1'))
print('placement %2s  %7.3f ms/step  plain(12) %7.3f  pool(4) %6.3f  dot(1) %6.3f' % ('$pl', d['ms_per_step'], plain, pool, dot))
" >> $out || exit 1
done
cat $out
