#!/bin/bash
# Build-time variants of the F(4x4,3x3) kernel on the GPU box (experiments build): each argument is "defines@placement",
# e.g. "-DW4_PR=5 -DW4_PR0=1@4".  -> gpurun_out/wino4_variants.txt
export ADN_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out
out=gpurun_out/wino4_variants.txt
: > $out
for v in "$@"; do
    defs="${v%@*}"; pl="${v#*@}"
    ADN_BUILD_DEFINES="$defs" python -m audiodenoiser_amd.build > /dev/null 2>gpurun_out/wino4_variants.err || { tail -5 gpurun_out/wino4_variants.err; exit 1; }
    ADN_BUILD_DEFINES="$defs" ADN_W4_PLACE=$pl python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>>gpurun_out/wino4_variants.err | python -c "
import json, sys
d = json.loads(sys.stdin.read())
t = d['forward']['per_launch_ms']
pool = sum(v for k, v in t.items() if k.endswith('+pool'))
dot = t['up4.conv2']
plain = sum(v for k, v in t.items() if ('conv1' in k or 'conv2' in k) and not k.endswith('+pool') and k not in ('up4.conv2', 'down1.conv1', 'out.conv1x1'))
print('%-28s placement %2s  %7.3f ms/step  plain(12) %7.3f  pool(4) %6.3f  dot(1) %6.3f' % ('$defs', '$pl', d['ms_per_step'], plain, pool, dot))
" >> $out || exit 1
done
cat $out
