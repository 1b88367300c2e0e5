#!/bin/bash
# Per-workgroup timeline (ADN_W4_TIMELINE) and timing ablations (ADN_WINO4_ABLATE) of wino4_conv_f32 on the CURRENT build
# (experiments variant libadn_exp.so, cross-compiled: python -m audiodenoiser_amd.build --variant exp).  -> gpurun_out/<tag>_wino4_timeline.txt
TAG=${1:-r03}
export ADN_LIBADN_PATH=$PWD/audiodenoiser_amd/_lib/variants/libadn_exp.so
out=gpurun_out/${TAG}_wino4_timeline.txt
echo "# tools/wino4_timeline.sh: batch 64 x 513x256, experiments build of the final round-3 kernel; s_memtime stamps of wave 0 per workgroup" > $out
ADN_W4_TIMELINE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-finite-check 2>&1 >/dev/null | grep "w4 timeline" | sort -u | head -24 >> $out
echo "#" >> $out
echo "# timing ablations (ADN_WINO4_ABLATE; results wrong by design): ms per step and selected launches" >> $out
for ab in 0 1 2 4 8 16 2048 4096 27; do
    ADN_WINO4_ABLATE=$ab python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-finite-check 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
t = d['forward']['per_launch_ms']
keys = ['down1.conv2+pool', 'down2.conv2+pool', 'down4.conv2+pool', 'up1.conv1(cat)', 'up3.conv1(cat)', 'up4.conv1(cat)']
c3 = sum(v for k, v in t.items() if ('conv1' in k or 'conv2' in k) and k not in ('down1.conv1', 'out.conv1x1'))
print('ablate %4s  %7.3f ms/step  3x3 %7.3f  ' % ('$ab', d['ms_per_step'], c3) + '  '.join('%s %.3f' % (k, t[k]) for k in keys))
" >> $out
done
cat $out
