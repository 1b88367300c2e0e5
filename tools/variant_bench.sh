#!/bin/bash
# Headline step time of library variants cross-compiled in the build container (python -m audiodenoiser_amd.build --variant NAME
# -D...), on one box, interleaved REPS times.   bash tools/variant_bench.sh TAG REPS name1 name2 ...   -> gpurun_out/TAG_variants.txt
# A name may carry environment settings: "name@VAR=value,VAR2=value"
TAG=$1; REPS=$2; shift 2
mkdir -p gpurun_out
out=gpurun_out/${TAG}_variants.txt
: > $out
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    name="${v%%@*}"; envs=""
    [ "$v" != "$name" ] && envs="${v#*@}"
    lib=$PWD/audiodenoiser_amd/_lib/variants/libadn_${name}.so
    [ -f "$lib" ] || { echo "missing $lib"; exit 1; }
    env ADN_LIBADN_PATH=$lib $(echo $envs | tr ',' ' ') timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS} 2>>gpurun_out/${TAG}_variants.err | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
t = d['forward']['per_launch_ms']
seven = sum(t[k] for k in ('down1.conv2+pool', 'down2.conv1', 'up4.conv2', 'down2.conv2+pool', 'down3.conv1', 'up3.conv2', 'up4.conv1(cat)'))
convt = sum(v for k, v in t.items() if 'convT' in k)
c3 = sum(v for k, v in t.items() if ('conv1' in k or 'conv2' in k) and k not in ('down1.conv1', 'out.conv1x1'))
print('%-34s %7.3f ms/step  frac %.4f  3x3(17) %7.3f  seven %6.3f  convT %5.3f  first %5.3f | %s' % ('$v', d['ms_per_step'], d['roofline']['frac'], c3, seven, convt, t['down1.conv1'], ' '.join('%.3f' % t[k] for k in ('down1.conv2+pool', 'down2.conv1', 'up4.conv2', 'down2.conv2+pool', 'down3.conv1', 'up3.conv2', 'up4.conv1(cat)'))))
" >> $out || { tail -3 gpurun_out/${TAG}_variants.err; exit 1; }
  done
done
cat $out
