#!/bin/bash
# Per-launch times of the four transposed convolutions (+ the layers around them) of variants:  bash tools/convt_times.sh TAG name[@ENV=..] ...
TAG=$1; shift
out=gpurun_out/${TAG}_convt.txt; : > $out
for v in "$@"; do
  name="${v%%@*}"; envs=""; [ "$v" != "$name" ] && envs="${v#*@}"
  env ADN_LIBADN_PATH=$PWD/audiodenoiser_amd/_lib/variants/libadn_${name}.so $(echo $envs | tr ',' ' ') python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS} 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['forward']['per_launch_ms']
print('%-30s step %.3f | convT up1..4: %s | conv1(cat) up1..4: %s' % ('$v', d['ms_per_step'], ' '.join('%.3f' % t['up%d.convT' % i] for i in (1,2,3,4)), ' '.join('%.3f' % t['up%d.conv1(cat)' % i] for i in (1,2,3,4))))
" >> $out
done
cat $out
