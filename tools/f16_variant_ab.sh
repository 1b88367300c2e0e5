#!/bin/bash
# fp16 step time of library variants (python -m audiodenoiser_amd.build --variant NAME -D...; "prod" = the in-tree production library),
# interleaved REPS times on one box.   bash tools/f16_variant_ab.sh TAG REPS prod name1 ...   -> gpurun_out/TAG_f16_ab.txt
TAG=$1; REPS=$2; shift 2
mkdir -p gpurun_out
out=gpurun_out/${TAG}_f16_ab.txt; : > $out
for v in "$@"; do
  [ "$v" = prod ] && continue
  ADN_LIBADN_PATH=$PWD/audiodenoiser_amd/_lib/variants/libadn_${v}.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -m gpu -k "fp16 or f16" > gpurun_out/${TAG}_pytest_${v}.log 2>&1 || { tail -5 gpurun_out/${TAG}_pytest_${v}.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/${TAG}_pytest_${v}.log)" >> $out
done
run() {
  v=$1
  if [ "$v" = prod ]; then lib=""; else lib=$PWD/audiodenoiser_amd/_lib/variants/libadn_${v}.so; fi
  env ${lib:+ADN_LIBADN_PATH=$lib} timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --extras f16 --f16-steps 10 --no-f16-b1 2>>gpurun_out/${TAG}_f16_ab.err | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['f16']; t = f['forward']['per_launch_ms']
keys = ('down1.conv2+pool', 'down2.conv1', 'down2.conv2+pool', 'down3.conv1', 'bottleneck.conv2', 'up1.conv1(cat)', 'up3.conv1(cat)', 'up3.conv2', 'up4.conv1(cat)', 'up4.conv2')
print('%-12s f16 %.3f ms/step | %s | convT %s' % ('$v', f['ms_per_step'], ' '.join('%.3f' % t[k] for k in keys), ' '.join('%.3f' % t['up%d.convT' % i] for i in (1, 2, 3, 4))))" >> $out
}
for rep in $(seq 1 $REPS); do
  for v in "$@"; do run $v; done
  for (( i=$#; i>0; i-- )); do run ${!i}; done
done
cat $out
