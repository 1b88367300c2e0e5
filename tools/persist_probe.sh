#!/bin/bash
# Timeline of a library variant's wino4_conv_f32 (ADN_W4_TIMELINE, experiments builds) + step times of variants.
#   bash tools/persist_probe.sh TAG timeline_variant REPS name1 name2 ...
TAG=$1; TLV=$2; REPS=$3; shift 3
mkdir -p gpurun_out
out=gpurun_out/${TAG}_timeline.txt
echo "# timeline of libadn_${TLV}.so" > $out
ADN_LIBADN_PATH=$PWD/audiodenoiser_amd/_lib/variants/libadn_${TLV}.so ADN_W4_TIMELINE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-finite-check 2>&1 >/dev/null | grep "w4 timeline" | sort -u | head -40 >> $out
cat $out
bash tools/variant_bench.sh $TAG $REPS "$@"
