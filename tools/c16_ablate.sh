#!/bin/bash
# conv16_f16 step timeline under ablations (experiments variant libadn_c16tl.so built with ADN_BUILD_EXPERIMENTS=1): what a step costs
# with only the copies, only the arithmetic, no stores.  -> gpurun_out/r04_c16_ablations.txt
out=gpurun_out/r04_c16_ablations.txt; : > $out
for abl in 0 1 2 3 4 6; do
  echo "== ADN_C16_ABLATE=$abl (1 no fragment reads / MFMAs, 2 no copies after the prologue, 4 no epilogue stores)" >> $out
  ADN_C16_ABLATE=$abl ADN_F16_CONV=16 ADN_F16_FIRST=0 ADN_C16_TIMELINE=1 ADN_LIBADN_PATH=$PWD/audiodenoiser_amd/_lib/variants/libadn_c16tl.so timeout -k 10 300 python bench.py --dtype f16 --batch-per-gpu 256 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-finite-check 2>&1 >/dev/null | grep "c16 timeline" | tail -17 | sed 's/;  by wave.*//' >> $out
done
cat $out
