#!/bin/bash
# A/B of the adn_stft_mag_fit kernels on one box (experiments variant built in the container: build.py --variant exp):
# persistent whole-line kernel vs the one-group-per-workgroup kernel, 10 k clips, n_fft 1024 and 512.  -> gpurun_out/<tag>_stft_ab.txt
TAG=${1:-r03}
export ADN_LIBADN_PATH=$PWD/audiodenoiser_amd/_lib/variants/libadn_exp.so
out=gpurun_out/${TAG}_stft_ab.txt
: > $out
for rep in 1 2; do
  for pers in 1 0; do
    for cfg in "1024 256 132300" "512 128 24000"; do
      set -- $cfg
      ADN_STFT_FIT_PERSISTENT=$pers python tools/bench_stft.py --fit --cpu-clips 0 --steps 20 --n-fft $1 --hop $2 --length $3 2>>gpurun_out/${TAG}_stft_ab.err | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('persistent=$pers n_fft $1: plain %.3f ms (frac %.4f)   fit %.3f ms (%.1f GB/s, frac %.4f)' % (d['ms_per_launch'], d['roofline']['frac'], d['fit']['ms_per_launch'], d['fit']['algorithmic_GBps'], d['fit']['frac']))
" >> $out || exit 1
    done
  done
done
cat $out
