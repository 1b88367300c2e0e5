#!/bin/bash
# Where the plain STFT's time goes, on the CURRENT build (experiments variant libadn_exp.so): ADN_STFT_ABLATE bits
# 1 = no audio loads, 2 = only one row block of stores, 4 = FFT passes skipped.   -> gpurun_out/<tag>_stft_ablations.txt
TAG=${1:-r03}
export ADN_LIBADN_PATH=$PWD/audiodenoiser_amd/_lib/variants/libadn_exp.so
out=gpurun_out/${TAG}_stft_ablations.txt
echo "# tools/stft_ablations.sh: 10 000 clips x 132 300 samples, n_fft 1024, hop 256, centred (BASELINE configs[2]); ms per launch" > $out
for abl in 0 4 5 7 3 1 2; do
  case $abl in
    0) what="full kernel";; 4) what="FFT passes removed (loads, window, post-processing, image, stores kept)";;
    5) what="... and no audio loads";; 7) what="neither FFT, loads nor stores (launch, constant set-up, window, post-processing, image writes, barriers)";;
    3) what="full FFT but neither loads nor stores";; 1) what="no audio loads";; 2) what="only one row block of stores";;
  esac
  ADN_STFT_ABLATE=$abl python tools/bench_stft.py --cpu-clips 0 --steps 20 2>>gpurun_out/${TAG}_stft_ablations.err | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-110s %7.3f' % ('$what (ADN_STFT_ABLATE=$abl)', d['ms_per_launch']))
" >> $out || exit 1
done
cat $out
