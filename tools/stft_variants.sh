#!/bin/bash
# Build-time variants of the STFT kernels on the GPU box (experiments build): each argument is a set of -D switches.
# Runs the STFT parity tests and tools/bench_stft.py (10 k clips, n_fft 1024 / hop 256) for each.  -> gpurun_out/stft_variants.txt
export ADN_BUILD_EXPERIMENTS=1
mkdir -p gpurun_out
out=gpurun_out/stft_variants.txt
: > $out
for defs in "$@"; do
    export ADN_BUILD_DEFINES="$defs"
    python -m audiodenoiser_amd.build > /dev/null 2>gpurun_out/stft_variants.err || { tail -5 gpurun_out/stft_variants.err; exit 1; }
    t=$(timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "stft or griffin or istft or wav or config0" 2>&1 | tail -1)
    for rep in 1 2; do
        python tools/bench_stft.py --cpu-clips 0 --steps 20 2>>gpurun_out/stft_variants.err | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s %7.3f ms per 10k clips  frac %.4f | tests: $t' % ('$defs', d['ms_per_launch'], d['roofline']['frac']))
" >> $out || exit 1
    done
done
cat $out
