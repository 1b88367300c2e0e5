#!/bin/bash
# fp32 transposed-convolution work on the GPU box: the tests that cover it, then the headline step with its per-launch times.   bash tools/f32_convt_check.sh TAG
TAG=${1:-f32ct}
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -m gpu -k "convt or golden and not fp16 or nonfinite_pixels_travel" > gpurun_out/${TAG}_tests.log 2>&1; rc=$?
tail -4 gpurun_out/${TAG}_tests.log
[ $rc -eq 0 ] || exit $rc
for mode in "" dma; do
ADN_CONVT_SPLIT=${mode:-1} timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_bench.err | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['forward']['per_launch_ms']
print('%-8s step %.3f | convT up1..4: %s sum %.3f' % ('${mode:-ring}', d['ms_per_step'], ' '.join('%.3f' % t['up%d.convT' % i] for i in (1,2,3,4)), sum(t['up%d.convT' % i] for i in (1,2,3,4))))"
done
