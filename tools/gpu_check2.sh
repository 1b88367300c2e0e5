#!/bin/bash
# GPU-box check with a separate log for the new shape tests, then everything else + bench.   bash tools/gpu_check2.sh TAG
TAG=${1:-check}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_shapes.py -x -q -m gpu -s > gpurun_out/${TAG}_shapes.log 2>&1; rc=$?
tail -5 gpurun_out/${TAG}_shapes.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_check.sh $TAG
