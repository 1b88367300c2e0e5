#!/bin/bash
# GPU parity tests of the network against a library variant:  bash tools/variant_test.sh TAG name [pytest -k expression]
TAG=$1; NAME=$2; K=${3:-"unet or golden or f4 or split or taps or config0 or reference or fp16"}
ADN_LIBADN_PATH=$PWD/audiodenoiser_amd/_lib/variants/libadn_${NAME}.so timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$K" > gpurun_out/${TAG}_pytest_${NAME}.log 2>&1
rc=$?; tail -3 gpurun_out/${TAG}_pytest_${NAME}.log; exit $rc
