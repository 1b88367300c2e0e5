#!/bin/bash
# Rehearsal of the driver's N > 1 command line on a one-GPU box: two ranks, both on device 0, gloo instead of RCCL
# (ADN_BENCH_REHEARSAL=1).  Exercises bench.py's multi-rank path end to end (256 clips per rank, gathered per-clip values,
# ranks_seen, all-gather probe); the number it prints is NOT a measurement.   -> gpurun_out/<tag>_rehearsal_n2.json
TAG=${1:-r03}
ADN_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 4 --warmup 2 > gpurun_out/${TAG}_rehearsal_n2.json 2> gpurun_out/${TAG}_rehearsal_n2.err || { tail -20 gpurun_out/${TAG}_rehearsal_n2.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_rehearsal_n2.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "value", d["value"], "ms/step", d["ms_per_step"], "config", d["config"]["workload"][:90])
print("ranks", d["ranks"])
assert d["n_gpus"] == 2 and d["ranks"]["ranks_seen"] == 2 and d["config"]["global_batch"] == 512 and d["config"]["batch_per_gpu"] == 256
PY
