#!/bin/bash
# fp16 path on the GPU box: parity tests of the fp16 kernels, then the f16 sub-benchmark with its per-launch times.   bash tools/f16_check.sh TAG
TAG=${1:-f16}
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -m gpu -k "fp16 or f16" > gpurun_out/${TAG}_f16_tests.log 2>&1; rc=$?
tail -4 gpurun_out/${TAG}_f16_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --extras f16 --f16-steps 10 2>gpurun_out/${TAG}_f16_bench.err > gpurun_out/${TAG}_f16_bench.json || { tail -5 gpurun_out/${TAG}_f16_bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_f16_bench.json").read().strip().splitlines()[-1]); f = d["f16"]
t = f.get("forward", {}).get("per_launch_ms", f.get("per_launch_ms"))
print("f16 ms/step", f["ms_per_step"], "frac", f["roofline"]["frac"])
print("convT", " ".join("%.3f" % t["up%d.convT" % i] for i in (1, 2, 3, 4)), "sum %.3f" % sum(t["up%d.convT" % i] for i in (1, 2, 3, 4)))
print(" ".join("%s=%.3f" % (k.replace(".conv", "."), v) for k, v in t.items() if "convT" not in k))
PY
