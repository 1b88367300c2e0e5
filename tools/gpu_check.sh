#!/bin/bash
# GPU-box check of the current tree: all -m gpu tests, smoke(), default bench line.  Outputs under gpurun_out/<tag>_*.
#     gpurun --timeout 1200 -- 'bash tools/gpu_check.sh r03a'
TAG=${1:-check}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/${TAG}_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
timeout -k 10 500 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "ranks", d["ranks"])
print("stft", d["stft"]["ms_per_launch"], d["stft"]["roofline"]["frac"], "fit", d["stft"]["fused_to_network_input"]["ms_per_launch"])
print("f16", d["f16"]["ms_per_step"], d["f16"]["roofline"]["frac"])
print("fp32_b256", d["fp32_b256"]["value"], d["fp32_b256"]["ms_per_step"], d["fp32_b256"]["frac"])
for k in ("default", "batch_invariant", "serving"): print("b1", k, json.dumps(d["b1"][k]))
print("e2e", json.dumps(d.get("e2e_config0")))
print("exact_f32", json.dumps(d.get("exact_f32")))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for k, v in d["forward"]["per_launch_ms"].items(): print("  %-22s %.4f" % (k, v))
PY
