#!/bin/bash
# Round-end validation on the GPU box: all -m gpu tests, smoke(), rocprof kernel stats + PMC (tools/profile_bench.sh),
# default bench line, batch-256 line, parity report against the reference goldens, SQ counters.  Outputs under gpurun_out/ (TAG_*).
#     gpurun --timeout 1200 -- 'bash tools/round_end_run.sh r05'
set -e
TAG=${1:-r05}
python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1 || { tail -20 gpurun_out/${TAG}_pytest_gpu.log; exit 1; }
tail -1 gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
bash tools/profile_bench.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1 || { tail -20 gpurun_out/${TAG}_profile.log; exit 1; }
tail -3 gpurun_out/${TAG}_profile.log
cp gpurun_out/prof_${TAG}/summary/pmc_traffic.json profiles/pmc_traffic.json     # the bench line below reads the traffic of THIS build
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --batch-per-gpu 256 --steps 30 --no-extras --no-cpu-baseline > gpurun_out/${TAG}_bench_b256.json 2>/dev/null
(python tools/report_parity.py; ADN_WINO_TILE=2 python tools/report_parity.py | sed 's/^winograd /wino F(2,3)/'; ADN_CONV_ALGO=direct python tools/report_parity.py) > gpurun_out/${TAG}_parity.txt 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['stft']['ms_per_launch'], d['f16']['frames_per_s'], d['f16']['ms_per_step'], d['f16']['forward']['convt']['ms'], d['forward']['convt']['ms'], d['cpu_baseline']['value'])
d=json.load(open('gpurun_out/${TAG}_bench_b256.json')); print(d['value'], d['ms_per_step'])"
cat gpurun_out/${TAG}_parity.txt
# matrix-pipe busy / clock / wave-state counters of the dominant kernels of the FINAL build (fp32: wino4_conv_f32; fp16: conv16 + convt16)
bash tools/sq_counters.sh wino4 > gpurun_out/${TAG}_wino4_sq_counters.txt 2>&1 || true
BENCH_ARGS="--dtype f16 --batch-per-gpu 256" bash tools/sq_counters.sh conv > gpurun_out/${TAG}_f16_sq_counters.txt 2>&1 || true
tail -12 gpurun_out/${TAG}_wino4_sq_counters.txt
