cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 5 300 python3 -m pytest tests/test_gpu_parity.py -q -x -k "unet or fp16 or split_k or config0 or capturable" 2>&1 | tail -4
timeout -k 5 200 python3 bench.py --steps 20 --no-cpu-baseline --stft-steps 3 --no-stft-cpu > gpurun_out/q.json 2> gpurun_out/q.err; echo rc=$?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/q.json'))
print('f32', d['value'], d['ms_per_step'], d['roofline']['frac'])
print({k:v for k,v in d['forward']['per_launch_ms'].items() if 'convT' in k})
f=d['f16']; print('f16', f['frames_per_s'], f['ms_per_step'], f['roofline']['frac'], f['roofline']['avg_launch_ms'])
print(f['forward']['per_launch_ms'])
PY
