#!/bin/bash
# Profile bench.py on the GPU box: kernel stats + two PMC passes (FETCH_SIZE, WRITE_SIZE) of the SAME command, then
# summarise into profiles/ (tracked).  Run through gpurun from the repo root:
#     gpurun --timeout 1100 -- 'bash tools/profile_bench.sh r02'
# The program after `--` is python3 itself (no wrappers: the profiler initialises the GPU before the program starts).
set -eo pipefail
TAG=${1:-r04}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --extras stft,f16 --stft-steps 25 --f16-steps 3 --no-f16-b1 --no-stft-cpu"
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1) && echo "stats pass done"
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1) && echo "fetch pass done"
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1) && echo "write pass done"
python3 tools/pmc_summary.py --stats "$OUT/stats" --fetch "$OUT/fetch" --write "$OUT/write" --tag "${TAG}_bench" \
    --out "$OUT/summary" --cmd "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --extras stft,f16 --stft-steps 25 --f16-steps 3 --no-f16-b1 --no-stft-cpu" --traffic-json
grep -h '^{' "$OUT/stats.log" | tail -1 > "$OUT/summary/${TAG}_bench_under_rocprof.json" || true
find "$OUT/stats" -name '*_kernel_stats.csv' -exec cp {} "$OUT/summary/${TAG}_bench_kernel_stats.csv" \;
# Griffin-Lim loop kernels: kernel stats of tools/bench_griffin_lim.py
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d "$OUT/gl" -- python3 $REPO/tools/bench_griffin_lim.py > "$OUT/gl.log" 2>&1) && echo "griffin-lim pass done"
python3 tools/pmc_summary.py --stats "$OUT/gl" --tag "${TAG}_griffin_lim" --out "$OUT/summary" --cmd "python3 tools/bench_griffin_lim.py" > /dev/null
grep -h '^{' "$OUT/gl.log" | tail -1 > "$OUT/summary/${TAG}_griffin_lim.json" || true
echo "summaries in $OUT/summary (copy into profiles/)"
