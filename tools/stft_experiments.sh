#!/bin/bash
# STFT ablations + SQ counters on the GPU box (experiments build of libadn.so: ADN_BUILD_EXPERIMENTS=1 must have been
# used for the in-tree library, and is exported here so the digest matches and nothing is rebuilt).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd)
export ADN_BUILD_EXPERIMENTS=1 TMPDIR=/tmp
OUT=$REPO/gpurun_out/stft_exp
mkdir -p $OUT
for ab in 0 1 2 3; do
  ADN_STFT_ABLATE=$ab python3 tools/bench_stft.py --cpu-clips 0 --steps 10 > $OUT/ablate_$ab.json 2>$OUT/ablate_$ab.err && echo "ablate=$ab $(cat $OUT/ablate_$ab.json | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["ms_per_launch"])')"
done
for v in "$@"; do
  ADN_STFT_VARIANT=$v python3 tools/bench_stft.py --cpu-clips 0 --steps 10 > $OUT/variant_$v.json 2>$OUT/variant_$v.err && echo "variant=$v $(cat $OUT/variant_$v.json | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["ms_per_launch"])')"
done
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d $OUT/sq1 -- python3 $REPO/tools/bench_stft.py --cpu-clips 0 --steps 3 --warmup 1 > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -f csv -d $OUT/sq2 -- python3 $REPO/tools/bench_stft.py --cpu-clips 0 --steps 3 --warmup 1 > $OUT/sq2.log 2>&1
cd $REPO
python3 tools/pmc_sq.py $OUT/sq1 stft
python3 tools/pmc_sq.py $OUT/sq2 stft
