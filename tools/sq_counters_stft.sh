#!/bin/bash
# SQ counter passes of the STFT kernels (tools/bench_stft.py --fit: plain stft_wave_kernel and the persistent stft_fit_kernel).
#   gpurun -- 'bash tools/sq_counters_stft.sh TAG'      -> gpurun_out/TAG_stft_sq_counters.txt
set -o pipefail
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd); OUT=$REPO/gpurun_out/sq_stft; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python3 $REPO/tools/bench_stft.py --fit --cpu-clips 0 --steps 3 --warmup 1"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -f csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -f csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INSTS_SMEM -f csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
cd $REPO
(python3 tools/pmc_sq.py $OUT/p1 stft; python3 tools/pmc_sq.py $OUT/p2 stft; python3 tools/pmc_sq.py $OUT/p3 stft) > gpurun_out/${TAG}_stft_sq_counters.txt
cat gpurun_out/${TAG}_stft_sq_counters.txt
