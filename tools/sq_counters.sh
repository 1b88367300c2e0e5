#!/bin/bash
# SQ counter passes of bench.py's forward (no sub-benchmarks): where the waves of each kernel spend their cycles.
#   gpurun -- 'bash tools/sq_counters.sh [kernel-name filter]'
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd); OUT=$REPO/gpurun_out/sq; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -f csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -f csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -f csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
cd $REPO
python3 tools/pmc_sq.py $OUT/p1 "${1:-wino}"; python3 tools/pmc_sq.py $OUT/p2 "${1:-wino}"; python3 tools/pmc_sq.py $OUT/p3 "${1:-wino}"
