#!/bin/bash
# Profiling passes for k_scan/k_accum on the GPU box (run via gpurun).  Outputs under gpurun_out/prof/.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --workload ${1:-cfgL}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 -L > $OUT/counters.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$tag -o pmc -- python3 $ARGS > $OUT/pmc_$tag.log 2>&1
done
find $OUT -name "*.csv" | head -50
