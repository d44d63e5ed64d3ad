#!/bin/bash
# Instruction-mix counters for the depth kernels (run via gpurun):  tools/prof_insts.sh [workload] [steps]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_insts; rm -rf $OUT; mkdir -p $OUT; cd $R
ARGS="bench.py --steps ${2:-5} --warmup 2 --in-flight 1 --no-cpu-baseline --no-verify --no-extras --workload ${1:-cfgL}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/a -o pmc -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/b -o pmc -- python3 $ARGS > $OUT/b.log 2>&1
python3 - <<PY
import sqlite3,glob
for d in sorted(glob.glob("$OUT/**/*.db", recursive=True)):
    db=sqlite3.connect(d)
    for k,c,v,n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"):
        if 'k_scan' in k or 'k_accum' in k or 'k_path' in k:
            print(f"{k.split('::')[-1][:28]:28s} {c:22s} {v:16.1f} n={n}")
PY
