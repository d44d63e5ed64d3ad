#!/bin/bash
# HBM traffic and LDS/instruction counters for the depth kernels (run via gpurun): one counter set per pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_mem; rm -rf $OUT; mkdir -p $OUT; cd $R
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-extras --workload ${1:-cfgL}"
[ -n "$PROF_PROG" ] && ARGS="$PROF_PROG"   # e.g. PROF_PROG="tools/linear_paths.py 1000000 200 0.5 0"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc -- python3 $ARGS > $OUT/p$i.log 2>&1
done
python3 - <<PY
import sqlite3,glob,re
for d in sorted(glob.glob("$OUT/*/pmc_results.db")):
    db=sqlite3.connect(d)
    for k,c,v,n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"):
        if 'k_scan' in k or 'k_accum' in k:
            name = re.search(r"k_(scan|accum)\w*(<[^>]*>)?", k).group(0)
            print(f"{name:24s} {c:22s} {v:16.1f}")
PY
