#!/bin/bash
# HBM traffic counters for the depth kernels (run via gpurun):  tools/prof_mem.sh [workload] [steps]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_mem; rm -rf $OUT; mkdir -p $OUT; cd $R
ARGS="bench.py --steps ${2:-5} --warmup 2 --in-flight 1 --no-cpu-baseline --no-verify --no-extras --workload ${1:-cfgL}"
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$set -o pmc -- python3 $ARGS > $OUT/$set.log 2>&1
done
python3 - <<PY
import sqlite3,glob
for d in sorted(glob.glob("$OUT/**/*.db", recursive=True)):
    db=sqlite3.connect(d)
    for k,c,v,n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"):
        if 'fgfa_dev' in k:
            print(f"{k.split('::')[-1][:40]:40s} {c:12s} {v/1024:12.1f} MB n={n}   (FETCH_SIZE: double it for wide reads on gfx950)")
PY
