#!/bin/bash
# Vector-memory pipeline counters for the depth kernels (run via gpurun).  One or two counters
# per pass (the TA/TCP blocks have few counter slots) and a hard timeout per pass: rocprofv3
# hangs after "Request exceeds the capabilities of the hardware".
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_ta; rm -rf $OUT; mkdir -p $OUT; cd $R
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-extras --workload ${1:-cfgL}"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc -- python3 $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed"
done
python3 - <<PY
import sqlite3,glob,re
for d in sorted(glob.glob("$OUT/*/pmc_results.db")):
    try:
        db=sqlite3.connect(d)
        for k,c,v,n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"):
            m=re.search(r"(k_\w+<[^>]*>)",k)
            if m: print(f"{m.group(1):22s} {c:36s} {v:16.1f}")
    except Exception as e:
        print(d, e)
PY
