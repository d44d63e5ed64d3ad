#!/bin/bash
# k_scan's launch-to-launch spread on one box: per-dispatch durations from a
# kernel trace, the same joined with GRBM_GUI_ACTIVE (duration vs cycles: clock or work?), and the
# per-workgroup timeline (FLATGFA_SCAN_TIME=<file>: XCC, items taken, start/end).  Output: gpurun_out/spread/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/spread; mkdir -p $OUT; cd $R
CMD="bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-extras --no-verify"
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --output-format csv -d $OUT/_trace -o t -- python3 $CMD > $OUT/trace.log 2>&1
f=$(find $OUT/_trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_trace.csv
rm -rf $OUT/_pmc; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/_pmc -o p -- python3 $CMD > $OUT/pmc.log 2>&1
python3 - <<PY > $OUT/pmc_per_dispatch.csv
import sqlite3,glob
print("kernel,dispatch_id,start_ns,end_ns,duration_ns,GRBM_GUI_ACTIVE")
for d in sorted(glob.glob("$OUT/_pmc/**/*.db", recursive=True)):
    db=sqlite3.connect(d)
    tabs=[r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        q="select c.kernel_name,c.dispatch_id,k.start,k.end,k.end-k.start,c.value from counters_collection c join kernels k on k.dispatch_id=c.dispatch_id where c.counter_name='GRBM_GUI_ACTIVE' and c.kernel_name like '%k_scan%' order by c.dispatch_id"
        for r in db.execute(q): print(",".join(str(x) for x in r))
    except Exception as e:
        print("# join failed:", e, tabs)
        for r in db.execute("select kernel_name,dispatch_id,value from counters_collection where counter_name='GRBM_GUI_ACTIVE' and kernel_name like '%k_scan%' order by dispatch_id"):
            print(",".join(str(x) for x in r))
PY
rm -f $OUT/wg_timeline.csv
FLATGFA_SCAN_TIME=$OUT/wg_timeline.csv python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras --no-verify > $OUT/timeline_bench.json 2> $OUT/timeline.log
rm -rf $OUT/_trace $OUT/_pmc
ls -la $OUT; head -3 $OUT/kernel_trace.csv; head -5 $OUT/pmc_per_dispatch.csv; tail -2 $OUT/timeline.log
