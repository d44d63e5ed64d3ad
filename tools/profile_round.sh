#!/bin/bash
# Produces the committed profile summaries for one round (run via gpurun from the repo root):
#   tools/profile_round.sh r01
# Writes gpurun_out/profiles/<tag>_*; copy them into profiles/ afterwards.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; cd $R
CMD="bench.py --steps 20 --warmup 3"
# 1. the bench line itself (all workloads; cfgL is the headline)
for w in cfgL cfgL-uniform cfgL-short cfgS; do
  python3 bench.py --steps 20 --warmup 3 --workload $w 2>/dev/null | tail -1 > $OUT/${TAG}_bench_$w.json
done
# 2. kernel trace + stats of the same command (csv)
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 $CMD --no-cpu-baseline > $OUT/_trace.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats.csv
# 3. PMC passes (separate runs; --kernel-trace only, as gpurun requires)
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/_pmc_$tag
  rocprofv3 --kernel-trace --pmc $set -d $OUT/_pmc_$tag -o p -- python3 $CMD --no-cpu-baseline --no-verify > $OUT/_pmc_$tag.log 2>&1
done
python3 - <<PY > $OUT/${TAG}_pmc_summary.txt
import sqlite3,glob
print("# rocprofv3 --pmc, averages per dispatch over the bench run (python3 $CMD); FETCH_SIZE/WRITE_SIZE in KB")
print("# (gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -- MI355X_MICROARCH.md; double it)")
for d in sorted(glob.glob("$OUT/_pmc_*/**/*.db", recursive=True)):
    db=sqlite3.connect(d)
    for k,c,v,n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name order by kernel_name,counter_name"):
        if 'fgfa_dev' in k:
            print("%-22s %-22s %18.1f  dispatches=%d" % (k.split("(")[0].split("::")[-1], c, v, n))
PY
rm -rf $OUT/_trace $OUT/_pmc_* ; ls -la $OUT
