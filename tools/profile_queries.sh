#!/bin/bash
# The part of tools/profile_round.sh that profiles the queries next to node depth (path depth of all paths, all-pairs overlap),
# on its own:  tools/profile_queries.sh r05   ->  gpurun_out/profiles/<tag>_rocprofv3_kernel_stats_queries.csv, <tag>_pmc_queries.txt
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; cd $R
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 tools/prof_queries.py cfgL 20 > $OUT/_trace4.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats_queries.csv
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $OUT/_pmcq_$set
  rocprofv3 --kernel-trace --pmc $set -d $OUT/_pmcq_$set -o p -- python3 tools/prof_queries.py cfgL 10 > $OUT/_pmcq_$set.log 2>&1
done
python3 - <<PY > $OUT/${TAG}_pmc_queries.txt
import sqlite3,glob,re
def short(k):
    m=re.search(r"(k_\w+)(<[^>]*>)?\(",k)
    if not m: return k[:40]
    n,t=m.group(1),m.group(2) or ""
    if n=="k_accum": return n+("<uniq>" if t.startswith("<true") else ("<depth+paths>" if t.startswith("<false, 12, true") else "<depth>"))
    return n
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), averages per dispatch, KB: tools/prof_queries.py cfgL 10 (path_depth_all x 10, path_overlaps x 10)")
print("# (gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -- MI355X_MICROARCH.md; double it)")
for d in sorted(glob.glob("$OUT/_pmcq_*/**/*.db", recursive=True)):
    db=sqlite3.connect(d)
    for k,c,v,n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name order by kernel_name,counter_name"):
        if 'fgfa_dev' in k:
            print("%-22s %-22s %18.1f  dispatches=%d" % (short(k), c, v, n))
PY
rm -rf $OUT/_trace $OUT/_pmcq_*
cut -c1-160 $OUT/${TAG}_rocprofv3_kernel_stats_queries.csv | head -8; cat $OUT/${TAG}_pmc_queries.txt
