#!/bin/bash
# Produces the committed profile summaries for one round (run via gpurun from the repo root):
#   tools/profile_round.sh r01
# Writes gpurun_out/profiles/<tag>_*; copy them into profiles/ afterwards.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; cd $R
# (the kernels' own durations: one call in flight -- bench.py's timed region keeps two, whose kernels share the chip;
#  its per-kernel samples, like these traces, are taken one call after the other)
CMD="bench.py --steps 40 --warmup 3 --in-flight 1 --no-cold"  # (--no-cold: roofline.frac_cold's plan has a trace of its own, 2c)
# 1. the bench line itself (all workloads; cfgL is the headline)
for w in cfgL cfgL-uniform cfgL-chrom cfgL-short cfgL-fewlong cfgL-4paths cfgL-medium cfgL-32k chrom-10k chrom-1k tiny-paths hap-1k hap-10k hap-100 hap-chr20 rep-chr20 cfgL-100kseg cfgL-4Mseg cfgL-16Mseg cfgL-64Mseg cfgM cfgS; do
  timeout 600 python3 bench.py --steps 40 --warmup 3 --workload $w 2>$OUT/_bench_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
  # (a workload whose secondary measurements do not apply -- a million tiny paths -- still gets its line)
  [ -s $OUT/${TAG}_bench_$w.json ] || timeout 600 python3 bench.py --steps 40 --warmup 3 --workload $w --no-extras 2>>$OUT/_bench_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
done
# 1b. chromosome-scale graphs (0.9 - 1.8 G steps; the secondary measurements are cfg-L's business)
for w in chr-like chr-like-2k hap-16M rep-16M hap-chr cfgL-x16 x16-16Mseg x16-16Mseg-chrom x16-16Mseg-contigs x16-16Mseg-fewlong chr-like-40M; do
  timeout 900 python3 bench.py --steps 10 --warmup 2 --workload $w --no-extras --no-cpu-baseline 2>$OUT/_bench_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
done
# 2. kernel trace + stats of the same command (csv)
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 $CMD --no-cpu-baseline --no-extras > $OUT/_trace.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats.csv
# 2b. ... with calls in flight (the default: three lanes of the pipeline), as the timed region runs (the kernels of consecutive calls overlap: their durations are longer, the calls shorter)
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 bench.py --steps 40 --warmup 3 --no-cold --no-cpu-baseline --no-extras > $OUT/_trace2.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats_in_flight.csv
# 2c. ... with no step kept in the Infinity Cache (FLATGFA_MALL_MB=0: what roofline.frac_cold is measured on)
rm -rf $OUT/_trace; FLATGFA_MALL_MB=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 $CMD --no-cpu-baseline --no-extras > $OUT/_trace3.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats_mall0.csv
# 2d. the queries next to node depth: path depth of all paths (a3) and all-pairs overlap (config 5)
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 tools/prof_queries.py cfgL 20 > $OUT/_trace4.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats_queries.csv
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $OUT/_pmcq_$set
  rocprofv3 --kernel-trace --pmc $set -d $OUT/_pmcq_$set -o p -- python3 tools/prof_queries.py cfgL 10 > $OUT/_pmcq_$set.log 2>&1
done
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $OUT/_pmcm_$set
  FLATGFA_MALL_MB=0 rocprofv3 --kernel-trace --pmc $set -d $OUT/_pmcm_$set -o p -- python3 $CMD --no-cpu-baseline --no-verify --no-extras > $OUT/_pmcm_$set.log 2>&1
done
python3 - <<PY > $OUT/${TAG}_pmc_queries_and_mall0.txt
import sqlite3,glob,re
def short(k):
    m=re.search(r"(k_\w+)(<[^>]*>)?\(",k)
    if not m: return k[:40]
    n,t=m.group(1),m.group(2) or ""
    if n=="k_accum": return n+("<uniq>" if t.startswith("<true") else ("<depth+paths>" if t.startswith("<false, 12, true") else "<depth>"))
    if n=="k_scan": return n
    return n
for title,pat in (("tools/prof_queries.py cfgL 10 (path_depth_all x 10, path_overlaps x 10)","$OUT/_pmcq_*/**/*.db"),("FLATGFA_MALL_MB=0 python3 $CMD (every step from HBM)","$OUT/_pmcm_*/**/*.db")):
    print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), averages per dispatch, KB: %s" % title)
    print("# (gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -- MI355X_MICROARCH.md; double it)")
    for d in sorted(glob.glob(pat, recursive=True)):
        db=sqlite3.connect(d)
        for k,c,v,n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name order by kernel_name,counter_name"):
            if 'fgfa_dev' in k:
                print("%-22s %-22s %18.1f  dispatches=%d" % (short(k), c, v, n))
PY
rm -rf $OUT/_pmcq_* $OUT/_pmcm_*
# 3. PMC passes (separate runs; --kernel-trace only, as gpurun requires): the headline workload, and the shapes next to it
for wl in cfgL cfgL-chrom cfgL-short; do
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/_pmc_$tag
  rocprofv3 --kernel-trace --pmc $set -d $OUT/_pmc_$tag -o p -- python3 $CMD --workload $wl --no-cpu-baseline --no-verify --no-extras > $OUT/_pmc_$tag.log 2>&1
done
[ $wl = cfgL ] && SUMMARY=$OUT/${TAG}_pmc_summary.txt || SUMMARY=$OUT/${TAG}_pmc_$wl.txt
WL=$wl python3 - <<PY > $SUMMARY
import sqlite3,glob,re,os
def short(k):
    m=re.search(r"(k_\w+)(<[^>]*>)?\(",k)
    if not m: return k[:40]
    n,t=m.group(1),m.group(2) or ""
    if n=="k_accum": return n+("<uniq>" if t.startswith("<true") else "<depth>")
    if n=="k_scan": return n
    return n+t
print("# rocprofv3 --pmc, averages per dispatch over the bench run (python3 $CMD --workload %s); FETCH_SIZE/WRITE_SIZE in KB" % os.environ["WL"])
print("# (gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -- MI355X_MICROARCH.md; double it)")
for d in sorted(glob.glob("$OUT/_pmc_*/**/*.db", recursive=True)):
    db=sqlite3.connect(d)
    for k,c,v,n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name order by kernel_name,counter_name"):
        if 'fgfa_dev' in k:
            print("%-22s %-22s %18.1f  dispatches=%d" % (short(k), c, v, n))
PY
[ $wl = cfgL ] && for t in FETCH_SIZE WRITE_SIZE; do rm -rf $OUT/_keep_$t; cp -r $OUT/_pmc_$t $OUT/_keep_$t; done
done
for t in FETCH_SIZE WRITE_SIZE; do rm -rf $OUT/_pmc_$t; mv $OUT/_keep_$t $OUT/_pmc_$t; done
# 4. per-kernel HBM traffic (what bench.py quotes as roofline.traffic)
python3 - <<PY > $OUT/latest_traffic.json
import sqlite3,glob,json,re
vals={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for d in glob.glob("$OUT/_pmc_%s/**/*.db" % c, recursive=True):
        db=sqlite3.connect(d)
        for k,v in db.execute("select kernel_name,avg(value) from counters_collection where counter_name=? group by kernel_name",(c,)):
            if 'fgfa_dev' not in k: continue
            m=re.search(r"(k_\w+)(<[^>]*>)?\(",k)
            if not m: continue
            name=m.group(1)
            if name=="k_accum": name+="<uniq>" if (m.group(2) or "").startswith("<true") else "<depth>"
            vals.setdefault(name,{})[c]=v
import subprocess
commit=subprocess.run(["git","-C","$R","rev-parse","--short","HEAD"],capture_output=True,text=True).stdout.strip() or None
if not commit:  # (a GPU box has a snapshot without .git/: the commit `make` stamped next to the library)
    try: commit=open("$R/pollen_amd/lib/HEAD").read().strip() or None
    except OSError: pass
out={"commit":commit,"source":"profiles/${TAG}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; KB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 note)","workload":"cfgL","kernels":{}}
for n,v in vals.items():
    f=v.get("FETCH_SIZE",0)*1024*2; w=v.get("WRITE_SIZE",0)*1024
    out["kernels"][n]={"fetch_bytes":f,"write_bytes":w,"hbm_bytes":f+w}
print(json.dumps(out,indent=1))
PY
rm -rf $OUT/_trace $OUT/_pmc_* ; for f in $OUT/_bench_*.err; do [ -s $f ] && { echo "== $f"; tail -3 $f; }; done; ls -la $OUT
