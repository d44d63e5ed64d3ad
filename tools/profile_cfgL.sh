#!/bin/bash
# The headline workload's committed evidence at HEAD (a subset of tools/profile_round.sh): the bench line, the
# rocprofv3 kernel-trace stats of the same command, and the FETCH/WRITE PMC passes.  Output: gpurun_out/profiles/<tag>_*.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; cd $R
timeout 900 python3 bench.py --steps 20 --warmup 5 2>$OUT/_bench_cfgL.err | tail -1 > $OUT/${TAG}_bench_cfgL.json
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/_trace.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats.csv
rm -rf $OUT/_trace
python3 -c "
import json; d=json.load(open('$OUT/${TAG}_bench_cfgL.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['n'], r['samples'], r['kernels_avg_ms']); print(json.dumps(d['extras']['cli_process']))"
head -3 $OUT/${TAG}_rocprofv3_kernel_stats.csv
