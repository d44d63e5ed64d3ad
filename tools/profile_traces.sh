#!/bin/bash
# The three kernel traces of tools/profile_round.sh alone (steps 2, 2b, 2c):  tools/profile_traces.sh r05
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; cd $R
CMD="bench.py --steps 40 --warmup 3 --in-flight 1 --no-cold"
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 $CMD --no-cpu-baseline --no-extras > $OUT/_trace.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats.csv
grep "^{\"metric" $OUT/_trace.log > $OUT/${TAG}_rocprofv3_kernel_stats.bench_line.json
rm -rf $OUT/_trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 bench.py --steps 40 --warmup 3 --no-cold --no-cpu-baseline --no-extras > $OUT/_trace2.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats_in_flight.csv
grep "^{\"metric" $OUT/_trace2.log > $OUT/${TAG}_rocprofv3_kernel_stats_in_flight.bench_line.json
rm -rf $OUT/_trace; FLATGFA_MALL_MB=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_trace -o t -- python3 $CMD --no-cpu-baseline --no-extras > $OUT/_trace3.log 2>&1
f=$(find $OUT/_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_rocprofv3_kernel_stats_mall0.csv
grep "^{\"metric" $OUT/_trace3.log > $OUT/${TAG}_rocprofv3_kernel_stats_mall0.bench_line.json
rm -rf $OUT/_trace
head -3 $OUT/${TAG}_rocprofv3_kernel_stats.csv $OUT/${TAG}_rocprofv3_kernel_stats_in_flight.csv $OUT/${TAG}_rocprofv3_kernel_stats_mall0.csv | cut -c1-160
