#!/bin/bash
# Bench lines of the named workloads (as tools/profile_round.sh makes them):  tools/profile_some.sh r05 small|big wl...
TAG=$1; KIND=$2; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; cd $R
for w in "$@"; do
  if [ $KIND = big ]; then
    timeout 900 python3 bench.py --steps 10 --warmup 2 --workload $w --no-extras --no-cpu-baseline 2>$OUT/_bench_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
  else
    timeout 600 python3 bench.py --steps 40 --warmup 3 --workload $w 2>$OUT/_bench_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
    [ -s $OUT/${TAG}_bench_$w.json ] || timeout 600 python3 bench.py --steps 40 --warmup 3 --workload $w --no-extras 2>>$OUT/_bench_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
  fi
  python3 -c "
import json
d=json.loads(open('$OUT/${TAG}_bench_$w.json').read()); r=d['roofline']
print('$w', d['ms_per_step'], r['kernels_avg_ms'], d['bit_exact_vs_oracle'])"
done
