#!/bin/bash
# The chromosome-scale graphs' bench lines alone (step 1b of tools/profile_round.sh):  tools/profile_big.sh r05
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; cd $R
for w in chr-like chr-like-2k hap-16M rep-16M hap-chr cfgL-x16 x16-16Mseg x16-16Mseg-chrom x16-16Mseg-contigs x16-16Mseg-fewlong chr-like-40M; do
  timeout 900 python3 bench.py --steps 10 --warmup 2 --workload $w --no-extras --no-cpu-baseline 2>$OUT/_bench_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
  python3 -c "
import json,sys
d=json.loads(open('$OUT/${TAG}_bench_$w.json').read()); r=d['roofline']
print('$w', d['ms_per_step'], r['kernels_avg_ms'], r['whole_call']['timed_region']['frac'], d['bit_exact_vs_oracle'])"
done
