#!/bin/bash
# The chromosome-scale workloads' bench lines at HEAD (part 1b of tools/profile_round.sh): gpurun_out/profiles/<tag>_bench_*.json
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; cd $R
for w in chr-like chr-like-2k hap-16M hap-chr cfgL-x16 x16-16Mseg x16-16Mseg-chrom x16-16Mseg-contigs x16-16Mseg-fewlong chr-like-40M tiny-paths; do
  timeout 900 python3 bench.py --steps 10 --warmup 2 --workload $w --no-extras --no-cpu-baseline 2>$OUT/_bench_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
  python3 -c "
import json
d=json.load(open('$OUT/${TAG}_bench_$w.json')); r=d['roofline']; print('$w', d['ms_per_step'], {k:round(v,3) for k,v in r['kernels_avg_ms'].items()}, d['bit_exact_vs_oracle'], r['plan_choice'][-44:])"
done
