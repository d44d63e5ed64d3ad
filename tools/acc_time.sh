#!/bin/bash
# Where pass 2 spends its time (FLATGFA_ACC_TIME), tagged and untagged, for the workloads in $WLS.
WLS=${WLS:-"cfgL cfgL-chrom"}
for w in $WLS; do
  for t in 1 0; do
    echo "== $w tagged=$t"
    FLATGFA_TAGGED=$t FLATGFA_ACC_TIME=1 python3 bench.py --steps 4 --warmup 1 --workload $w --no-cpu-baseline --no-extras --no-verify 2>&1 | grep -E "^k_accum" | tail -2
  done
done
