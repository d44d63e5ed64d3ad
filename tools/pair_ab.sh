#!/bin/bash
# (FLATGFA_DEBUG_SKIP / FLATGFA_ACC_SKIP / FLATGFA_ACC_PAIR / FLATGFA_ACC_SMALL exist in measurement builds only:
#  tools/variants.sh measure "-DFGFA_MEASURE" here, then FLATGFA_LIB=pollen_amd/lib_measure/libflatgfa.so on the GPU box)
# A/B on one box: pass 2 with one workgroup per window against two (FLATGFA_ACC_PAIR), workloads in $WLS
for w in ${WLS:-cfgL}; do
  for p in 1 0 1 0; do
    FLATGFA_ACC_PAIR=$p python3 bench.py --steps 40 --warmup 3 --workload $w --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pair=$p', '$w', d['ms_per_step'], d['roofline']['kernels_avg_ms'], d['bit_exact_vs_oracle'])"
  done
done
