#!/bin/bash
# (FLATGFA_DEBUG_SKIP / FLATGFA_ACC_SKIP / FLATGFA_ACC_PAIR / FLATGFA_ACC_SMALL exist in measurement builds only:
#  tools/variants.sh measure "-DFGFA_MEASURE" here, then FLATGFA_LIB=pollen_amd/lib_measure/libflatgfa.so on the GPU box)
# Ablation sweep of k_scan on cfg-L (diagnostics; results with a skip mask are not valid depths).
# mask bits: 1 no record stores, 4 no bitset ORs, 8 no block processing (loads only), 16 loads from a cache-resident megabyte
for m in 0 16 17 21 24; do
  echo -n "skip=$m "
  FLATGFA_DEBUG_SKIP=$m python bench.py --steps 10 --warmup 2 --workload ${1:-cfgL} --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
done
