#!/bin/bash
# Ablation sweep of k_scan on cfg-L (diagnostics; results with a skip mask are not valid depths).
# mask bits: 1 no record stores, 2 no bitset scan-out, 4 no bitset ORs, 8 no tile processing
for m in 0 1 2 4 6 7 8 10 15; do
  echo -n "skip=$m "
  FLATGFA_DEBUG_SKIP=$m python bench.py --steps 10 --warmup 2 --workload ${1:-cfgL} --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
done
