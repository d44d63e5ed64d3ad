#!/bin/bash
# Pass 2 of a tagged call with parts of its work left out (FLATGFA_ACC_SKIP; results are then wrong by construction)
W=${1:-cfgL}
for m in 0; do
  echo -n "skip=$m  "
  FLATGFA_ACC_SKIP=$m FLATGFA_ACC_TIME=1 python3 bench.py --steps 4 --warmup 1 --workload $W --no-cpu-baseline --no-extras --no-verify 2>&1 | grep -E "^k_accum" | tail -1 | cut -c1-175
done
