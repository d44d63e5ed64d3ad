#!/bin/bash
# k_scan's workgroup timeline (FLATGFA_SCAN_TIME) and the kernels' event times for the workloads in $WLS; extra environment in $1
for w in ${WLS:-cfgL}; do
  echo "== $w [$1]"
  env $1 FLATGFA_SCAN_TIME=1 python3 bench.py --steps 6 --warmup 2 --workload $w --no-cpu-baseline --no-extras --no-verify 2>&1 | grep "^k_scan" | tail -1 | cut -c60-
  env $1 python3 bench.py --steps 40 --warmup 3 --workload $w --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
done
