#!/bin/bash
# Same-box A/B of environment settings (and FLATGFA_LIB variants): tools/gpu_ab.sh "<workloads>" "<env A>" "<env B>" ... ; two rounds
WLS=$1; shift
O=gpurun_out/ab; mkdir -p $O
for r in 1 2; do for w in $WLS; do i=0; for e in "$@"; do i=$((i+1))
  env $e python3 bench.py --steps 40 --warmup 3 --workload $w --no-cpu-baseline --no-extras 2>$O/err_$i.txt | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('[$e]', '$w', d['ms_per_step'], r['kernels_avg_ms'], 'exact' if d['bit_exact_vs_oracle'] else 'NOT EXACT')
except Exception as ex: print('[$e] $w FAILED', ex); print(open('$O/err_$i.txt').read()[-400:])"
done; done; done
