#!/bin/bash
# One gpurun call's worth of checking a kernel change: the GPU test suite, then bench lines of the
# workloads in $WLS (default: the headline and the shapes next to it).  Output under gpurun_out/$1/.
TAG=${1:-quick}; shift
WLS=${WLS:-"cfgL cfgL-chrom chrom-10k cfgL-fewlong cfgL-32k cfgL-4paths cfgL-uniform cfgL-short cfgL-medium cfgL-100kseg cfgL-16Mseg"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -z "$NO_TESTS" ]; then timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log; fi
for w in $WLS; do
  timeout 300 python3 bench.py --steps 40 --warmup 3 --workload $w --no-cpu-baseline --no-extras 2>$OUT/bench_$w.err | tail -1 > $OUT/bench_$w.json
  python3 - $OUT/bench_$w.json $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], d['ms_per_step'], d['roofline']['kernels_avg_ms'], 'exact' if d['bit_exact_vs_oracle'] else 'NOT EXACT')
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
