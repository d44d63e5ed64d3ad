#!/bin/bash
# Same-box A/B of compile-time variants of the depth kernels (depth_scan.hip, depth_scan_paths.hip, depth_accum.hip): every argument is one set of -D flags
# (quote it; "" = the defaults); the workloads are those in $WLS.  Run on the GPU box via gpurun.
cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wextra -Wno-unused-parameter -Wno-unused-result"
for rep in $(seq ${REPS:-2}); do
  for flags in "$@"; do
    touch pollen_amd/csrc/depth_fast_kernels.hpp
    make -C pollen_amd/csrc CXXFLAGS="$BASE $flags" > /tmp/build.log 2>&1 || { echo "[$flags]: build failed"; tail -3 /tmp/build.log; continue; }
    for wl in ${WLS:-cfgL}; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --workload $wl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$flags] $wl', d['ms_per_step'], d['bit_exact_vs_oracle'], d['roofline']['kernels_avg_ms'])"; done
  done
done
touch pollen_amd/csrc/depth_fast_kernels.hpp; make -C pollen_amd/csrc > /dev/null 2>&1
