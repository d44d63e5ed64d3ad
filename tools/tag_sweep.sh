#!/bin/bash
# Same-box sweep of compile-time variants of the depth kernels (each argument one set of -D flags, "" = defaults):
# pass 2's phase times (FLATGFA_ACC_TIME) and the kernels' event times for the workloads in $WLS; --no-verify,
# so ablated builds (FGFA_TAG_ABLATE) can be measured.  Run on the GPU box via gpurun.
cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wextra -Wno-unused-parameter -Wno-unused-result"
for flags in "$@"; do
  touch pollen_amd/csrc/depth_fast_kernels.hpp
  make -C pollen_amd/csrc CXXFLAGS="$BASE $flags" > /tmp/build.log 2>&1 || { echo "[$flags]: build failed"; tail -3 /tmp/build.log; continue; }
  for wl in ${WLS:-cfgL}; do
    FLATGFA_ACC_TIME=1 timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify --workload $wl 2>&1 | grep "^k_accum" | tail -1 | cut -c1-150 | sed "s/^/[$flags] $wl /"
    timeout 300 python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras --no-verify --workload $wl 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$flags] $wl', d['ms_per_step'], d['roofline']['kernels_avg_ms'])"
  done
done
touch pollen_amd/csrc/depth_fast_kernels.hpp; make -C pollen_amd/csrc > /dev/null 2>&1
