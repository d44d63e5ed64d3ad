#!/bin/bash
# k_scan emit width (FGFA_WIDE chunks of 64 queue entries side by side): rebuilds depth_fast.o on the
# GPU box per variant, then times the workloads in $WLS (default: cfgL and cfgL-chrom).
cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wextra -Wno-unused-parameter -Wno-unused-result"
run() {
  touch pollen_amd/csrc/depth_fast_kernels.hpp
  make -C pollen_amd/csrc CXXFLAGS="$BASE $2" > /tmp/build.log 2>&1 || { echo "$1: build failed"; tail -3 /tmp/build.log; return; }
  for wl in ${WLS:-cfgL cfgL-chrom}; do timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --workload $wl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $wl', d['ms_per_step'], d['bit_exact_vs_oracle'], d['roofline']['kernels_avg_ms'])"; done
}
for v in ${@:-1 2 4 1}; do run "wide${v%%:*}_slots${v##*:}" "-DFGFA_WIDE=${v%%:*} -DFGFA_SLOTS_MODE=${v##*:}"; done  # args: width:mode
