#!/bin/bash
# Cache-policy experiments (rebuilds the kernels' objects on the GPU box per variant).
cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wextra -Wno-unused-parameter -Wno-unused-result"
run() {
  touch pollen_amd/csrc/depth_fast_kernels.hpp
  make -C pollen_amd/csrc CXXFLAGS="$BASE $2" > /tmp/build.log 2>&1 || { echo "$1: build failed"; tail -3 /tmp/build.log; return; }
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['bit_exact_vs_oracle'], d['roofline']['kernels_avg_ms'])"
}
run base ""
run rec_nt "-DFGFA_REC_POLICY='\" nt\"'"
run rec_sc1 "-DFGFA_REC_POLICY='\" sc1\"'"
run base2 ""
