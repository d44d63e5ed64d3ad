#!/bin/bash
# Cache-policy experiment for k_scan's step loads / record stores: rebuilds depth_fast.o on the GPU box per variant.
cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wextra -Wno-unused-parameter -Wno-unused-result"
run() {
  touch pollen_amd/csrc/depth_fast.hip
  make -C pollen_amd/csrc CXXFLAGS="$BASE $2" > /tmp/build.log 2>&1 || { echo "$1: build failed"; tail -3 /tmp/build.log; return; }
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['bit_exact_vs_oracle'], d['roofline']['kernels_avg_ms'])"
}
run base ""
run load_nt "-DFGFA_LOAD_POLICY='\" nt\"'"
run load_sc1 "-DFGFA_LOAD_POLICY='\" sc1\"'"
run load_sc0sc1 "-DFGFA_LOAD_POLICY='\" sc0 sc1\"'"
run load_sc0_nt "-DFGFA_LOAD_POLICY='\" sc0 nt\"'"
run load_sc1_nt "-DFGFA_LOAD_POLICY='\" sc1 nt\"'"
run load_nt_store_nt "-DFGFA_NT_STORE -DFGFA_LOAD_POLICY='\" nt\"'"
