#!/bin/bash
# Same-box A/B of the library as built (pollen_amd/lib) against the build kept in pollen_amd/lib_prev, three rounds, warm and with every
# step from HBM:   gpurun -- tools/ab_prev.sh [workloads...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
WLS=${@:-cfgL cfgL-chrom}
for rep in 1 2 3; do
  for wl in $WLS; do
    for lib in pollen_amd/lib_prev/libflatgfa.so ""; do
      for mall in "" 0; do
        env ${lib:+FLATGFA_LIB=$lib} ${mall:+FLATGFA_MALL_MB=$mall} python3 tools/ab_kernels.py $wl 24 2>/dev/null | tail -1
      done
    done
  done
done
