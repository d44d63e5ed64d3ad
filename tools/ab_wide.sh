#!/bin/bash
# Same-box A/B of k_scan's emission width (chunks of 64 queue entries emitted side by side: -DFGFA_WIDE=2 / 3 against the default 4),
# warm and with every step from HBM (FLATGFA_MALL_MB=0), three rounds.  Build first: tools/variants.sh w2 "-DFGFA_WIDE=2" w3 "-DFGFA_WIDE=3"
#   gpurun -- tools/ab_wide.sh [workloads...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
WLS=${@:-cfgL cfgL-chrom}
for rep in 1 2 3; do
  for wl in $WLS; do
    for lib in "" pollen_amd/lib_w2/libflatgfa.so pollen_amd/lib_w3/libflatgfa.so; do
      for mall in "" 0; do
        env ${lib:+FLATGFA_LIB=$lib} ${mall:+FLATGFA_MALL_MB=$mall} python3 tools/ab_kernels.py $wl 24 2>/dev/null | tail -1
      done
    done
  done
done
