#!/bin/bash
# One-box sweeps of plan-shaping hooks on the shapes whose pass 2 is dominated by its per-window fixed part (sparse sub-buckets):
# fewer pass-1 workgroups = fewer, fuller sub-buckets per window.    gpurun -- tools/ab_sweep.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for wl in cfgL-16Mseg cfgL-64Mseg; do
  for wgs in "" 192 128 96 64; do
    env ${wgs:+FLATGFA_SCAN_WGS=$wgs} python3 tools/ab_kernels.py $wl 12 2>/dev/null | tail -1
  done
done
for wl in hap-chr20 rep-chr20; do
  for wgs in "" 192 128; do
    env ${wgs:+FLATGFA_SCAN_WGS=$wgs} python3 tools/ab_kernels.py $wl 12 2>/dev/null | tail -1
  done
  FLATGFA_PACKED=0 python3 tools/ab_kernels.py $wl 12 2>/dev/null | tail -1
  FLATGFA_ACC_SLOTS=4 python3 tools/ab_kernels.py $wl 12 2>/dev/null | tail -1
done
