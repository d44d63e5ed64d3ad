#!/bin/bash
# 4096-segment windows on graphs of 16 M segments (the default there: 8192), same box (profiles/NOTES.md R5.12)
out=gpurun_out/ab_wb12.txt
: > $out
for wl in x16-16Mseg x16-16Mseg-chrom x16-16Mseg-contigs x16-16Mseg-fewlong chr-like chr-like-2k hap-16M hap-chr cfgL-16Mseg; do
  timeout 600 python3 tools/ab_kernels.py $wl 6 2>&1 | grep -v amdgpu.ids >> $out
  FLATGFA_WB=12 timeout 600 python3 tools/ab_kernels.py $wl 6 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
