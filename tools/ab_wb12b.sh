#!/bin/bash
out=gpurun_out/ab_wb12b.txt
: > $out
for wl in x16-16Mseg x16-16Mseg-chrom chr-like-2k x16-16Mseg-contigs; do
  FLATGFA_LIB=pollen_amd/lib_head/libflatgfa.so timeout 600 python3 tools/ab_kernels.py $wl 6 2>&1 | grep -v amdgpu.ids >> $out
  timeout 600 python3 tools/ab_kernels.py $wl 6 2>&1 | grep -v amdgpu.ids >> $out
done
for wl in x16-16Mseg x16-16Mseg-chrom; do python3 tools/pipeline_probe.py $wl 30 2>&1 | grep -E "plan:|in flight|identical|exact" >> $out; done
cat $out
