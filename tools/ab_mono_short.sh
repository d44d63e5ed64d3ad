#!/bin/bash
# Same-box A/B for the no-claim paths of the wave-per-path kernels (profiles/NOTES.md R5.11):
#   HEAD's library against this one on lists without such paths, FLATGFA_NO_CLAIM=0 against the default on lists of them.
out=gpurun_out/ab_mono_short.txt
: > $out
for rep in 1 2; do
  for wl in cfgL-short cfgL-medium chrom-1k tiny-paths; do
    FLATGFA_LIB=pollen_amd/lib_head/libflatgfa.so python3 tools/ab_kernels.py $wl 16 >> $out 2>&1
    python3 tools/ab_kernels.py $wl 16 >> $out 2>&1
  done
  for wl in hap-1k hap-10k hap-100; do
    FLATGFA_NO_CLAIM=0 python3 tools/ab_kernels.py $wl 16 >> $out 2>&1
    python3 tools/ab_kernels.py $wl 16 >> $out 2>&1
  done
done
cat $out
