#!/bin/bash
# The tagged walk that keeps track of its bitsets' owners (FLATGFA_ACC_OWN=1) against the default, same box (profiles/NOTES.md R5.15)
out=gpurun_out/ab_own.txt
: > $out
for wl in "$@"; do
  for own in 0 1; do
    FLATGFA_ACC_OWN=$own timeout 600 python3 tools/ab_kernels.py $wl 6 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-220 >> $out
  done
done
cat $out
