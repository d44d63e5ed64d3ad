#!/bin/bash
# Per-block no-claim marks off / on, same box (profiles/NOTES.md R5.17)
out=gpurun_out/ab_blocks.txt
: > $out
for wl in "$@"; do
  FLATGFA_NO_CLAIM_BLOCKS_OFF=1 timeout 600 python3 tools/ab_kernels.py $wl 8 2>&1 | tail -1 | cut -c1-230 >> $out
  timeout 600 python3 tools/ab_kernels.py $wl 8 2>&1 | tail -1 | cut -c1-230 >> $out
done
cat $out
