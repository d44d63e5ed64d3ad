#!/bin/bash
# k_scan_short / k_scan_medium with parts switched off (tools/variants.sh sa<N> "-DFGFA_SHORT_ABLATE=<N>" builds them)
for w in short medium; do for t in "" sa1 sa2 sa4 sa8 sa12 sa16; do
  if [ -z "$t" ]; then echo "== $w base"; python3 tools/short_probe.py $w 2>&1 | grep -E "^uniq|^depth"; else
  echo "== $w $t"; FLATGFA_LIB=pollen_amd/lib_$t/libflatgfa.so python3 tools/short_probe.py $w 2>&1 | grep -E "^uniq|^depth"; fi
done; done
