#!/bin/bash
# Same-box A/B: 8192-segment windows (FLATGFA_WB=13) against the default 4096 on graphs of a few million segments whose windows
# outnumber the CUs several times over (hap-chr20 / rep-chr20: 977 windows), and the staging threads of the upload.
#   gpurun -- tools/ab_wb13.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for rep in 1 2; do
  for wl in hap-chr20 rep-chr20 cfgL-4Mseg; do
    for wb in "" 13; do
      env ${wb:+FLATGFA_WB=$wb} python3 tools/ab_kernels.py $wl 12 2>/dev/null | tail -1
    done
  done
done
for t in 2 4 8 12 16; do
  echo "upload threads $t:"; FLATGFA_UPLOAD_THREADS=$t python3 tools/upload_probe.py 2>&1 | grep -E "^run|steps: upload" | tail -4
done
