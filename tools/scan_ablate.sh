#!/bin/bash
# (FLATGFA_DEBUG_SKIP / FLATGFA_ACC_SKIP / FLATGFA_ACC_PAIR / FLATGFA_ACC_SMALL exist in measurement builds only:
#  tools/variants.sh measure "-DFGFA_MEASURE" here, then FLATGFA_LIB=pollen_amd/lib_measure/libflatgfa.so on the GPU box)
# k_scan with parts switched off (FLATGFA_DEBUG_SKIP: 1 no record stores, 2 no emission, 4 no pass B, 8 no tiles; the
# diagnostic build of the kernel, results are wrong): what a workload's pass 1 spends where.   tools/scan_ablate.sh cfgL-chrom
W=${1:-cfgL-chrom}
for d in "" 1 2 6 8; do
  echo "== $W FLATGFA_DEBUG_SKIP=$d"
  FLATGFA_DEBUG_SKIP=$d python3 - "$W" <<'PY' 2>&1 | grep -E "^uniq|status" | cut -c1-200
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pollen_amd as pa
from pollen_amd import device as dev
import bench
w = bench.WORKLOADS[sys.argv[1]]
S, P, L, model = w[0], w[1], w[2], w[3]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, sl = g.soa()
os.environ.setdefault("FLATGFA_DEPTH_PATH", "bucketed")
plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, sl))
d = torch.zeros(S, dtype=torch.int32, device="cuda:0"); u = torch.zeros_like(d)
def quiet():
    try: plan.status()
    except Exception as ex: print("status:", str(ex)[:80])
for _ in range(3): plan.seg_depth(d, u)
quiet(); dev.profile_enable(True); dev.profile_read()
for _ in range(10): plan.seg_depth(d, u)
quiet(); dev.profile_enable(False)
per = {}
for n, ms in dev.profile_read(): per.setdefault(n, []).append(ms)
print("uniq", {k: round(float(np.mean(v)), 4) for k, v in per.items()}, plan.describe()[:60])
PY
done
