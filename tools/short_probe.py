#!/usr/bin/env python3
"""Per-kernel times of node depth with and without unique depth on a short-path workload (what the claims of k_scan_short cost)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import pollen_amd as pa
from pollen_amd import device as dev
S, P, L, model = {"short": (1_000_000, 100_000, 1000, "pangenome"), "medium": (1_000_000, 10_000, 10_000, "pangenome"),
                  "chrom1k": (1_000_000, 100_000, 1000, "chromosome")}[sys.argv[1] if len(sys.argv) > 1 else "short"]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, sl = g.soa()
plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, sl))
print(plan.describe())
def quiet_status():  # (ablation builds, FGFA_SHORT_ABLATE, may flag what they break)
    try: plan.status()
    except Exception as ex: print("status:", ex)
d = torch.zeros(S, dtype=torch.int32, device="cuda:0"); u = torch.zeros_like(d)
for with_u in (True, False):
    for _ in range(3): plan.seg_depth(d, u if with_u else None)
    quiet_status(); dev.profile_enable(True); dev.profile_read()
    for _ in range(10): plan.seg_depth(d, u if with_u else None)
    quiet_status(); dev.profile_enable(False)
    per = {}
    for n, ms in dev.profile_read(): per.setdefault(n, []).append(ms)
    print("uniq" if with_u else "depth only", {k: round(float(np.mean(v)), 4) for k, v in per.items()})
