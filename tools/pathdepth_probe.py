#!/usr/bin/env python3
"""Per-kernel times of path depth of all paths (what `fgfa depth` computes) on cfg-L."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import pollen_amd as pa
from pollen_amd import device as dev
S, P, L = 1_000_000, 1000, 100_000
g = pa.synth(1, S, P, L, "pangenome", True)
steps, pb, pe, sl = g.soa()
plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, sl))
print(plan.describe())
d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
ln = torch.zeros(P, dtype=torch.int64, device="cuda:0"); wt = torch.zeros(P, dtype=torch.int64, device="cuda:0")
for _ in range(3): plan.path_depth_all(d, ln, wt)
plan.status(); dev.profile_enable(True); dev.profile_read()
for _ in range(10): plan.path_depth_all(d, ln, wt)
plan.status(); dev.profile_enable(False)
per = {}
for n, ms in dev.profile_read(): per.setdefault(n, []).append(ms)
print({k: (round(float(np.mean(v)), 4), len(v)) for k, v in per.items()})
