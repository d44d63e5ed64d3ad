#!/usr/bin/env python3
"""Per-kernel HIP-event times of path depth of all paths (a3: `fgfa depth`) beside node depth alone (a2) on cfg-L."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import pollen_amd as pa
from pollen_amd import device as dev
S, P, L = 1_000_000, 1000, 100_000
g = pa.synth(1, S, P, L, sys.argv[1] if len(sys.argv) > 1 else "pangenome", False)
steps, pb, pe, sl = g.soa()
plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, sl))
print(plan.describe())
d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
ln = torch.zeros(P, dtype=torch.int64, device="cuda:0"); ws = torch.zeros_like(ln)
for name, fn in (("a3 path depth", lambda: plan.path_depth_all(d, ln, ws)), ("a2 depth only", lambda: plan.seg_depth(d, None))):
    for _ in range(3): fn()
    plan.status(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20): fn()
    plan.status(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    dev.profile_enable(True); dev.profile_read()
    for _ in range(10): fn()
    plan.status(); dev.profile_enable(False)
    per = {}
    for n, ms in dev.profile_read(): per.setdefault(n, []).append(ms)
    print(name, round(wall, 4), {k: round(float(np.mean(v)), 4) for k, v in per.items()})
