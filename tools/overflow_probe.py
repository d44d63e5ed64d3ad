#!/usr/bin/env python3
"""Does a plan's bucket capacity settle?  Node depth + status, ten times, printing the plan after every call
(dynamic dealing makes a sub-bucket's fill vary from call to call on graphs whose paths run along them)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import pollen_amd as pa
from pollen_amd import device as dev
S, P, L, model = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, sl = g.soa()
plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, sl))
print(plan.describe(), flush=True)
d = torch.zeros(S, dtype=torch.int32, device="cuda:0"); u = torch.zeros_like(d)
ref = None
for i in range(10):
    plan.seg_depth(d, u)
    plan.status()
    cs = (int(d.to(torch.int64).sum()), int(u.to(torch.int64).sum()))
    print(i, plan.describe()[-40:], cs, flush=True)
    assert ref is None or ref == cs
    ref = cs
