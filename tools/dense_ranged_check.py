#!/usr/bin/env python3
"""k_scan_dense on plans cut into segment ranges (its general, predicated phases: a walk per range skips the steps outside it)
against numpy, on a GPU box: python tools/dense_ranged_check.py"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["FLATGFA_DENSE"] = "1"; os.environ["FLATGFA_RANGE_SEGS"] = "40960"; os.environ["FLATGFA_DEPTH_PATH"] = "bucketed"; os.environ["FLATGFA_BIG_GROUPS"] = "1"
import numpy as np, torch
import pollen_amd as pa
from pollen_amd import device as dev
for (S, P, L, model) in ((300_000, 40, 30_000, "uniform"), (200_000, 300, 5_000, "pangenome"), (123_457, 7, 200_001, "uniform")):
    g = pa.synth(3, S, P, L, model, False)
    steps, pb, pe, sl = g.soa()
    plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, sl))
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0"); u = torch.zeros_like(d)
    plan.seg_depth(d, u); plan.status()
    ids = (steps >> 1).astype(np.int64)
    want_d = np.bincount(ids, minlength=S)
    want_u = np.zeros(S, dtype=np.int64)
    for b, e in zip(pb, pe):
        want_u[np.unique(ids[b:e])] += 1
    ok = (d.cpu().numpy() == want_d).all() and (u.cpu().numpy() == want_u).all()
    print(S, P, L, model, plan.describe()[:90], "OK" if ok else "MISMATCH")
