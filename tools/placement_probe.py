#!/usr/bin/env python3
"""Does k_scan's time depend on WHERE its buffers lie?  One process, the cfg-L graph: several copies of the step
array (each a fresh device allocation, the earlier ones kept so that the next lands elsewhere) x several plans
(each with fresh scratch), k_scan / k_accum by HIP events over 12 calls each.  Usage: python tools/placement_probe.py [copies] [plans]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pollen_amd as pa  # noqa: E402
from pollen_amd import device as dev  # noqa: E402

n_copies = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_plans = int(sys.argv[2]) if len(sys.argv) > 2 else 2
S, P, L = 1_000_000, 1000, 100_000
g = pa.synth(1, S, P, L, "pangenome", False)
steps, pb, pe, seg_len = g.soa()
d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
keep = []
for c in range(n_copies):
    graph = dev.DeviceGraph(steps, pb, pe, S, seg_len)
    keep.append(graph)
    for q in range(n_plans):
        plan = dev.DepthPlan(graph)
        keep.append(plan)
        for _ in range(3):
            plan.seg_depth(d, u)
        plan.status()
        dev.profile_enable(True)
        dev.profile_read()
        for _ in range(12):
            plan.seg_depth(d, u)
        plan.status()
        dev.profile_enable(False)
        per = {}
        for name, ms in dev.profile_read():
            per.setdefault(name, []).append(ms)
        print(f"steps copy {c} at 0x{graph.steps.data_ptr():x}  plan {q}: " + "  ".join(f"{k} {np.median(v) * 1e3:.1f} us" for k, v in per.items()), flush=True)
    if c % 2 == 1:
        # a spacer, so that the next copy does not land right behind this one
        keep.append(torch.empty(int(37e6) * (c + 1), dtype=torch.int32, device="cuda:0"))
