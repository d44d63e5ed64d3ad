#!/usr/bin/env python3
"""Cycles per phase of k_scan_short's waves (a -DFGFA_SHORT_PROF build: FLATGFA_LIB=pollen_amd/lib_sprof/libflatgfa.so)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pollen_amd as pa
from pollen_amd import device as dev
S = 1_000_000
g = pa.synth(1, S, 100_000, 1000, "pangenome", False)
steps, pb, pe, sl = g.soa()
plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, sl))
print(plan.describe())
d = torch.zeros(S, dtype=torch.int32, device="cuda:0"); u = torch.zeros_like(d)
print("=== timed call")
plan.seg_depth(d, u); plan.status(); torch.cuda.synchronize()
