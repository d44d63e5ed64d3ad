#!/usr/bin/env python3
"""Repeat the same query many times and compare every result with the first, on the device:
an intermittent race in the kernels would show up as a mismatch.  Usage: stress_repeat.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import pollen_amd as pa  # noqa: E402
from pollen_amd import device as dev  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for name, (S, P, L, model) in {"cfgL": (1_000_000, 1000, 100_000, "pangenome"), "short": (1_000_000, 50_000, 1000, "pangenome"),
                               "mixed-4M": (4_000_000, 3000, 20_000, "pangenome"), "uniform": (300_000, 200, 50_000, "uniform"),
                               "chromosome": (2_000_000, 300, 150_000, "chromosome"), "tiny": (500_000, 300_000, 90, "pangenome"),
                               "32k (eight bitsets per wave)": (1_000_000, 2000, 32_000, "pangenome"),
                               "16M haplotypes (packed buckets)": (16_000_000, 4000, 100_000, "haplotype"),
                               "contigs on 8 M segments (sparse tags: the owner-tracking walk)": (8_000_000, 40_000, 10_000, "chromosome"),
                               "in-order short contigs": (1_000_000, 60_000, 1000, "haplotype")}.items():
    g = pa.synth(9, S, P, L, model, False)
    steps, pb, pe, seg_len = g.soa()
    plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0"))
    ref = torch.zeros(2 * S, dtype=torch.int32, device="cuda:0")
    out = torch.zeros(2 * S, dtype=torch.int32, device="cuda:0")
    plan.seg_depth(ref[:S], ref[S:])
    plan.status()
    n_bad = 0
    for i in range(reps):
        out.fill_(-1)
        plan.seg_depth(out[:S], out[S:])
        if not torch.equal(out, ref):
            n_bad += 1
    plan.status()
    print(f"{name}: {reps} repeats, {n_bad} differ from the first  [{plan.describe()[:160]}]", flush=True)
    bad += n_bad
sys.exit(1 if bad else 0)
