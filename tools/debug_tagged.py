#!/usr/bin/env python3
"""Diagnostic: node depth of one synthetic graph, tagged against untagged against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pollen_amd as pa
from oracle import flatgfa_oracle as fo
from pollen_amd.device import DepthPlan, DeviceGraph


def run(S, P, L, model, env):
    for k in ("FLATGFA_TAGGED", "FLATGFA_PIECE_STEPS", "FLATGFA_DEPTH_PATH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    g = pa.synth(3, S, P, L, model, False)
    steps, pb, pe, seg_len = g.soa()
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S, seg_len))
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    plan.seg_depth(d, u)
    plan.status()
    pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
    wd, wu = fo.seg_depth_with_uniq(pools)
    gd, gu = d.cpu().numpy().view(np.uint32), u.cpu().numpy().view(np.uint32)
    bd, bu = np.nonzero(gd != wd)[0], np.nonzero(gu != wu)[0]
    print(f"S={S} P={P} L={L} {model} {env}: depth wrong at {len(bd)}, uniq wrong at {len(bu)}")
    for name, bad, got, want in (("depth", bd, gd, wd), ("uniq", bu, gu, wu)):
        if len(bad):
            diff = got[bad].astype(np.int64) - want[bad].astype(np.int64)
            print(f"   {name}: first {bad[:8].tolist()} diff {diff[:8].tolist()}; diff histogram {dict(zip(*np.unique(diff, return_counts=True)))}; "
                  f"windows touched {len(np.unique(bad >> 12))} of {(S + 4095) >> 12}; offsets in window (first 8) {(bad[:8] & 4095).tolist()}")
    plan.close()


if __name__ == "__main__":
    cases = [(300_000, 4, 200_000, "pangenome", {"FLATGFA_PIECE_STEPS": "4096"}),
             (300_000, 4, 200_000, "pangenome", {"FLATGFA_PIECE_STEPS": "4096", "FLATGFA_TAGGED": "0"}),
             (300_000, 4, 200_000, "pangenome", {"FLATGFA_PIECE_STEPS": "4000"}),
             (300_000, 1, 200_000, "pangenome", {"FLATGFA_PIECE_STEPS": "4096"}),
             (300_000, 1, 64_000, "pangenome", {"FLATGFA_PIECE_STEPS": "32768"}),
             (300_000, 40, 100_000, "chromosome", {"FLATGFA_PIECE_STEPS": "32768"}),
             (1_000_000, 4, 25_000_000, "pangenome", {})]
    for c in cases:
        run(*c[:4], dict(c[4], FLATGFA_DEPTH_PATH="bucketed"))
