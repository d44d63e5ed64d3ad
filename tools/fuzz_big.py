#!/usr/bin/env python3
"""Differential check on graphs of the size and shape of chromosome graphs (run on a GPU box): millions
of segments, a few paths of millions of steps, some of hundreds of thousands and many short contigs in
one graph, all walking along the segment ids (some downwards).  These are the shapes that take the plan
through its less travelled branches -- split paths with shared bitsets on 4096-segment windows, groups
of paths, bucket arrays beyond 2^30 records, several segment ranges.  Node depth, unique depth and the
path depth of all paths against the C oracle.
Usage: python tools/fuzz_big.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pollen_amd as pa  # noqa: E402
from oracle import flatgfa_oracle as fo  # noqa: E402


def walk(rng, S, L, down):
    u = rng.integers(0, 100, size=L, dtype=np.int16)
    k = rng.integers(0, 64, size=L, dtype=np.int16)
    inc = np.where(u < 70, 1, np.where(u < 95, 2 + (k & 3), np.where(u < 99, 8 + k, 0))).astype(np.int64)
    jump = u >= 99
    jump[0] = True
    base = np.where(jump, rng.integers(0, S, size=L), 0).astype(np.int64)
    inc[jump] = 0
    if down:
        inc = -inc
    c = np.cumsum(inc)
    last = np.maximum.accumulate(np.where(jump, np.arange(L), 0))
    ids = (base[last] + c - c[last]) % S
    return ((ids << 1) | rng.integers(0, 2, size=L)).astype(np.uint32)


def main():
    import torch
    from pollen_amd import device as dev
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        S = int(rng.choice([3_000_000, 5_000_000, 9_000_000, 17_000_000, 35_000_000]))
        budget = int(rng.choice([60_000_000, 150_000_000, 300_000_000]))
        lens = []
        n_huge = int(rng.integers(0, 30))
        for _ in range(n_huge):
            lens.append(int(rng.integers(1_000_000, 12_000_000)))
        while sum(lens) < budget:
            r = rng.integers(0, 10)
            lens.append(int(rng.integers(200, 3000)) if r < 5 else int(rng.integers(3000, 60_000)) if r < 8 else int(rng.integers(60_000, 900_000)))
        rng.shuffle(lens)
        t0 = time.time()
        parts = [walk(rng, S, L, rng.integers(0, 4) == 0) for L in lens]
        steps = np.concatenate(parts)
        del parts
        pe = np.cumsum(np.array(lens, dtype=np.uint64)).astype(np.uint32)
        pb = (pe - np.array(lens, dtype=np.uint32)).astype(np.uint32)
        P = len(lens)
        seg_len = rng.integers(1, 30, size=S).astype(np.uint32)
        paths = np.zeros(P, dtype=fo.PATH_DT)
        paths["steps_start"], paths["steps_end"] = pb, pe
        segs = np.zeros(S, dtype=fo.SEG_DT)
        segs["seq_end"] = seg_len
        pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
        pools.paths, pools.steps, pools.segs = paths, steps, segs
        want_d, want_u = fo.seg_depth_with_uniq(pools)
        want_ln, want_mean = fo.path_depth(pools, np.arange(P, dtype=np.uint32))
        t1 = time.time()
        plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0"))
        d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        d2 = torch.full((S,), -3, dtype=torch.int32, device="cuda:0")
        ln = torch.full((P,), 7, dtype=torch.int64, device="cuda:0")
        ws = torch.full((P,), -1, dtype=torch.int64, device="cuda:0")
        ok = True
        for rep in range(3):  # (k_scan deals its items out anew every time)
            plan.seg_depth(d, u)
            plan.status()
            ok = ok and bool((d.cpu().numpy().view(np.uint32) == want_d).all()) and bool((u.cpu().numpy().view(np.uint32) == want_u).all())
        plan.path_depth_all(d2, ln, ws)
        plan.status()
        got_ln = ln.cpu().numpy().view(np.uint64)
        got_mean = ws.cpu().numpy().view(np.uint64).astype(np.float64) / got_ln.astype(np.float64)
        ok = ok and bool((d2.cpu().numpy().view(np.uint32) == want_d).all()) and bool((got_ln == want_ln).all()) and got_mean.tobytes() == want_mean.tobytes()
        print(f"case {case}: S={S} P={P} N={len(steps)} huge={n_huge} [{plan.describe()}] host {t1 - t0:.1f} s -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
        plan.close()
        del d, u, d2, steps, want_d, want_u
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
