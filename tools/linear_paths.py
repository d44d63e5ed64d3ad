"""A graph shaped like a real chromosome graph rather than the benchmark's random walk: every path
runs through the segments once, in order, skipping a random 30 % of them (so runs of consecutive
ids are short, 3.3 on average, and every path crosses every window exactly once).  Checks node depth
against the oracle and prints the kernel times.
Usage (GPU box): python tools/linear_paths.py [S] [P] [density] [reverse every k-th path, 0 = none]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from oracle import flatgfa_oracle as fo  # noqa: E402
from pollen_amd import device as dev  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 140
DENS = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
REV = int(sys.argv[4]) if len(sys.argv) > 4 else 7
rng = np.random.default_rng(5)
chunks, begins = [], [0]
for p in range(P):
    ids = np.nonzero(rng.random(S) < DENS)[0].astype(np.uint32)
    if REV and p % REV == REV // 2:
        ids = ids[::-1]  # some paths run the other way: every step its own run
    chunks.append((ids << 1) | rng.integers(0, 2, size=len(ids)).astype(np.uint32))
    begins.append(begins[-1] + len(ids))
steps = np.concatenate(chunks)
pb, pe = np.array(begins[:-1], np.uint32), np.array(begins[1:], np.uint32)
seg_len = np.ones(S, np.uint32)
print(f"S={S} P={P} N={len(steps)}", flush=True)
paths = np.zeros(P, dtype=fo.PATH_DT)
paths["steps_start"], paths["steps_end"] = pb, pe
segs = np.zeros(S, dtype=fo.SEG_DT)
segs["seq_end"] = seg_len
pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
pools.paths, pools.steps, pools.segs = paths, steps, segs
want_d, want_u = fo.seg_depth_with_uniq(pools)
plan = dev.DepthPlan(dev.DeviceGraph(steps, pb, pe, S, seg_len))
d = torch.zeros(S, dtype=torch.int32, device="cuda")
u = torch.zeros_like(d)
for _ in range(3):
    plan.seg_depth(d, u)
plan.status()
ok = (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()
dev.profile_enable(True)
dev.profile_read()
for _ in range(5):
    plan.seg_depth(d, u)
torch.cuda.synchronize()
dev.profile_enable(False)
per = {}
for n, ms in dev.profile_read():
    per.setdefault(n, []).append(ms)
avg = {k: round(float(np.mean(v)), 4) for k, v in per.items()}
tot = sum(avg.values())
print("bit-exact:", bool(ok), avg, f"sum {tot:.4f} ms = {len(steps) / tot / 1e6:.0f} G steps/s" if tot else "")
sys.exit(0 if ok else 1)
