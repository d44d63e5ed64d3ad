"""Per-kernel HIP-event times of seg_depth_with_uniq on one workload, one call after the other: for same-box A/B runs of
library variants (FLATGFA_LIB=pollen_amd/lib_<tag>/libflatgfa.so, tools/variants.sh) and of environment knobs.

    python3 tools/ab_kernels.py <workload> [calls]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import pollen_amd as pa  # noqa: E402
from pollen_amd import device as dev  # noqa: E402
from bench import WORKLOADS  # noqa: E402

wl = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S, P, L, model = WORKLOADS[wl]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, seg_len = g.soa()
graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
plan = dev.DepthPlan(graph)
extra = [dev.DepthPlan(graph) for _ in range(int(os.environ.get("AB_EXTRA_PLANS", "0")))]  # (other plans of the same graph alive beside the one measured)
if os.environ.get("AB_SIDE_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream("cuda:0"))
d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
for _ in range(3):
    plan.seg_depth(d, u)
plan.status()
plan.describe()  # (waits for what the plan makes behind its creation on a side stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    plan.seg_depth(d, u)
plan.status()
t1 = time.perf_counter()
dev.profile_enable(True)
dev.profile_read()
for _ in range(calls):
    plan.seg_depth(d, u)
plan.status()
dev.profile_enable(False)
per = {}
for name, ms in dev.profile_read():
    per.setdefault(name, []).append(ms)
tag = os.environ.get("FLATGFA_LIB", "default").split("/")[-2] if os.environ.get("FLATGFA_LIB") else "default"
knobs = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if (k.startswith("FLATGFA_") and k != "FLATGFA_LIB") or k.startswith("AB_"))
print(f"{wl} [{tag}{' ' + knobs if knobs else ''}] call {(t1 - t0) / calls * 1e3:.4f} ms | " +
      " ".join(f"{k} {np.mean(v) * 1e3:.1f}us" for k, v in per.items()) + f" | checksum {int(d.sum().item())} {int(u.sum().item())}")
