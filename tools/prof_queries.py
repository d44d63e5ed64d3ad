"""The two queries next to node depth, run a few times each so that rocprofv3 can time their kernels (tools/profile_round.sh):
path_depth of all paths (a3: k_scan, k_accum<depth+paths>, k_path_reduce) and all-pairs path overlap (BASELINE.json configs[4]:
k_coarse_bits and k_handle_bits once, k_pair_touch per call) on one workload.

    python3 tools/prof_queries.py [workload] [calls]
"""
import sys

import torch

sys.path.insert(0, ".")
import pollen_amd as pa  # noqa: E402
from pollen_amd import device as dev  # noqa: E402
from bench import WORKLOADS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfgL"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S, P, L, model = WORKLOADS[wl]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, seg_len = g.soa()
graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
plan = dev.DepthPlan(graph)
d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
ln = torch.zeros(P, dtype=torch.int64, device="cuda:0")
ws = torch.zeros(P, dtype=torch.int64, device="cuda:0")
for _ in range(calls):
    plan.path_depth_all(d, ln, ws)
plan.status()
q = torch.arange(P, dtype=torch.int32, device="cuda:0")
t = torch.zeros(P * P, dtype=torch.uint8, device="cuda:0")
for _ in range(calls):
    plan.path_overlaps(q, t)
plan.status()
print(wl, "path_depth_all and path_overlaps x", calls, "touching pairs", int(t.sum().item()), "sum of lengths", int(ln.sum().item()))
