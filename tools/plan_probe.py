"""What plan a workload gets under the environment's knobs, and what its calls cost:  python3 tools/plan_probe.py <workload> [calls]"""
import sys, time
import torch
sys.path.insert(0, ".")
import pollen_amd as pa
from pollen_amd import device as dev
from bench import WORKLOADS
wl = sys.argv[1]; calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S, P, L, model = WORKLOADS[wl]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, seg_len = g.soa()
graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
t0 = time.perf_counter(); plan = dev.DepthPlan(graph); t1 = time.perf_counter()
print(f"{wl}: plan in {t1 - t0:.2f} s: {plan.describe()}")
d = torch.zeros(S, dtype=torch.int32, device="cuda:0"); u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
for _ in range(2): plan.seg_depth(d, u)
plan.status()
t0 = time.perf_counter()
for _ in range(calls): plan.seg_depth(d, u)
plan.status(); t1 = time.perf_counter()
print(f"{wl}: {(t1 - t0) / calls * 1e3:.4f} ms per call, checksum {int(d.sum().item())} {int(u.sum().item())}")
