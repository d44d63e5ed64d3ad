"""A plan made and dropped in a loop beside K plans that stay (what bench.py's `extras.first_answer` does beside its pipelines): python3 tools/flow_probe2.py <workload> [K] [n]"""
import sys, time
import torch
sys.path.insert(0, ".")
import pollen_amd as pa
from pollen_amd import device as dev
from bench import WORKLOADS
wl = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 5; n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
S, P, L, model = WORKLOADS[wl]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, seg_len = g.soa()
graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
d = torch.empty(S, dtype=torch.int32, device="cuda:0"); u = torch.empty(S, dtype=torch.int32, device="cuda:0")
stay = [dev.DepthPlan(graph) for _ in range(K)]
for p in stay:
    p.seg_depth(d, u); p.status(); p.describe()
ts = []
for r in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); plan = dev.DepthPlan(graph, first=(d, u)); ts.append(1e3 * (time.perf_counter() - t0))
    plan.close()
print(f"{wl} beside {K} plans: " + " ".join("%.1f" % t for t in ts), flush=True)
