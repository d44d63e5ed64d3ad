import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pollen_amd as pa
from pollen_amd import device as dev
from bench import WORKLOADS
wl = sys.argv[1]
S, P, L, model = WORKLOADS[wl]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, seg_len = g.soa()
graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
d = torch.empty(S, dtype=torch.int32, device="cuda:0"); u = torch.empty(S, dtype=torch.int32, device="cuda:0")
def run(name, after):
    ts = []
    for r in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); plan = dev.DepthPlan(graph, first=(d, u)); t1 = time.perf_counter()
        ts.append(1e3 * (t1 - t0))
        after(plan)
        plan.close()
    print(f"{wl} {name}: " + " ".join("%.2f" % t for t in ts), flush=True)
run("create, close", lambda p: None)
run("create, describe, close", lambda p: p.describe())
run("create, D2H of the result, close", lambda p: d.cpu())
def q(p):
    p.seg_depth(d, u); p.status()
run("create, a query, close", q)
run("create, sleep 50 ms, close", lambda p: time.sleep(0.05))
run("create, close (again)", lambda p: None)
