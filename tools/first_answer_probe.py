"""What a resident graph image costs until its first verified answer (flatgfa_dev_plan_create_first), and where the plan's creation
spends it (FLATGFA_TIMING=1 prints the stages):  python3 tools/first_answer_probe.py <workload> [runs]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import pollen_amd as pa
from pollen_amd import device as dev
from bench import WORKLOADS
from oracle import flatgfa_oracle as fo
wl = sys.argv[1]; runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S, P, L, model = WORKLOADS[wl]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, seg_len = g.soa()
want_d, want_u = fo.seg_depth_with_uniq(fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER}))
graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
d = torch.empty(S, dtype=torch.int32, device="cuda:0"); u = torch.empty(S, dtype=torch.int32, device="cuda:0")
for r in range(runs):
    d.fill_(-1); u.fill_(-1); torch.cuda.synchronize()
    t0 = time.perf_counter(); plan = dev.DepthPlan(graph, first=(d, u)); t1 = time.perf_counter()   # (complete on return: no synchronize)
    ok = bool((d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all())
    t2 = time.perf_counter(); desc = plan.describe(); t3 = time.perf_counter()   # (waits for the marks' job)
    plan.seg_depth(d, u); plan.status()
    ok2 = bool((d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all())
    print(f"{wl} run {r}: first answer {1e3 * (t1 - t0):.3f} ms (exact: {ok}); marks waited for {1e3 * (t3 - t2):.3f} ms more; second call exact: {ok2}; {desc}", flush=True)
    plan.close()
t0 = time.perf_counter(); plan = dev.DepthPlan(graph); t1 = time.perf_counter()
print(f"{wl}: flatgfa_dev_plan_create alone {1e3 * (t1 - t0):.3f} ms")
