"""What a plan found out about a graph (flatgfa_dev_plan_describe: which kernel walks how many paths, how many items carry the
no-claim tag, the bucket capacity ...) for bench workloads and seeded random graphs, one line each -- to be diffed between two
builds of the library (FLATGFA_LIB=pollen_amd/lib_prev/libflatgfa.so) or two settings of a test hook
(FLATGFA_COUNT_PIECES=5): the facts the plan-time counting kernel delivers must not depend on how it is launched.
    python3 tools/plan_facts.py [--time] [workloads...]      (default: a spread of shapes + 40 random graphs)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import pollen_amd as pa
from pollen_amd import device as dev
from bench import WORKLOADS
sys.path.insert(0, "tools")
from fuzz_gpu import random_graph

args = [a for a in sys.argv[1:] if not a.startswith("--")]
timed = "--time" in sys.argv
wls = args or ["cfgS", "cfgM", "cfgL", "cfgL-short", "cfgL-medium", "cfgL-4paths", "cfgL-fewlong", "cfgL-chrom", "chrom-1k", "hap-1k", "hap-10k",
               "hap-chr20", "rep-chr20", "cfgL-4Mseg", "tiny-paths"]
for wl in wls:
    S, P, L, model = WORKLOADS[wl]
    g = pa.synth(1, S, P, L, model, False)
    steps, pb, pe, seg_len = g.soa()
    graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
    d = torch.empty(S, dtype=torch.int32, device="cuda:0"); u = torch.empty(S, dtype=torch.int32, device="cuda:0")
    ts = []
    for r in range(4 if timed else 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); plan = dev.DepthPlan(graph, first=(d, u)); t1 = time.perf_counter()
        ts.append(1e3 * (t1 - t0))
        desc = plan.describe()
        plan.close()
    print(f"{wl}: {desc}" + (f"  first answer ms: {' '.join('%.3f' % t for t in ts)}" if timed else ""), flush=True)
    del graph, g
if not args:
    rng = np.random.default_rng(20261003)
    for k in range(40):
        S, P, steps, pb, pe = random_graph(rng)
        seg_len = np.ones(S, dtype=np.uint32)
        graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
        plan = dev.DepthPlan(graph)
        print(f"random {k} (S={S} P={P} N={len(steps)}): {plan.describe()}", flush=True)
        plan.close()
