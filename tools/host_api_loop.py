"""Repeated queries through the host API (flatgfa_seg_depth: kernels + one copy back + widening to u64) with the results dropped
between them, and the same keeping them: what freeing 16 MB of host memory per query costs the next one on this driver (NOTES R6.6c).
    python3 tools/host_api_loop.py [workload] [queries]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pollen_amd as pa
from bench import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "cfgL"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
S, P, L, model = WORKLOADS[wl]
g = pa.synth(1, S, P, L, model, False)
g.to_device(0)
def loop(name, keep):
    kept, ts = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        r = g.seg_depth_with_uniq()
        ts.append(1e3 * (time.perf_counter() - t0))
        if keep:
            kept.append(r)
        del r
    print(f"{wl} {name}: " + " ".join("%.2f" % t for t in ts), flush=True)
loop("results dropped", False)
loop("results kept", True)
loop("results dropped", False)
def table(name):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        t = g.depth_table()
        ts.append(1e3 * (time.perf_counter() - t0))
        del t
    print(f"{wl} {name}: " + " ".join("%.2f" % t for t in ts), flush=True)
table("depth table, dropped")
