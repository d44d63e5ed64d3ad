"""Calls in flight: K plans of one resident graph on K streams, called in turn, so that one call's pass 2
runs beside the next call's pass 1 (profiles/NOTES.md R5.1).  Every call is a whole seg_depth_with_uniq into
its own output buffer; all K results are compared with each other and, for the first, with the oracle.

    python3 tools/pipeline_probe.py [workload] [calls]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import pollen_amd as pa  # noqa: E402
from pollen_amd import device as dev  # noqa: E402
from bench import WORKLOADS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfgL"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
S, P, L, model = WORKLOADS[wl]
g = pa.synth(1, S, P, L, model, False)
steps, pb, pe, seg_len = g.soa()
device = torch.device("cuda", 0)
graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
KMAX = 3
plans = [dev.DepthPlan(graph) for _ in range(KMAX)]
streams = [torch.cuda.Stream(device) for _ in range(KMAX)]
bufs = [torch.zeros(2 * S, dtype=torch.int32, device=device) for _ in range(KMAX)]
print("plan:", plans[0].describe())
print("plan[1]:", plans[1].describe())


def call(k):
    with torch.cuda.stream(streams[k]):
        plans[k].seg_depth(bufs[k][:S], bufs[k][S:])


for k in range(KMAX):
    for _ in range(3):
        call(k)
torch.cuda.synchronize()
for k in range(KMAX):
    with torch.cuda.stream(streams[k]):
        plans[k].status()
ref = bufs[0].cpu().numpy()
for k in range(1, KMAX):
    assert (bufs[k].cpu().numpy() == ref).all(), f"plan {k} differs"
if S * L * P <= 200_000_000:
    from oracle import flatgfa_oracle as fo
    pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
    wd, wu = fo.seg_depth_with_uniq(pools)
    got = ref.view(np.uint32)
    assert (got[:S] == wd).all() and (got[S:] == wu).all(), "differs from the oracle"
    print("bit-exact vs oracle")

for rounds in range(2):
    for K in (1, 2, 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(calls):
            call(i % K)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"{wl}: {K} in flight: {(t1 - t0) / calls * 1e3:.4f} ms per call = {P * L / ((t1 - t0) / calls) / 1e9:.1f} G steps/s")
for k in range(KMAX):
    bufs[k].zero_()
for i in range(calls):
    call(i % 2)
torch.cuda.synchronize()
for k in range(KMAX):
    with torch.cuda.stream(streams[k]):
        plans[k].status()
assert (bufs[0].cpu().numpy() == ref).all() and (bufs[1].cpu().numpy() == ref).all(), "pipelined results differ"
print("pipelined results identical")
