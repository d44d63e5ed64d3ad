import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pollen_amd as pa
from pollen_amd import device as dev
from pollen_amd.sharded import ShardedDepth
g = pa.synth(1, 1_000_000, 1000, 100_000, "pangenome", False)
steps, pb, pe, seg_len = g.soa()
graph = dev.DeviceGraph(steps, pb, pe, 1_000_000, seg_len, device="cuda:0")
plan = dev.DepthPlan(graph)
op = ShardedDepth(1_000_000, plan.seg_depth, device=torch.device("cuda:0"), with_uniq=True)
for _ in range(3): op.run()
torch.cuda.synchronize()
for prof in (False, True, False):
    dev.profile_enable(prof); dev.profile_read()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): op.run()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    dev.profile_enable(False); k = dev.profile_read()
    print("events" if prof else "no events", round((t1 - t0) / 50 * 1e3, 5), "ms/step")
