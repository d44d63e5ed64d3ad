import sys, os
sys.path.insert(0, os.getcwd())
import torch
import pollen_amd as pa
from pollen_amd import device as dev
g = pa.synth(3, 1_200_000, 3000, 9000, "pangenome", False)   # short + medium + long mix, two range passes
steps, pb, pe, seg_len = g.soa()
graph = dev.DeviceGraph(steps, pb, pe, 1_200_000, seg_len, device="cuda:0")
d = torch.zeros(1_200_000, dtype=torch.int32, device="cuda:0"); u = torch.zeros_like(d)
def used():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
seen = {}
for i in range(40):
    plan = dev.DepthPlan(graph); plan.seg_depth(d, u); plan.status(); plan.close()
    if i in (2, 39):
        seen[i] = used()
        print(i, round(seen[i], 1), "MiB in use")
# host API handles
for i in range(10):
    h = pa.synth(4, 50_000, 100, 5000, "pangenome", False); h.seg_depth_with_uniq(); h.path_overlaps([0, 1]); h.close()
after = used()
print("after handles", round(after, 1), "MiB in use")
TOL = 64.0  # MiB: allocator slack, not a leak
if seen[39] - seen[2] > TOL or after - seen[39] > TOL:
    print("LEAK: device memory grew by more than", TOL, "MiB")
    sys.exit(1)
# the bucket array the library keeps for the next plan (include/flatgfa.h: flatgfa_dev_release_scratch) goes back on request
dev._lib.lib().flatgfa_dev_release_scratch()
released = used()
print("after flatgfa_dev_release_scratch", round(released, 1), "MiB in use")
if released > after:
    print("flatgfa_dev_release_scratch gave nothing back")
    sys.exit(1)
