import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pollen_amd as pa
from pollen_amd import device as dev
from pollen_amd.sharded import ShardedDepth, local_slice, shard_paths
from oracle import flatgfa_oracle as fo
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); device = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
S, P, L = 10000, 100, 10000
g = pa.synth(1, S, P, L, "pangenome", False)
steps, pb, pe, seg_len = g.soa()
lo, hi = shard_paths(pb, pe, world)[rank]
ls, lb, le = local_slice(steps, pb, pe, lo, hi)
graph = dev.DeviceGraph(ls, lb, le, S, seg_len, device=str(device))
plan = dev.DepthPlan(graph)
d = torch.zeros(S, dtype=torch.int32, device=device); u = torch.zeros(S, dtype=torch.int32, device=device)
plan.seg_depth(d, u); plan.status()
pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
sub = fo.Pools(**{n: getattr(pools, n) for n in fo.POOL_ORDER})
sub.steps = ls
paths = np.zeros(hi - lo, dtype=fo.PATH_DT); paths["steps_start"], paths["steps_end"] = lb, le
sub.paths = paths
wd, wu = fo.seg_depth_with_uniq(sub)
print(rank, "local ok", bool((d.cpu().numpy().view(np.uint32) == wd).all()), bool((u.cpu().numpy().view(np.uint32) == wu).all()), lo, hi, len(ls), flush=True)
op = ShardedDepth(S, plan.seg_depth, device=device, with_uniq=True)
op.run(); op.finish(); torch.cuda.synchronize()
fd, fu = fo.seg_depth_with_uniq(pools)
got = op.buf.cpu().numpy().view(np.uint32)
print(rank, "reduced ok", bool((got[:S] == fd).all()), bool((got[S:] == fu).all()), int(got[:S].sum()), int(fd.sum()), flush=True)
dist.barrier(); dist.destroy_process_group()
