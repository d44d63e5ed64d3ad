"""Small-case differential check of seg_depth_with_uniq against the oracle (run via gpurun)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pollen_amd as pa
from oracle import flatgfa_oracle as fo

os.environ.setdefault("FLATGFA_SHORT_MAX", "0")
for (S, P, L) in [(1000, 1, 300), (1000, 1, 2000), (1000, 4, 2000), (3000, 8, 5000), (1000, 1, 100000)]:
    g = pa.synth(3, S, P, L, "pangenome", False)
    pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
    wd, wu = fo.seg_depth_with_uniq(pools)
    d, u = g.seg_depth_with_uniq()
    bd = np.nonzero(d != wd)[0]
    bu = np.nonzero(u != wu)[0]
    print(S, P, L, "depth bad", len(bd), "uniq bad", len(bu), "sum d", int(d.sum()), int(wd.sum()), "sum u", int(u.sum()), int(wu.sum()))
    if len(bd):
        print("  first bad depth idx", bd[:10], d[bd[:10]], wd[bd[:10]])
    if len(bu):
        print("  first bad uniq idx", bu[:10], u[bu[:10]], wu[bu[:10]])
