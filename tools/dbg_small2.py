import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import pollen_amd as pa
from oracle import flatgfa_oracle as fo
g = pa.synth(3, 1000, 1, 100000, "pangenome", False)
pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
wd, wu = fo.seg_depth_with_uniq(pools)
d, u = g.seg_depth_with_uniq()
print(os.environ.get("TAG"), "sum d", int(d.sum()), int(wd.sum()), "sum u", int(u.sum()), int(wu.sum()))
''' % root
for tag, env in [("default", {}), ("onepiece", {"FLATGFA_PIECE_STEPS": "1000000"}), ("bigcap", {"FLATGFA_BUCKET_CAP": "20000"}),
                 ("bigcap+onepiece", {"FLATGFA_BUCKET_CAP": "20000", "FLATGFA_PIECE_STEPS": "1000000"}),
                 ("pieces8", {"FLATGFA_BUCKET_CAP": "20000", "FLATGFA_PIECE_STEPS": "12500"})]:
    e = dict(os.environ, FLATGFA_DBG_ACC="1", FLATGFA_SHORT_MAX="0", TAG=tag, **env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(r.stdout.strip(), "|", r.stderr.strip()[-200:])
