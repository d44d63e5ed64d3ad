import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pollen_amd as pa
from pollen_amd import device as dev
g = pa.synth(1, 1_000_000, 1000, 100_000, "pangenome", False)
g.to_device(0)
for nq in (1, 8, 64, 1000):
    q = list(range(nq))
    dev.profile_enable(True); dev.profile_read()
    t0 = time.perf_counter(); t = g.path_overlaps(q); t1 = time.perf_counter()
    dev.profile_enable(False); k = dev.profile_read()
    print(nq, "queries:", round((t1 - t0) * 1e3, 2), "ms host;", {n: round(ms, 3) for n, ms in k}, "touching pairs:", int(t.sum()))
