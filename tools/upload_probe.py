#!/usr/bin/env python3
"""Where making a graph resident spends its time (FLATGFA_TIMING lines on stderr): a cfg-L .flatgfa
file in /dev/shm, loaded and sent to the device three times in a fresh process that has not
touched the GPU before.  FLATGFA_UPLOAD_THREADS=n picks the number of staging threads."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["FLATGFA_TIMING"] = "1"
import pollen_amd as pa
g = pa.synth(1, 1_000_000, 1000, 100_000, "pangenome", False)
path = f"/dev/shm/upload_probe_{os.getpid()}.flatgfa"
g.write_flatgfa(path)
if len(sys.argv) > 1 and sys.argv[1] == "mem":  # the in-memory graph first: is the first upload slow, or the first upload of a file?
    t1 = time.perf_counter()
    g.to_device(0)
    print(f"in-memory graph: to_device {1e3 * (time.perf_counter() - t1):.2f} ms", file=sys.stderr)
try:
    for rep in range(3):
        t0 = time.perf_counter()
        g2 = pa.load(path)
        t1 = time.perf_counter()
        g2.to_device(0)
        t2 = time.perf_counter()
        g2.seg_depth_with_uniq()
        t3 = time.perf_counter()
        print(f"run {rep}: load {1e3 * (t1 - t0):.2f} ms, to_device {1e3 * (t2 - t1):.2f} ms, first query {1e3 * (t3 - t2):.2f} ms", file=sys.stderr)
        g2.close()
finally:
    os.unlink(path)
