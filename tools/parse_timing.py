"""GFA text -> FlatGFA: how long the parser takes with its step lists parsed in order and by
several threads (FLATGFA_PARSE_THREADS), on a synthetic graph written to a temporary file.
Usage: python tools/parse_timing.py [S] [P] [L]"""
import hashlib
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pollen_amd as pa  # noqa: E402

S, P, L = (int(x) for x in (sys.argv[1:4] + ["200000", "300", "100000"][len(sys.argv) - 1:]))
g = pa.synth(3, S, P, L, "pangenome", True)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "g.gfa")
    t = time.time()
    g.write_gfa(path)
    n = os.path.getsize(path)
    print(f"wrote {n / 1e6:.0f} MB of GFA text in {time.time() - t:.2f} s", flush=True)
    os.environ["FLATGFA_TIMING"] = "1"
    seen = set()
    for th in ("0", "1", "4", "16", "32"):
        os.environ["FLATGFA_PARSE_THREADS"] = th
        best = 1e9
        for _ in range(3):
            t = time.time()
            g2 = pa.parse(path)
            best = min(best, time.time() - t)
        seen.add(hashlib.sha256(g2.pool("steps").tobytes() + g2.pool("paths").tobytes()).hexdigest())
        print(f"threads {th}: {best:.3f} s = {n / best / 1e6:.0f} MB/s", flush=True)
    assert len(seen) == 1 and (g2.pool("steps") == g.pool("steps")).all()
    print("identical pools")
