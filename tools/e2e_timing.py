"""Where `load -> to_device -> query -> table` spends its time (FLATGFA_TIMING=1), with the device already initialised."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
torch.zeros(1, device="cuda:0")  # HIP runtime + device initialised before anything is timed (as in bench.py)
import pollen_amd as pa
g = pa.synth(1, 1_000_000, 1000, 100_000, "pangenome", False)
g.write_flatgfa("/dev/shm/e2e.flatgfa")
os.environ["FLATGFA_TIMING"] = "1"
for rep in range(2):
    t0 = time.perf_counter(); g2 = pa.load("/dev/shm/e2e.flatgfa"); t1 = time.perf_counter()
    g2.to_device(0); t2 = time.perf_counter()
    g2.seg_depth_with_uniq(); t3 = time.perf_counter()
    txt = g2.depth_table(); t4 = time.perf_counter()
    print("load %.1f to_device %.1f query %.1f table %.1f total %.1f ms (%d bytes)" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3, len(txt)))
    g2.close()
os.unlink("/dev/shm/e2e.flatgfa")
