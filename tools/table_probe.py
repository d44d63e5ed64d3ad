import sys, time, ctypes
sys.path.insert(0, ".")
import pollen_amd as pa
from pollen_amd import _lib
from bench import WORKLOADS
S, P, L, model = WORKLOADS["cfgL"]
g = pa.synth(1, S, P, L, model, False)
g.to_device(0)
g.seg_depth_with_uniq()
lib = _lib.lib()
for i in range(8):
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    t0 = time.perf_counter()
    rc = lib.flatgfa_depth_table(g._h, ctypes.byref(p), ctypes.byref(n))
    t1 = time.perf_counter()
    b = bytes((ctypes.c_char * n.value).from_address(p.value))
    t2 = time.perf_counter()
    lib.flatgfa_free_text(p)
    t3 = time.perf_counter()
    del b
    t4 = time.perf_counter()
    print("C call %.2f ms, bytes copy %.2f, free %.2f, del %.2f" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)))
