"""ctypes loader for libflatgfa.so (the C ABI declared in include/flatgfa.h).

The library is built in-tree by ``make -C pollen_amd/csrc`` (``__graft_entry__.build()``).
There is no fallback of any kind: if the shared object is missing, import fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, c_bool, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint32, c_uint64,
                    c_void_p)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLATGFA_LIB") or os.path.join(PKG_DIR, "lib", "libflatgfa.so")  # (FLATGFA_LIB: a measurement build, tools/variants.sh)


class flatgfa_string_t(ctypes.Structure):
    _fields_ = [("data", c_void_p), ("len", c_int)]


class flatgfa_handle_t(ctypes.Structure):
    _fields_ = [("segment_id", c_uint32), ("is_forward", c_bool)]


class flatgfa_dev_graph_t(ctypes.Structure):
    _fields_ = [("steps", c_void_p), ("n_steps", c_uint64), ("path_begin", c_void_p), ("path_end", c_void_p),
                ("n_paths", c_uint32), ("n_segs", c_uint32), ("seg_len", c_void_p)]


# name -> (restype, argtypes); this table is also what tests/test_capi_symbols.py checks
# against include/flatgfa.h.
SIGNATURES = {
    "flatgfa_parse": (c_void_p, [c_char_p]),
    "flatgfa_free": (None, [c_void_p]),
    "flatgfa_get_segment_count": (c_uint32, [c_void_p]),
    "flatgfa_get_seq": (flatgfa_string_t, [c_void_p, c_uint32]),
    "flatgfa_path_count": (c_uint32, [c_void_p]),
    "flatgfa_get_path_name": (flatgfa_string_t, [c_void_p, c_uint32]),
    "flatgfa_get_path_step_count": (c_uint32, [c_void_p, c_uint32]),
    "flatgfa_get_step": (c_bool, [c_void_p, c_size_t, c_size_t, POINTER(flatgfa_handle_t)]),
    "flatgfa_last_error": (c_char_p, []),
    "flatgfa_parse_bytes": (c_void_p, [c_char_p, c_size_t]),
    "flatgfa_parse_stream_bytes": (c_void_p, [c_char_p, c_size_t]),
    "flatgfa_load": (c_void_p, [c_char_p]),
    "flatgfa_write_flatgfa": (c_int, [c_void_p, c_char_p]),
    "flatgfa_write_flatgfa_prealloc": (c_int, [c_void_p, c_char_p, c_char_p, c_size_t, c_uint32]),
    "flatgfa_translate_prealloc": (c_int, [c_char_p, c_size_t, c_int, c_char_p, c_uint32]),
    "flatgfa_print_gfa": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]),
    "flatgfa_free_text": (None, [c_void_p]),
    "flatgfa_synth": (c_void_p, [c_uint64, c_uint32, c_uint32, c_uint32, c_int, c_bool]),
    "flatgfa_pool": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint64)]),
    "flatgfa_find_path": (c_int64, [c_void_p, c_char_p, c_size_t]),
    "flatgfa_device_count": (c_int, []),
    "flatgfa_warm_device": (c_int, [c_int]),
    "flatgfa_keep_host_memory": (c_int, [c_int]),
    "flatgfa_to_device": (c_int, [c_void_p, c_int]),
    "flatgfa_residency_ms": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    "flatgfa_seg_depth": (c_int, [c_void_p, c_void_p, c_void_p]),
    "flatgfa_path_depth": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_void_p]),
    "flatgfa_depth_table": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]),
    "flatgfa_path_depth_table": (c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_void_p), POINTER(c_size_t)]),
    "flatgfa_path_depth_bed": (c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_void_p), POINTER(c_size_t)]),
    "flatgfa_format_float": (c_int, [c_double, c_int, c_char_p, c_int]),
    "flatgfa_seg_depth_subset": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_void_p]),
    "flatgfa_path_overlaps": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p]),
    "flatgfa_overlap_table": (c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_void_p), POINTER(c_size_t)]),
    "flatgfa_interval_depth": (c_int, [c_void_p, c_uint32, c_void_p, c_void_p, c_uint64, c_void_p]),
    "flatgfa_window_depth_table": (c_int, [c_void_p, c_uint32, c_uint64, POINTER(c_void_p), POINTER(c_size_t)]),
    "flatgfa_bed_depth_table": (c_int, [c_void_p, c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t)]),
    "flatgfa_sharded_create": (c_void_p, [c_void_p, c_void_p, c_int, ctypes.c_uint]),
    "flatgfa_sharded_free": (None, [c_void_p]),
    "flatgfa_sharded_layout": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint32),
                                       POINTER(c_uint32), POINTER(c_uint32), POINTER(c_int)]),
    "flatgfa_sharded_collective_bytes": (c_uint64, [c_void_p, c_int]),
    "flatgfa_sharded_seg_depth": (c_int, [c_void_p, c_void_p, c_void_p]),
    "flatgfa_sharded_path_depth": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_void_p]),
    "flatgfa_sharded_enqueue": (c_int, [c_void_p, c_int]),
    "flatgfa_sharded_sync": (c_int, [c_void_p]),
    "flatgfa_sharded_fetch": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "flatgfa_sharded_ranks_seen": (c_int, [c_void_p]),
    "flatgfa_shard_cuts": (c_int, [c_void_p, c_uint32, c_int, ctypes.c_uint, c_void_p]),
    "flatgfa_dev_path_depth_all": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flatgfa_dev_path_overlaps": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_void_p]),
    "flatgfa_dev_plan_create": (c_void_p, [POINTER(flatgfa_dev_graph_t), c_void_p, c_void_p]),
    "flatgfa_dev_plan_create_first": (c_void_p, [POINTER(flatgfa_dev_graph_t), c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "flatgfa_dev_plan_steps_changed": (c_int, [c_void_p, c_void_p]),
    "flatgfa_dev_pipeline_steps_changed": (c_int, [c_void_p]),
    "flatgfa_dev_plan_destroy": (None, [c_void_p]),
    "flatgfa_dev_release_scratch": (None, []),
    "flatgfa_dev_plan_describe": (c_int, [c_void_p, c_char_p, c_int]),
    "flatgfa_dev_pipeline_create": (c_void_p, [POINTER(flatgfa_dev_graph_t), c_void_p, c_void_p, c_int]),
    "flatgfa_dev_pipeline_destroy": (None, [c_void_p]),
    "flatgfa_dev_pipeline_seg_depth": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "flatgfa_dev_pipeline_path_depth_all": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flatgfa_dev_pipeline_join": (c_int, [c_void_p, c_void_p]),
    "flatgfa_dev_pipeline_status": (c_int, [c_void_p]),
    "flatgfa_dev_pipeline_describe": (c_int, [c_void_p, c_char_p, c_int]),
    "flatgfa_dev_seg_depth": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "flatgfa_dev_path_sums": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flatgfa_dev_status": (c_int, [c_void_p, c_void_p]),
    "flatgfa_dev_profile_enable": (None, [c_int]),
    "flatgfa_dev_profile_read": (c_int, [POINTER(c_char_p), POINTER(c_float), c_int]),
    "flatgfa_dev_profile_overhead_ms": (c_float, [c_int, c_int, c_int, c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load libflatgfa.so once.  When torch is importable it is imported first so that the HIP
    runtime (libamdhip64.so.7) is the single copy torch ships, shared by both."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C pollen_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). pollen_amd has no fallback path.")
    try:  # noqa: SIM105
        import torch  # noqa: F401  (shares one HIP runtime with the process)
    except ImportError:
        pass
    cdll = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)  # AttributeError here == a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = cdll
    # Freed host memory stays with the process (include/flatgfa.h: flatgfa_keep_host_memory): on this driver a process that
    # unmaps host memory pays 10-30 ms on its next kernel launch or copy.  FLATGFA_KEEP_HOST_MEMORY=0 leaves malloc as it is.
    if os.environ.get("FLATGFA_KEEP_HOST_MEMORY", "1") != "0":
        cdll.flatgfa_keep_host_memory(1)
    return _lib


def last_error() -> str:
    return (lib().flatgfa_last_error() or b"").decode(errors="replace")
