"""Host-side mirror of the reference's FlatGFA interface for the depth path, over the C ABI.

Names and argument meaning follow the reference (cucapra/pollen):
  parse / parse_bytes / load / write_flatgfa / write_gfa      flatgfa-py/flatgfa.pyi:80-93
  seg_depth / seg_depth_with_uniq / path_depth                flatgfa/src/ops/depth.rs:15,45,88
  depth_table / path_depth_table  (`fgfa depth [-d] [-r P]`)  flatgfa/src/cli/cmds.rs:234-285

Every depth method runs the HIP kernels in libflatgfa.so; there is no Python or CPU
implementation of them in this package.
"""
from __future__ import annotations

import ctypes
import os
from typing import Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import flatgfa_handle_t

# Packed reference layouts (flatgfa.rs:71-82, 99-112, 121-133; pool.rs:80-86).
SEG_DT = np.dtype([("name", "<u8"), ("seq_start", "<u4"), ("seq_end", "<u4"),
                   ("opt_start", "<u4"), ("opt_end", "<u4")])
PATH_DT = np.dtype([("name_start", "<u4"), ("name_end", "<u4"), ("steps_start", "<u4"),
                    ("steps_end", "<u4"), ("ov_start", "<u4"), ("ov_end", "<u4")])
LINK_DT = np.dtype([("from_", "<u4"), ("to", "<u4"), ("ov_start", "<u4"), ("ov_end", "<u4")])
SPAN_DT = np.dtype([("start", "<u4"), ("end", "<u4")])
POOLS = ["header", "segs", "paths", "links", "steps", "seq_data", "overlaps", "alignment",
         "name_data", "optional_data", "line_order"]
_POOL_DT = {"header": np.dtype("u1"), "segs": SEG_DT, "paths": PATH_DT, "links": LINK_DT,
            "steps": np.dtype("<u4"), "seq_data": np.dtype("u1"), "overlaps": SPAN_DT,
            "alignment": np.dtype("<u4"), "name_data": np.dtype("u1"),
            "optional_data": np.dtype("u1"), "line_order": np.dtype("u1")}

ERR_BOUNDS = -2
ERR_NO_DEVICE = -3


class FlatGFAError(RuntimeError):
    def __init__(self, what: str, code: int = 0):
        super().__init__(f"{what}: {_lib.last_error()}" + (f" (code {code})" if code else ""))
        self.code = code


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise FlatGFAError(what, rc)


def _take_text(ptr: ctypes.c_void_p, n: ctypes.c_size_t) -> bytes:
    out = bytes((ctypes.c_char * n.value).from_address(ptr.value)) if ptr.value and n.value else b""  # (string_at takes a C int)
    _lib.lib().flatgfa_free_text(ptr)
    return out


class FlatGFA:
    """An owned graph handle (`flatgfa_t`).  Freed on garbage collection or `close()`."""

    def __init__(self, handle: int):
        if not handle:
            raise FlatGFAError("cannot create FlatGFA")
        self._h = ctypes.c_void_p(handle)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().flatgfa_free(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self) -> "FlatGFA":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    # ---- flatgfa-py style list views (flatgfa-py/flatgfa.pyi:80-88) ----
    def _views(self):
        from . import views
        if getattr(self, "_pools_cache", None) is None:
            self._pools_cache = views._Pools(self)
        return views, self._pools_cache

    @property
    def segments(self):
        v, p = self._views()
        return v.SegmentList(p)

    @property
    def paths(self):
        v, p = self._views()
        return v.PathList(p)

    @property
    def links(self):
        v, p = self._views()
        return v.LinkList(p)

    # ---- flatgfa-c accessors (flatgfa-c/src/lib.rs:80-172) ----
    @property
    def segment_count(self) -> int:
        return _lib.lib().flatgfa_get_segment_count(self._h)

    @property
    def path_count(self) -> int:
        return _lib.lib().flatgfa_path_count(self._h)

    def get_seq(self, segment_id: int) -> Optional[bytes]:
        s = _lib.lib().flatgfa_get_seq(self._h, segment_id)
        if segment_id >= self.segment_count:
            assert not s.data and s.len == 0
            return None
        return ctypes.string_at(s.data, s.len)

    def get_path_name(self, path_index: int) -> Optional[bytes]:
        if path_index >= self.path_count:
            s = _lib.lib().flatgfa_get_path_name(self._h, path_index)
            assert not s.data and s.len == 0
            return None
        s = _lib.lib().flatgfa_get_path_name(self._h, path_index)
        return ctypes.string_at(s.data, s.len)

    def get_path_step_count(self, path_index: int) -> int:
        return _lib.lib().flatgfa_get_path_step_count(self._h, path_index)

    def get_step(self, path_index: int, step_index: int) -> Optional[Tuple[int, bool]]:
        out = flatgfa_handle_t()
        ok = _lib.lib().flatgfa_get_step(self._h, path_index, step_index, ctypes.byref(out))
        return (out.segment_id, bool(out.is_forward)) if ok else None

    def find_path(self, name: bytes) -> Optional[int]:
        i = _lib.lib().flatgfa_find_path(self._h, name, len(name))
        return None if i < 0 else int(i)

    # ---- pools ----
    def pool(self, name: str) -> np.ndarray:
        """A copy of one pool as a numpy array in the reference's packed layout."""
        ix = POOLS.index(name)
        data, n, es = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
        _check(_lib.lib().flatgfa_pool(self._h, ix, ctypes.byref(data), ctypes.byref(n), ctypes.byref(es)), "pool")
        dt = _POOL_DT[name]
        assert dt.itemsize == es.value
        if n.value == 0:
            return np.zeros(0, dtype=dt)
        # (ctypes.string_at takes a C int: pools beyond 2 GiB would be cut short)
        raw = (ctypes.c_char * (n.value * es.value)).from_address(data.value)
        return np.frombuffer(raw, dtype=dt).copy()

    def soa(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """The structure-of-arrays image the kernels read: (steps, path_begin, path_end, seg_len)."""
        paths, segs = self.pool("paths"), self.pool("segs")
        return (self.pool("steps"), np.ascontiguousarray(paths["steps_start"]),
                np.ascontiguousarray(paths["steps_end"]),
                (segs["seq_end"] - segs["seq_start"]).astype(np.uint32))

    # ---- writers (flatgfa-py: write_flatgfa / write_gfa / str) ----
    def write_flatgfa(self, filename: str) -> None:
        _check(_lib.lib().flatgfa_write_flatgfa(self._h, os.fsencode(filename)), "write_flatgfa")

    def write_flatgfa_prealloc(self, filename: str, gfa_text: Optional[bytes] = None, factor: int = 32) -> None:
        """The preallocated container of `fgfa -m -p FACTOR -o OUT [-I GFA]` (cli/main.rs:216-248):
        capacities estimated from `gfa_text` (the text this graph was parsed from), or guessed from `factor`."""
        _check(_lib.lib().flatgfa_write_flatgfa_prealloc(self._h, os.fsencode(filename), gfa_text, len(gfa_text) if gfa_text is not None else 0,
                                                         int(factor)), "write_flatgfa_prealloc")

    def gfa_text(self) -> bytes:
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(_lib.lib().flatgfa_print_gfa(self._h, ctypes.byref(p), ctypes.byref(n)), "print_gfa")
        return _take_text(p, n)

    def write_gfa(self, filename: str) -> None:
        with open(filename, "wb") as f:
            f.write(self.gfa_text())

    def __str__(self) -> str:
        return self.gfa_text().decode(errors="replace")

    # ---- depth queries (HIP) ----
    def to_device(self, device: int = 0) -> None:
        _check(_lib.lib().flatgfa_to_device(self._h, device), "to_device")

    def residency_ms(self) -> Tuple[float, float]:
        """(host-to-device copies, plan creation) of to_device on this handle, milliseconds."""
        h2d, plan = ctypes.c_double(0), ctypes.c_double(0)
        _check(_lib.lib().flatgfa_residency_ms(self._h, ctypes.byref(h2d), ctypes.byref(plan)), "residency_ms")
        return h2d.value, plan.value

    def seg_depth_with_uniq(self) -> Tuple[np.ndarray, np.ndarray]:
        """ops/depth.rs:15-39 -> (depths, uniq_depths), uint64, indexed by segment id."""
        S = self.segment_count
        d, u = np.zeros(S, np.uint64), np.zeros(S, np.uint64)
        _check(_lib.lib().flatgfa_seg_depth(self._h, d.ctypes.data, u.ctypes.data), "seg_depth_with_uniq")
        return d, u

    def seg_depth(self) -> np.ndarray:
        """ops/depth.rs:45-56"""
        d = np.zeros(self.segment_count, np.uint64)
        _check(_lib.lib().flatgfa_seg_depth(self._h, d.ctypes.data, None), "seg_depth")
        return d

    def path_depth(self, path_ids: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, np.ndarray]:
        """ops/depth.rs:88-111 -> (path_lengths uint64, mean depths float64) for `path_ids`
        (default: all paths, in order)."""
        ids = np.arange(self.path_count, dtype=np.uint32) if path_ids is None \
            else np.ascontiguousarray(path_ids, dtype=np.uint32)
        ln, mean = np.zeros(len(ids), np.uint64), np.zeros(len(ids), np.float64)
        _check(_lib.lib().flatgfa_path_depth(self._h, ids.ctypes.data, len(ids), ln.ctypes.data, mean.ctypes.data),
               "path_depth")
        return ln, mean

    def depth_table(self) -> bytes:
        """The bytes `fgfa depth -d` prints."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(_lib.lib().flatgfa_depth_table(self._h, ctypes.byref(p), ctypes.byref(n)), "depth_table")
        return _take_text(p, n)

    def path_depth_table(self, names: Optional[Iterable[bytes]] = None) -> bytes:
        """The bytes `fgfa depth [-r NAME]...` prints; unknown names are dropped (cmds.rs:270-274)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        if names is None:
            rc = _lib.lib().flatgfa_path_depth_table(self._h, None, 0, ctypes.byref(p), ctypes.byref(n))
        else:
            found = [self.find_path(nm) for nm in names]
            ids = np.array([i for i in found if i is not None] + [0], dtype=np.uint32)
            rc = _lib.lib().flatgfa_path_depth_table(self._h, ids.ctypes.data, len(ids) - 1, ctypes.byref(p),
                                                     ctypes.byref(n))
        _check(rc, "path_depth_table")
        return _take_text(p, n)

    def path_depth_bed(self, names: Optional[Iterable[bytes]] = None) -> bytes:
        """PathDepth::as_bed (depth.rs:173-183) as three-column BED text: `name\\t0\\tlength` per path."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        if names is None:
            rc = _lib.lib().flatgfa_path_depth_bed(self._h, None, 0, ctypes.byref(p), ctypes.byref(n))
        else:
            found = [self.find_path(nm) for nm in names]
            ids = np.array([i for i in found if i is not None] + [0], dtype=np.uint32)
            rc = _lib.lib().flatgfa_path_depth_bed(self._h, ids.ctypes.data, len(ids) - 1, ctypes.byref(p), ctypes.byref(n))
        _check(rc, "path_depth_bed")
        return _take_text(p, n)


    # ---- the rows next to the depth path ----
    def _ids(self, paths) -> np.ndarray:
        """Path ids from a list of ids or names; unknown names raise (slow_odgi asserts, overlap.py:21)."""
        out = []
        for p in paths:
            if isinstance(p, (bytes, bytearray)):
                i = self.find_path(bytes(p))
                if i is None:
                    raise FlatGFAError(f"path {bytes(p)!r} not found")
                out.append(i)
            else:
                out.append(int(p))
        return np.ascontiguousarray(out, dtype=np.uint32)

    def seg_depth_subset(self, paths) -> Tuple[np.ndarray, np.ndarray]:
        """Node depth counting only `paths` (`odgi depth -d -s`, slow_odgi/depth.py:12)."""
        ids = self._ids(paths)
        S = self.segment_count
        d, u = np.zeros(S, np.uint64), np.zeros(S, np.uint64)
        _check(_lib.lib().flatgfa_seg_depth_subset(self._h, ids.ctypes.data if len(ids) else None, len(ids),
                                                   d.ctypes.data, u.ctypes.data), "seg_depth_subset")
        return d, u

    def path_overlaps(self, queries) -> np.ndarray:
        """touch[k, j] = path j touches query k (slow_odgi/overlap.py:6-14)."""
        ids = self._ids(queries)
        t = np.zeros((len(ids), self.path_count), np.uint8)
        _check(_lib.lib().flatgfa_path_overlaps(self._h, ids.ctypes.data if len(ids) else None, len(ids),
                                                t.ctypes.data if t.size else None), "path_overlaps")
        return t

    def overlap_table(self, queries) -> bytes:
        """The bytes `slow_odgi overlap --paths FILE` prints."""
        ids = self._ids(queries)
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(_lib.lib().flatgfa_overlap_table(self._h, ids.ctypes.data if len(ids) else None, len(ids),
                                                ctypes.byref(p), ctypes.byref(n)), "overlap_table")
        return _take_text(p, n)

    def interval_depth(self, path, starts, ends) -> np.ndarray:
        """ops/window_depth.rs:176-180"""
        (pid,) = self._ids([path])
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        en = np.ascontiguousarray(ends, dtype=np.uint64)
        out = np.zeros(len(st), np.float64)
        _check(_lib.lib().flatgfa_interval_depth(self._h, int(pid), st.ctypes.data, en.ctypes.data, len(st),
                                                 out.ctypes.data), "interval_depth")
        return out

    def window_depth_table(self, path, window: int) -> bytes:
        """The bytes `fgfa window-depth PATH SIZE` prints."""
        (pid,) = self._ids([path])
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(_lib.lib().flatgfa_window_depth_table(self._h, int(pid), window, ctypes.byref(p), ctypes.byref(n)),
               "window_depth_table")
        return _take_text(p, n)

    def bed_depth_table(self, bed: bytes) -> bytes:
        """The bytes `fgfa depth -b FILE.bed` prints."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(_lib.lib().flatgfa_bed_depth_table(self._h, bed, len(bed), ctypes.byref(p), ctypes.byref(n)),
               "bed_depth_table")
        return _take_text(p, n)


SHARD_WHOLE_PATHS = 1  # FLATGFA_SHARD_WHOLE_PATHS
SHARD_NO_RCCL = 2      # FLATGFA_SHARD_NO_RCCL


class ShardedFlatGFA:
    """One graph sharded over the GPUs of a node by this process (`flatgfa_sharded_t`): the local
    kernels on every shard, one RCCL all-reduce of the fused [depth | uniq] vector, results as the
    single-device calls give them.  `devices[i]` is shard i's HIP device (default: shard i on
    device i mod the device count); devices may repeat (shards that share one exchange by
    device-side adds)."""

    def __init__(self, graph: FlatGFA, n_shards: int, devices: Optional[Sequence[int]] = None, flags: int = 0):
        self.graph = graph  # the handle borrows the graph
        self.n_shards = int(n_shards)
        arr = None
        if devices is not None:
            if len(devices) != self.n_shards:
                raise ValueError("one device per shard")
            arr = (ctypes.c_int * self.n_shards)(*[int(d) for d in devices])
        self._h = ctypes.c_void_p(_lib.lib().flatgfa_sharded_create(graph._h, arr, self.n_shards, int(flags)))
        if not self._h.value:
            raise FlatGFAError(f"flatgfa_sharded_create: {_lib.last_error()}")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().flatgfa_sharded_free(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self) -> "ShardedFlatGFA":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def layout(self) -> List[dict]:
        """Per shard: its device, its stretch of the steps pool, the first path it walks, how many
        paths or pieces it walks; plus how many paths the handle cuts and whether RCCL carries the exchange."""
        out = []
        for i in range(self.n_shards):
            dev, uses = ctypes.c_int(), ctypes.c_int()
            b, e = ctypes.c_uint64(), ctypes.c_uint64()
            fp, npc, nsp = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
            _check(_lib.lib().flatgfa_sharded_layout(self._h, i, ctypes.byref(dev), ctypes.byref(b), ctypes.byref(e), ctypes.byref(fp),
                                                     ctypes.byref(npc), ctypes.byref(nsp), ctypes.byref(uses)), "sharded_layout")
            out.append({"device": dev.value, "step_begin": b.value, "step_end": e.value, "first_path": fp.value,
                        "pieces": npc.value, "split_paths": nsp.value, "rccl": bool(uses.value)})
        return out

    def collective_bytes(self, with_uniq: bool = True) -> int:
        """Bytes every shard contributes to the one collective of a call (the cut paths' touch counters travel packed)."""
        return int(_lib.lib().flatgfa_sharded_collective_bytes(self._h, 1 if with_uniq else 0))

    def seg_depth_with_uniq(self) -> Tuple[np.ndarray, np.ndarray]:
        S = self.graph.segment_count
        d, u = np.zeros(S, dtype=np.uint64), np.zeros(S, dtype=np.uint64)
        _check(_lib.lib().flatgfa_sharded_seg_depth(self._h, d.ctypes.data, u.ctypes.data), "sharded_seg_depth")
        return d, u

    def seg_depth(self) -> np.ndarray:
        d = np.zeros(self.graph.segment_count, dtype=np.uint64)
        _check(_lib.lib().flatgfa_sharded_seg_depth(self._h, d.ctypes.data, None), "sharded_seg_depth")
        return d

    def path_depth(self, path_ids: Optional[Iterable[int]] = None) -> Tuple[np.ndarray, np.ndarray]:
        ids = np.arange(self.graph.path_count, dtype=np.uint32) if path_ids is None else np.asarray(list(path_ids), dtype=np.uint32)
        ln, mean = np.zeros(len(ids), dtype=np.uint64), np.zeros(len(ids), dtype=np.float64)
        _check(_lib.lib().flatgfa_sharded_path_depth(self._h, ids.ctypes.data if len(ids) else None, len(ids),
                                                     ln.ctypes.data if len(ids) else None, mean.ctypes.data if len(ids) else None),
               "sharded_path_depth")
        return ln, mean

    def ranks_seen(self) -> int:
        """How many ranks the handle's exchange really spans (an all-reduce of ones over its communicator)."""
        n = int(_lib.lib().flatgfa_sharded_ranks_seen(self._h))
        if n < 0:
            _check(n, "sharded_ranks_seen")
        return n

    def enqueue(self, with_uniq: bool = True) -> None:
        _check(_lib.lib().flatgfa_sharded_enqueue(self._h, 1 if with_uniq else 0), "sharded_enqueue")

    def sync(self) -> None:
        _check(_lib.lib().flatgfa_sharded_sync(self._h), "sharded_sync")

    def fetch(self, shard: int = 0, with_uniq: bool = True):
        S = self.graph.segment_count
        d = np.zeros(S, dtype=np.uint64)
        u = np.zeros(S, dtype=np.uint64) if with_uniq else None
        _check(_lib.lib().flatgfa_sharded_fetch(self._h, int(shard), d.ctypes.data, u.ctypes.data if with_uniq else None), "sharded_fetch")
        return (d, u) if with_uniq else d


def shard_cuts(path_steps, n_shards: int, flags: int = 0) -> np.ndarray:
    """Where `ShardedFlatGFA` cuts a graph whose paths, in path order, have `path_steps[p]` steps
    (flatgfa_shard_cuts: host only, no device needed): n_shards + 1 cut points counted in path steps along
    the path order; shard r walks [cuts[r], cuts[r + 1])."""
    ps = np.ascontiguousarray(path_steps, dtype=np.uint64)
    out = np.zeros(int(n_shards) + 1, dtype=np.uint64)
    _check(_lib.lib().flatgfa_shard_cuts(ps.ctypes.data if len(ps) else None, len(ps), int(n_shards), int(flags), out.ctypes.data), "shard_cuts")
    return out


def parse(filename: Union[str, os.PathLike]) -> FlatGFA:
    """Parse a GFA text file (flatgfa_parse, flatgfa-c/src/lib.rs:63)."""
    return FlatGFA(_lib.lib().flatgfa_parse(os.fsencode(filename)))


def parse_bytes(gfa: bytes) -> FlatGFA:
    return FlatGFA(_lib.lib().flatgfa_parse_bytes(gfa, len(gfa)))


def parse_stream_bytes(gfa: bytes) -> FlatGFA:
    return FlatGFA(_lib.lib().flatgfa_parse_stream_bytes(gfa, len(gfa)))


def translate_prealloc(gfa: bytes, filename: Union[str, os.PathLike], factor: int = 32, from_stream: bool = False) -> None:
    """`fgfa -m -p FACTOR -o OUT [-I GFA]` (cli/main.rs:216-248): the text is parsed straight into the
    mapped, preallocated output file (file.rs:255-272); no graph is built in between.  `from_stream`:
    the text came from stdin (capacities guessed from `factor`, parse_stream's rules)."""
    _check(_lib.lib().flatgfa_translate_prealloc(gfa, len(gfa), 1 if from_stream else 0, os.fsencode(filename), int(factor)),
           "translate_prealloc")


def load(filename: Union[str, os.PathLike]) -> FlatGFA:
    """Map a binary `.flatgfa` file (file::view, flatgfa/src/file.rs:185)."""
    return FlatGFA(_lib.lib().flatgfa_load(os.fsencode(filename)))


def synth(seed: int, n_segs: int, n_paths: int, steps_per_path: int, model: str = "pangenome",
          with_seq: bool = False) -> FlatGFA:
    """The deterministic synthetic graph of SURVEY.md 8(d)."""
    m = {"pangenome": 0, "uniform": 1, "chromosome": 2, "haplotype": 3, "repeats": 4}[model]
    return FlatGFA(_lib.lib().flatgfa_synth(seed, n_segs, n_paths, steps_per_path, m, with_seq))


def format_float(x: float, digits: int) -> str:
    buf = ctypes.create_string_buffer(600)
    n = _lib.lib().flatgfa_format_float(x, digits, buf, 600)
    return buf.raw[:n].decode()


def device_count() -> int:
    return _lib.lib().flatgfa_device_count()
