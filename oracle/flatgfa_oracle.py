"""flatgfa_oracle -- TEST INFRASTRUCTURE ONLY.

A small, clarity-first CPU restatement of the *data side* of the reference's
depth path: GFA text -> FlatGFA pools (parse.rs / gfaline.rs / namemap.rs /
memfile.rs), the zero-copy ``.flatgfa`` container (file.rs), and ctypes
wrappers around ``depth_oracle.c`` (ops/depth.rs).  Pure Python + numpy; the
parser uses Python loops and is meant for small inputs only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; nothing under ``pollen_amd/`` does.

Pools are kept in the reference's own packed layouts so that the C oracle and
the ``.flatgfa`` reader/writer see the same bytes the Rust code would:

  Segment  24 B  name:u64@0  seq:{u32,u32}@8  optional:{u32,u32}@16   flatgfa.rs:71-82
  Path     24 B  name:{u32,u32}@0  steps:{u32,u32}@8  overlaps:{u32,u32}@16   flatgfa.rs:99-112
  Link     16 B  from:u32@0  to:u32@4  overlap:{u32,u32}@8            flatgfa.rs:121-133
  Handle    4 B  (seg << 1) | orient, Forward=0, Backward=1           flatgfa.rs:149-154,186-209
  AlignOp   4 B  (len << 8) | opcode, M=0 N=1 I=2 D=3                 flatgfa.rs:211-251, gfaline.rs:178-184
  Span      8 B  {start:u32, end:u32}                                 pool.rs:80-86
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SEG_DT = np.dtype([("name", "<u8"), ("seq_start", "<u4"), ("seq_end", "<u4"),
                   ("opt_start", "<u4"), ("opt_end", "<u4")])
PATH_DT = np.dtype([("name_start", "<u4"), ("name_end", "<u4"),
                    ("steps_start", "<u4"), ("steps_end", "<u4"),
                    ("ov_start", "<u4"), ("ov_end", "<u4")])
LINK_DT = np.dtype([("from_", "<u4"), ("to", "<u4"),
                    ("ov_start", "<u4"), ("ov_end", "<u4")])
SPAN_DT = np.dtype([("start", "<u4"), ("end", "<u4")])
assert SEG_DT.itemsize == 24 and PATH_DT.itemsize == 24 and LINK_DT.itemsize == 16

MAGIC = 0xB101_1054  # file.rs:9
# LineKind, flatgfa.rs:262-269
LK_HEADER, LK_SEGMENT, LK_PATH, LK_LINK = 0, 1, 2, 3
# pool order inside a .flatgfa file, file.rs:14-27
POOL_ORDER = ["header", "segs", "paths", "links", "steps", "seq_data",
              "overlaps", "alignment", "name_data", "optional_data", "line_order"]
POOL_DTYPES = {"header": np.dtype("u1"), "segs": SEG_DT, "paths": PATH_DT,
               "links": LINK_DT, "steps": np.dtype("<u4"), "seq_data": np.dtype("u1"),
               "overlaps": SPAN_DT, "alignment": np.dtype("<u4"),
               "name_data": np.dtype("u1"), "optional_data": np.dtype("u1"),
               "line_order": np.dtype("u1")}


class ParseError(Exception):
    """Raised where the reference would panic (unwrap / assert / index)."""


@dataclass
class Pools:
    """The eleven pools of a FlatGFA (flatgfa.rs:19-67) as numpy arrays."""
    header: np.ndarray
    segs: np.ndarray
    paths: np.ndarray
    links: np.ndarray
    steps: np.ndarray
    seq_data: np.ndarray
    overlaps: np.ndarray
    alignment: np.ndarray
    name_data: np.ndarray
    optional_data: np.ndarray
    line_order: np.ndarray

    def path_name(self, i: int) -> bytes:
        p = self.paths[i]
        return self.name_data[p["name_start"]:p["name_end"]].tobytes()

    def seg_lens(self) -> np.ndarray:
        return (self.segs["seq_end"] - self.segs["seq_start"]).astype(np.uint32)


# --------------------------------------------------------------------------
# GFA text -> pools
# --------------------------------------------------------------------------

def _memchr_split(buf: bytes) -> List[bytes]:
    """memfile.rs:51-63.  Lines end at '\\n'; a final line with no terminator
    is never yielded (memchr returns None -> the iterator stops)."""
    out = []
    pos = 0
    n = len(buf)
    while pos < n:
        end = buf.find(b"\n", pos)
        if end < 0:
            break
        out.append(buf[pos:end])
        pos = end + 1
    return out


def _parse_num(s: bytes) -> Tuple[int, bytes]:
    """gfaline.rs:153-158 (atoi::FromRadix10: leading ASCII digits only)."""
    i = 0
    while i < len(s) and 48 <= s[i] <= 57:
        i += 1
    if i == 0:
        raise ParseError("expected number")
    return int(s[:i]), s[i:]


def _parse_byte(s: bytes, b: int) -> bytes:
    """gfaline.rs:145-150"""
    if not s or s[0] != b:
        raise ParseError("expected byte")
    return s[1:]


def _parse_field(s: bytes) -> Tuple[bytes, bytes]:
    """gfaline.rs:128-142: up to the next tab; rest is empty if none."""
    end = s.find(b"\t")
    if end < 0:
        return s, b""
    return s[:end], s[end + 1:]


def _parse_orient(s: bytes) -> Tuple[int, bytes]:
    """gfaline.rs:161-171.  Forward=0, Backward=1."""
    if not s:
        raise ParseError("expected orientation")
    if s[0] == 0x2B:
        return 0, s[1:]
    if s[0] == 0x2D:
        return 1, s[1:]
    raise ParseError("expected orient")


_ALIGN_OPCODE = {ord("M"): 0, ord("N"): 1, ord("D"): 3, ord("I"): 2}  # gfaline.rs:178-184


def _parse_align(s: bytes) -> Tuple[List[int], bytes]:
    """gfaline.rs:174-198"""
    ops = []
    while s and 48 <= s[0] <= 57:
        ln, s = _parse_num(s)
        if not s:
            raise ParseError("align op: index out of bounds")
        if s[0] not in _ALIGN_OPCODE:
            raise ParseError("expected align op")
        if ln & ~0xFF:
            raise ParseError("length too large")  # flatgfa.rs:228
        if ln >= 1 << 32:
            raise ParseError("number too large")
        ops.append((ln << 8) | _ALIGN_OPCODE[s[0]])
        s = s[1:]
    return ops, s


def _parse_overlaps(s: bytes) -> List[List[int]]:
    """gfaline.rs:102-125"""
    if s == b"*":
        return []
    out = []
    while s:
        ops, s = _parse_align(s)
        out.append(ops)
        if s:
            s = _parse_byte(s, 0x2C)
    return out


def _parse_steps(s: bytes):
    """StepsParser, gfaline.rs:200-263.  Returns ([(name, is_forward)], rest).
    Note the quirk: the byte that stops the scan has already been consumed
    when `rest()` is taken."""
    out = []
    idx = 0
    seg = 0
    state_seg = True
    stopped = False
    while idx < len(s):
        byte = s[idx]
        idx += 1
        if state_seg:
            if byte == 0x2B or byte == 0x2D:
                state_seg = False
                out.append((seg, byte == 0x2B))
            elif 48 <= byte <= 57:
                seg = seg * 10 + (byte - 48)
            else:
                stopped = True
                break
        else:
            if byte == 0x2C:
                state_seg = True
                seg = 0
            else:
                stopped = True
                break
    del stopped
    return out, s[idx:]


class _NameMap:
    """namemap.rs:7-33"""

    def __init__(self):
        self.sequential_max = 0
        self.others: Dict[int, int] = {}

    def insert(self, name: int, idx: int) -> None:
        nm1 = (name - 1) & 0xFFFFFFFFFFFFFFFF  # release-mode wrap
        if nm1 == self.sequential_max and nm1 == idx:
            self.sequential_max += 1
        else:
            self.others[name] = idx

    def get(self, name: int) -> int:
        if name <= self.sequential_max:
            idx = (name - 1) & 0xFFFFFFFF
        else:
            if name not in self.others:
                raise ParseError("unknown segment name")
            idx = self.others[name]
        return idx


def _handle(idx: int, forward: bool) -> int:
    """Handle::new, flatgfa.rs:191-198 (+ From<bool>, :170-178)."""
    if idx & (1 << 31):
        raise ParseError("index too large")
    return ((idx << 1) | (0 if forward else 1)) & 0xFFFFFFFF


def parse_gfa(buf: bytes) -> Pools:
    """Parser::parse_mem, parse.rs:77-126 (+ add_seg/add_link/add_path :138-159)."""
    header = bytearray()
    segs: List[Tuple[int, int, int, int, int]] = []
    seq_data = bytearray()
    optional_data = bytearray()
    line_order = bytearray()
    deferred: List[bytes] = []
    names = _NameMap()
    have_header = False

    for line in _memchr_split(buf):
        if len(line) == 0:
            raise ParseError("index out of bounds: empty line")  # parse.rs:83 line[0]
        if line[0] in (0x50, 0x4C):  # 'P' / 'L'
            line_order.append(LK_PATH if line[0] == 0x50 else LK_LINK)
            deferred.append(line)
            continue
        if len(line) < 2 or line[1] != 0x09:
            raise ParseError("expected marker and tab")
        rest = line[2:]
        if line[0] == 0x48:  # 'H'
            line_order.append(LK_HEADER)
            if have_header and len(header) > 0:
                raise ParseError("duplicate header")  # flatgfa.rs:444 assert
            header += rest
            have_header = True
        elif line[0] == 0x53:  # 'S'
            name, r = _parse_num(rest)
            r = _parse_byte(r, 0x09)
            seq, data = _parse_field(r)
            line_order.append(LK_SEGMENT)
            s0 = len(seq_data)
            seq_data += seq
            o0 = len(optional_data)
            optional_data += data
            idx = len(segs)
            segs.append((name & 0xFFFFFFFFFFFFFFFF, s0, len(seq_data), o0, len(optional_data)))
            names.insert(name, idx)
        else:
            raise ParseError("unhandled line kind")

    links: List[Tuple[int, int, int, int]] = []
    paths: List[Tuple[int, int, int, int, int, int]] = []
    steps: List[int] = []
    overlaps: List[Tuple[int, int]] = []
    alignment: List[int] = []
    name_data = bytearray()

    for line in deferred:
        if len(line) < 2 or line[1] != 0x09:
            raise ParseError("expected marker and tab")
        rest = line[2:]
        if line[0] == 0x4C:
            fs, r = _parse_num(rest)
            r = _parse_byte(r, 0x09)
            fo, r = _parse_orient(r)
            r = _parse_byte(r, 0x09)
            ts, r = _parse_num(r)
            r = _parse_byte(r, 0x09)
            to, r = _parse_orient(r)
            r = _parse_byte(r, 0x09)
            ops, r = _parse_align(r)
            if r:
                raise ParseError("expected end of line")
            fh = _handle(names.get(fs), fo == 0)
            th = _handle(names.get(ts), to == 0)
            a0 = len(alignment)
            alignment.extend(ops)
            links.append((fh, th, a0, len(alignment)))
        else:
            pname, r = _parse_field(rest)
            psteps, r = _parse_field(r)
            povs = _parse_overlaps(r)
            parsed, srest = _parse_steps(psteps)
            st0 = len(steps)
            for nm, fwd in parsed:
                steps.append(_handle(names.get(nm), fwd))
            if srest:
                raise ParseError("steps: trailing bytes")  # parse.rs:155
            ov0 = len(overlaps)
            for ops in povs:
                a0 = len(alignment)
                alignment.extend(ops)
                overlaps.append((a0, len(alignment)))
            n0 = len(name_data)
            name_data += pname
            paths.append((n0, len(name_data), st0, len(steps), ov0, len(overlaps)))

    def arr(lst, dt):
        a = np.zeros(len(lst), dtype=dt)
        for i, t in enumerate(lst):
            a[i] = t
        return a

    return Pools(
        header=np.frombuffer(bytes(header), dtype="u1").copy(),
        segs=arr(segs, SEG_DT),
        paths=arr(paths, PATH_DT),
        links=arr(links, LINK_DT),
        steps=np.array(steps, dtype="<u4"),
        seq_data=np.frombuffer(bytes(seq_data), dtype="u1").copy(),
        overlaps=arr(overlaps, SPAN_DT),
        alignment=np.array(alignment, dtype="<u4"),
        name_data=np.frombuffer(bytes(name_data), dtype="u1").copy(),
        optional_data=np.frombuffer(bytes(optional_data), dtype="u1").copy(),
        line_order=np.frombuffer(bytes(line_order), dtype="u1").copy(),
    )


# --------------------------------------------------------------------------
# .flatgfa container
# --------------------------------------------------------------------------

def dump_flatgfa(p: Pools) -> bytes:
    """file::dump with Toc::full, file.rs:82-98,290-307: magic, 11 x {len,capacity}
    (capacity == len), then the pools back to back with no padding."""
    toc = [np.uint64(MAGIC).tobytes()]
    body = []
    for name in POOL_ORDER:
        a = getattr(p, name)
        n = int(a.shape[0])
        toc.append(np.array([n, n], dtype="<u8").tobytes())
        body.append(np.ascontiguousarray(a).tobytes())
    return b"".join(toc + body)


def estimate_toc(buf: bytes) -> List[int]:
    """parse.rs:176-216 + Toc::estimate (file.rs:136-158): capacities of the eleven pools, in file
    order, from a scan of the GFA text's line types and lengths."""
    segs = links = paths = header_bytes = seg_bytes = path_bytes = 0
    rest = buf
    while rest:
        marker = rest[0:1]
        nl = rest.find(b"\n")
        nxt = nl if nl >= 0 else len(rest) + 1
        if marker == b"H":
            header_bytes += nxt
        elif marker == b"S":
            segs += 1
            seg_bytes += nxt
        elif marker == b"L":
            links += 1
        elif marker == b"P":
            paths += 1
            path_bytes += nxt
        else:
            raise ParseError("unknown line type")
        if nxt >= len(rest):
            break
        rest = rest[nxt + 1:]
    return [header_bytes, segs, paths, links, path_bytes // 3, seg_bytes, (links + paths) * 2, links * 2 + paths * 4,
            paths * 512, links * 16, segs + links + paths + 8]


def guess_toc(factor: int) -> List[int]:
    """Toc::guess, file.rs:117-132."""
    f = factor
    return [128, 32 * f * f, f, 32 * f * f, 1024 * f * f, 512 * f * f, 256 * f, 64 * f * f, 64 * f, 512 * f * f, 64 * f * f]


def dump_flatgfa_prealloc(p: Pools, caps: List[int]) -> bytes:
    """What prealloc_translate leaves in the file (cli/main.rs:216-248): file::init with the
    estimated capacities (file.rs:255-272), the parse into the fixed-capacity store, then the
    table of contents rewritten with the lengths reached (Toc::for_fixed_store, file.rs:100-114).
    Every region is `capacity` items long; the bytes behind `len` items stay zero (a fresh file)."""
    toc = [np.uint64(MAGIC).tobytes()]
    body = []
    for name, cap in zip(POOL_ORDER, caps):
        a = getattr(p, name)
        n = int(a.shape[0])
        if n > cap:
            raise ParseError(f"{name}: {n} entries do not fit a capacity of {cap}")  # (SliceVec push panics)
        toc.append(np.array([n, cap], dtype="<u8").tobytes())
        raw = np.ascontiguousarray(a).tobytes()
        body.append(raw + bytes(cap * POOL_DTYPES[name].itemsize - len(raw)))
    return b"".join(toc + body)


def view_flatgfa(data: bytes) -> Pools:
    """file::view, file.rs:163-213.  `len` items are taken from each region of
    `capacity` items."""
    if len(data) < 8 + 11 * 16:
        raise ParseError("short file")
    magic = int(np.frombuffer(data, dtype="<u8", count=1)[0])
    if magic != MAGIC:
        raise ParseError("bad magic")
    sizes = np.frombuffer(data, dtype="<u8", count=22, offset=8).reshape(11, 2)
    off = 8 + 11 * 16
    out = {}
    for i, name in enumerate(POOL_ORDER):
        ln, cap = int(sizes[i, 0]), int(sizes[i, 1])
        dt = POOL_DTYPES[name]
        if ln > cap or off + cap * dt.itemsize > len(data):
            raise ParseError("region out of bounds")
        out[name] = np.frombuffer(data, dtype=dt, count=ln, offset=off).copy()
        off += cap * dt.itemsize
    return Pools(**out)


# --------------------------------------------------------------------------
# C oracle (ops/depth.rs) via ctypes
# --------------------------------------------------------------------------

_LIB: Optional[ctypes.CDLL] = None


def build(force: bool = False) -> str:
    """Compile depth_oracle.c into oracle/_build/libdepth_oracle.so."""
    out_dir = os.path.join(HERE, "_build")
    so = os.path.join(out_dir, "libdepth_oracle.so")
    srcs = [os.path.join(HERE, "depth_oracle.c"), os.path.join(HERE, "overlap_oracle.c")]
    srcs = [s for s in srcs if os.path.exists(s)]
    deps = srcs + [os.path.join(HERE, "fgfa_depth_cpu.c")]
    if not force and os.path.exists(so) and os.path.exists(os.path.join(out_dir, "fgfa_depth_cpu")) and all(
            os.path.getmtime(so) >= os.path.getmtime(s) for s in deps if os.path.exists(s)):
        return so
    os.makedirs(out_dir, exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-std=c11", "-fPIC", "-shared", "-pthread",
                           "-Wall", "-Wextra", "-o", so] + srcs)
    # the CPU process bench.py times next to the product's CLI (fgfa_depth_cpu FILE.flatgfa [-d])
    main_c = os.path.join(HERE, "fgfa_depth_cpu.c")
    if os.path.exists(main_c):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-std=c11", "-pthread", "-Wall", "-Wextra", "-o",
                               os.path.join(out_dir, "fgfa_depth_cpu"), main_c, os.path.join(HERE, "depth_oracle.c")])
    return so


def cpu_cli() -> str:
    """Path of the oracle's stand-alone depth process (built with the library)."""
    build()
    return os.path.join(HERE, "_build", "fgfa_depth_cpu")


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        u8p, u32p, u64p, f64p = (ctypes.c_void_p,) * 4
        _LIB.oracle_seg_depth_with_uniq.argtypes = [u8p, ctypes.c_uint64, u32p, ctypes.c_uint64,
                                                     ctypes.c_uint64, u64p, u64p]
        _LIB.oracle_seg_depth.argtypes = [u8p, ctypes.c_uint64, u32p, ctypes.c_uint64,
                                          ctypes.c_uint64, u64p]
        _LIB.oracle_path_depth.argtypes = [u8p, ctypes.c_uint64, u32p, ctypes.c_uint64,
                                           u8p, ctypes.c_uint64, u32p, ctypes.c_uint64, u64p, f64p]
        _LIB.oracle_format_float.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        _LIB.oracle_emit_seg_depth.argtypes = [u8p, ctypes.c_uint64, u64p, u64p,
                                               ctypes.POINTER(ctypes.c_uint64)]
        _LIB.oracle_emit_seg_depth.restype = ctypes.c_void_p
        _LIB.oracle_emit_path_depth.argtypes = [u8p, u8p, u32p, ctypes.c_uint64, u64p, f64p,
                                                ctypes.POINTER(ctypes.c_uint64)]
        _LIB.oracle_emit_path_depth.restype = ctypes.c_void_p
        _LIB.oracle_free.argtypes = [ctypes.c_void_p]
        _LIB.oracle_path_touches.argtypes = [u8p, ctypes.c_uint64, u32p, ctypes.c_uint64, ctypes.c_uint64,
                                             u32p, ctypes.c_uint64, u8p]
        _LIB.oracle_emit_overlap.argtypes = [u8p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64, u64p, u8p,
                                             ctypes.POINTER(ctypes.c_uint64)]
        _LIB.oracle_emit_overlap.restype = ctypes.c_void_p
        _LIB.oracle_interval_depth.argtypes = [u8p, ctypes.c_uint64, u32p, ctypes.c_uint64, u8p, ctypes.c_uint64,
                                               u64p, ctypes.c_uint32, u64p, u64p, ctypes.c_uint64, f64p]
    return _LIB


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _c(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a)


def seg_depth_with_uniq(p: Pools) -> Tuple[np.ndarray, np.ndarray]:
    paths, steps = _c(p.paths), _c(p.steps)
    S = len(p.segs)
    d = np.zeros(S, dtype=np.uint64)
    u = np.zeros(S, dtype=np.uint64)
    rc = lib().oracle_seg_depth_with_uniq(_ptr(paths), len(paths), _ptr(steps), len(steps), S,
                                          _ptr(d), _ptr(u))
    if rc:
        raise ParseError(f"oracle_seg_depth_with_uniq rc={rc}")
    return d, u


def seg_depth_with_uniq_mt(p: Pools, n_threads: int) -> Tuple[np.ndarray, np.ndarray]:
    """Path-parallel variant for the all-host-cores baseline (never the checker)."""
    paths, steps = _c(p.paths), _c(p.steps)
    S = len(p.segs)
    d = np.zeros(S, dtype=np.uint64)
    u = np.zeros(S, dtype=np.uint64)
    fn = lib().oracle_seg_depth_with_uniq_mt
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    rc = fn(_ptr(paths), len(paths), _ptr(steps), len(steps), S, _ptr(d), _ptr(u), int(n_threads))
    if rc:
        raise ParseError(f"oracle_seg_depth_with_uniq_mt rc={rc}")
    return d, u


def seg_depth(p: Pools) -> np.ndarray:
    paths, steps = _c(p.paths), _c(p.steps)
    S = len(p.segs)
    d = np.zeros(S, dtype=np.uint64)
    rc = lib().oracle_seg_depth(_ptr(paths), len(paths), _ptr(steps), len(steps), S, _ptr(d))
    if rc:
        raise ParseError(f"oracle_seg_depth rc={rc}")
    return d


def path_depth(p: Pools, path_ids=None) -> Tuple[np.ndarray, np.ndarray]:
    paths, steps, segs = _c(p.paths), _c(p.steps), _c(p.segs)
    ids = np.arange(len(paths), dtype=np.uint32) if path_ids is None \
        else np.ascontiguousarray(path_ids, dtype=np.uint32)
    ln = np.zeros(len(ids), dtype=np.uint64)
    dp = np.zeros(len(ids), dtype=np.float64)
    rc = lib().oracle_path_depth(_ptr(paths), len(paths), _ptr(steps), len(steps), _ptr(segs),
                                 len(segs), _ptr(ids), len(ids), _ptr(ln), _ptr(dp))
    if rc:
        raise ParseError(f"oracle_path_depth rc={rc}")
    return ln, dp


def format_float(x: float, digits: int) -> str:
    buf = ctypes.create_string_buffer(600)
    n = lib().oracle_format_float(x, digits, buf, 600)
    return buf.raw[:n].decode()


def emit_seg_depth(p: Pools, depths: np.ndarray, uniq: np.ndarray) -> bytes:
    segs = _c(p.segs)
    d = np.ascontiguousarray(depths, dtype=np.uint64)
    u = np.ascontiguousarray(uniq, dtype=np.uint64)
    n = ctypes.c_uint64(0)
    ptr = lib().oracle_emit_seg_depth(_ptr(segs), len(segs), _ptr(d), _ptr(u), ctypes.byref(n))
    out = ctypes.string_at(ptr, n.value)
    lib().oracle_free(ptr)
    return out


def emit_path_depth(p: Pools, path_ids, lengths: np.ndarray, depths: np.ndarray) -> bytes:
    paths, names = _c(p.paths), _c(p.name_data)
    ids = np.ascontiguousarray(path_ids, dtype=np.uint32)
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    dp = np.ascontiguousarray(depths, dtype=np.float64)
    n = ctypes.c_uint64(0)
    ptr = lib().oracle_emit_path_depth(_ptr(paths), _ptr(names), _ptr(ids), len(ids), _ptr(ln),
                                       _ptr(dp), ctypes.byref(n))
    out = ctypes.string_at(ptr, n.value)
    lib().oracle_free(ptr)
    return out


def find_path(p: Pools, name: bytes) -> Optional[int]:
    """FlatGFA::find_path, flatgfa.rs:387-389 (first match, linear)."""
    for i in range(len(p.paths)):
        if p.path_name(i) == name:
            return i
    return None


def fgfa_depth(p: Pools, seg_depth_flag: bool = False, path_names: Optional[List[bytes]] = None) -> bytes:
    """cmds::depth, cli/cmds.rs:234-285 (without -b): the bytes `fgfa depth` prints."""
    if seg_depth_flag:
        d, u = seg_depth_with_uniq(p)
        return emit_seg_depth(p, d, u)
    if not path_names:
        ids = np.arange(len(p.paths), dtype=np.uint32)
    else:
        found = [find_path(p, n) for n in path_names]
        ids = np.array([i for i in found if i is not None], dtype=np.uint32)
    ln, dp = path_depth(p, ids)
    return emit_path_depth(p, ids, ln, dp)


# --------------------------------------------------------------------------
# rows next to the depth path: overlap (slow_odgi/overlap.py), subset depth (slow_odgi/depth.py:12),
# window / interval depth (flatgfa/src/ops/window_depth.rs)
# --------------------------------------------------------------------------

def path_touches(p: Pools, query_ids) -> np.ndarray:
    paths, steps = _c(p.paths), _c(p.steps)
    ids = np.ascontiguousarray(query_ids, dtype=np.uint32)
    out = np.zeros((len(ids), len(paths)), dtype=np.uint8)
    rc = lib().oracle_path_touches(_ptr(paths), len(paths), _ptr(steps), len(steps), len(p.segs), _ptr(ids), len(ids),
                                   _ptr(out) if out.size else None)
    if rc:
        raise ParseError(f"oracle_path_touches rc={rc}")
    return out


def overlap_table(p: Pools, query_names: List[bytes]) -> bytes:
    """The bytes `slow_odgi overlap --paths FILE` prints."""
    ids = []
    for nm in query_names:
        i = find_path(p, nm)
        if i is None:
            raise ParseError("query path not in graph")  # overlap.py:21 asserts
        ids.append(i)
    ids = np.array(ids, dtype=np.uint32)
    touch = path_touches(p, ids)
    ln, _ = path_depth(p, ids)  # len(pathseq[ip]) == the path's length in base pairs
    paths, names = _c(p.paths), _c(p.name_data)
    n = ctypes.c_uint64(0)
    ptr = lib().oracle_emit_overlap(_ptr(paths), _ptr(names), len(paths), _ptr(ids), len(ids), _ptr(ln),
                                    _ptr(touch) if touch.size else None, ctypes.byref(n))
    out = ctypes.string_at(ptr, n.value)
    lib().oracle_free(ptr)
    return out


def seg_depth_subset(p: Pools, path_ids) -> Tuple[np.ndarray, np.ndarray]:
    """slow_odgi/depth.py:12: only crossings on the listed paths count (each listed path once per listing)."""
    sub = Pools(**{n: getattr(p, n) for n in POOL_ORDER})
    sub.paths = np.ascontiguousarray(p.paths[np.asarray(path_ids, dtype=np.intp)])
    return seg_depth_with_uniq(sub)


def interval_depth(p: Pools, path_id: int, starts, ends) -> np.ndarray:
    paths, steps, segs = _c(p.paths), _c(p.steps), _c(p.segs)
    d = seg_depth(p)
    st = np.ascontiguousarray(starts, dtype=np.uint64)
    en = np.ascontiguousarray(ends, dtype=np.uint64)
    out = np.zeros(len(st), dtype=np.float64)
    rc = lib().oracle_interval_depth(_ptr(paths), len(paths), _ptr(steps), len(steps), _ptr(segs), len(segs), _ptr(d),
                                     int(path_id), _ptr(st), _ptr(en), len(st), _ptr(out))
    if rc:
        raise ParseError(f"oracle_interval_depth rc={rc}")
    return out


def parse_bed(buf: bytes):
    """flatbed.rs:125-158 (+ memfile.rs:51-63): [(name, start, end)]."""
    out = []
    for line in _memchr_split(buf):
        if line.startswith(b"#"):
            continue
        name, rest = _parse_field(line)
        start, rest = _parse_num(rest)
        if not rest:
            raise ParseError("BED: index out of bounds")
        end, _ = _parse_num(rest[1:])
        out.append((name, start, end))
    return out


def _emit_intervals(rows, depths) -> bytes:
    """IntervalDepth::emit, window_depth.rs:158-170"""
    return b"".join(b"%s\t%d\t%d\t%s\n" % (nm, s, e, format_float(float(d), 4).encode())
                    for (nm, s, e), d in zip(rows, depths))


def window_depth_table(p: Pools, path_name: bytes, window: int) -> bytes:
    """cmds::window_depth, cli/cmds.rs:488-496 + window_depth.rs:183-200"""
    pid = find_path(p, path_name)
    if pid is None or window <= 0:
        raise ParseError("path not found / bad window")
    ln, _ = path_depth(p, [pid])
    rows, pos = [], 0
    while pos < int(ln[0]):
        e = min(pos + window, int(ln[0]))
        rows.append((path_name, pos, e))
        pos = e
    return _emit_intervals(rows, interval_depth(p, pid, [r[1] for r in rows], [r[2] for r in rows]))


def bed_depth_table(p: Pools, bed: bytes) -> bytes:
    """cmds::depth -b, cli/cmds.rs:246-255 + window_depth.rs:203-211"""
    rows = parse_bed(bed)
    if not rows:
        raise ParseError("BED: no intervals")
    pid = find_path(p, rows[0][0])
    if pid is None:
        raise ParseError("path not found in graph")
    return _emit_intervals(rows, interval_depth(p, pid, [r[1] for r in rows], [r[2] for r in rows]))
