"""List-like views over a FlatGFA, mirroring the reference's Python bindings
(cucapra/pollen flatgfa-py/flatgfa.pyi:4-93, behaviour per flatgfa-py/test/test_flatgfa.py):
``graph.segments``, ``graph.paths``, ``graph.links`` act like lists (len, index, slice, iterate,
``find``); a path acts like a list of step handles; items are equatable, hashable and print as
their GFA line.  Read-only host-side accessors over the pools -- no device work here.
"""
from __future__ import annotations

from typing import Iterator, Optional, Union

import numpy as np

_ALIGN_LETTER = "MNDI"  # print.rs:14-23


class _Pools:
    """Lazily fetched copies of the pools a view needs."""

    def __init__(self, graph):
        self.graph = graph
        self._cache = {}

    def __getattr__(self, name):
        if name.startswith("_") or name == "graph":
            raise AttributeError(name)
        if name not in self._cache:
            self._cache[name] = self.graph.pool(name)
        return self._cache[name]


def _alignment(pools: _Pools, start: int, end: int) -> str:
    if start == end:
        return "0M"
    return "".join(f"{int(op) >> 8}{_ALIGN_LETTER[int(op) & 3]}" for op in pools.alignment[start:end])


class Segment:
    def __init__(self, pools: _Pools, idx: int):
        self._p, self.id = pools, int(idx)

    @property
    def name(self) -> int:
        return int(self._p.segs[self.id]["name"])

    def sequence(self) -> bytes:
        s = self._p.segs[self.id]
        return self._p.seq_data[int(s["seq_start"]):int(s["seq_end"])].tobytes()

    def __len__(self) -> int:
        s = self._p.segs[self.id]
        return int(s["seq_end"]) - int(s["seq_start"])

    def __eq__(self, other):
        return isinstance(other, Segment) and other._p.graph is self._p.graph and other.id == self.id

    def __hash__(self):
        return hash(("seg", id(self._p.graph), self.id))

    def __str__(self) -> str:  # print.rs:86-96
        s = self._p.segs[self.id]
        out = f"S\t{self.name}\t{self.sequence().decode(errors='replace')}"
        if s["opt_end"] != s["opt_start"]:
            out += "\t" + self._p.optional_data[int(s["opt_start"]):int(s["opt_end"])].tobytes().decode(errors="replace")
        return out


class Handle:
    """An oriented segment reference; has no identity of its own (flatgfa.rs:186-209)."""

    def __init__(self, pools: _Pools, bits: int):
        self._p, self._bits = pools, int(bits)

    @property
    def seg_id(self) -> int:
        return self._bits >> 1

    @property
    def segment(self) -> Segment:
        return Segment(self._p, self._bits >> 1)

    @property
    def is_forward(self) -> bool:
        return (self._bits & 1) == 0

    def __eq__(self, other):
        return isinstance(other, Handle) and other._p.graph is self._p.graph and other._bits == self._bits

    def __hash__(self):
        return hash(("handle", id(self._p.graph), self._bits))

    def __str__(self) -> str:  # print.rs:37-43
        return f"{self.segment.name}{'+' if self.is_forward else '-'}"


class StepList:
    def __init__(self, pools: _Pools, start: int, end: int):
        self._p, self._s, self._e = pools, int(start), int(end)

    def __len__(self) -> int:
        return self._e - self._s

    def __iter__(self) -> Iterator[Handle]:
        for b in self._p.steps[self._s:self._e]:
            yield Handle(self._p, b)

    def __getitem__(self, i: Union[int, slice]):
        if isinstance(i, slice):
            a, b, st = i.indices(len(self))
            if st != 1:
                raise ValueError("step slices must be contiguous")
            return StepList(self._p, self._s + a, self._s + max(a, b))
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("step index out of range")
        return Handle(self._p, self._p.steps[self._s + i])


class Path:
    def __init__(self, pools: _Pools, idx: int):
        self._p, self.id = pools, int(idx)

    def _row(self):
        return self._p.paths[self.id]

    @property
    def name(self) -> str:
        r = self._row()
        return self._p.name_data[int(r["name_start"]):int(r["name_end"])].tobytes().decode(errors="replace")

    def _steps(self) -> StepList:
        r = self._row()
        return StepList(self._p, r["steps_start"], r["steps_end"])

    def __len__(self) -> int:
        return len(self._steps())

    def __iter__(self) -> Iterator[Handle]:
        return iter(self._steps())

    def __getitem__(self, i):
        return self._steps()[i]

    def __eq__(self, other):
        return isinstance(other, Path) and other._p.graph is self._p.graph and other.id == self.id

    def __hash__(self):
        return hash(("path", id(self._p.graph), self.id))

    def __str__(self) -> str:  # print.rs:45-66
        r = self._row()
        steps = ",".join(str(h) for h in self)
        ov = self._p.overlaps[int(r["ov_start"]):int(r["ov_end"])]
        ovs = "*" if len(ov) == 0 else ",".join(_alignment(self._p, int(o["start"]), int(o["end"])) for o in ov)
        return f"P\t{self.name}\t{steps}\t{ovs}"


class Link:
    def __init__(self, pools: _Pools, idx: int):
        self._p, self.id = pools, int(idx)

    @property
    def from_(self) -> Handle:
        return Handle(self._p, self._p.links[self.id]["from_"])

    @property
    def to(self) -> Handle:
        return Handle(self._p, self._p.links[self.id]["to"])

    def __eq__(self, other):
        return isinstance(other, Link) and other._p.graph is self._p.graph and other.id == self.id

    def __hash__(self):
        return hash(("link", id(self._p.graph), self.id))

    def __str__(self) -> str:  # print.rs:68-84
        r = self._p.links[self.id]
        f, t = self.from_, self.to
        return (f"L\t{f.segment.name}\t{'+' if f.is_forward else '-'}\t{t.segment.name}\t"
                f"{'+' if t.is_forward else '-'}\t{_alignment(self._p, int(r['ov_start']), int(r['ov_end']))}")


class _ListView:
    _item = None
    _pool = ""

    def __init__(self, pools: _Pools, start: int = 0, end: Optional[int] = None):
        self._p = pools
        self._s = start
        self._e = len(getattr(pools, self._pool)) if end is None else end

    def __len__(self) -> int:
        return self._e - self._s

    def __iter__(self):
        for i in range(self._s, self._e):
            yield self._item(self._p, i)

    def __getitem__(self, i):
        if isinstance(i, slice):
            a, b, st = i.indices(len(self))
            if st != 1:
                raise ValueError("slices must be contiguous")
            return type(self)(self._p, self._s + a, self._s + max(a, b))
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("index out of range")
        return self._item(self._p, self._s + i)


class SegmentList(_ListView):
    _item, _pool = Segment, "segs"

    def find(self, name: int) -> Optional[Segment]:  # FlatGFA::find_seg, flatgfa.rs:380-384 (linear, first match)
        hits = np.nonzero(self._p.segs["name"][self._s:self._e] == np.uint64(name))[0]
        return Segment(self._p, self._s + int(hits[0])) if len(hits) else None


class PathList(_ListView):
    _item, _pool = Path, "paths"

    def find(self, name: Union[str, bytes]) -> Optional[Path]:  # FlatGFA::find_path, flatgfa.rs:387-389
        want = name.encode() if isinstance(name, str) else bytes(name)
        for p in self:
            r = p._row()
            if self._p.name_data[int(r["name_start"]):int(r["name_end"])].tobytes() == want:
                return p
        return None


class LinkList(_ListView):
    _item, _pool = Link, "links"
