"""Path-sharded node depth across the GPUs of one node.

depth and uniq are sums over paths of per-path contributions (ops/depth.rs:25-36), and paths
are independent, so each rank owns a contiguous group of WHOLE paths (never splitting a path
keeps uniq a plain sum), computes its partial vectors with the HIP kernels, and one sum
all-reduce of the per-segment vector(s) combines them -- RCCL over xGMI under
torch.distributed's "nccl" backend.  The message is only 4*S (or 8*S) bytes, so the collective
is latency-bound and is issued once, on the fused [depth | uniq] buffer.

This module never cuts a path (a graph with fewer paths than ranks leaves ranks idle): the C ABI's
`flatgfa_sharded_*` (pollen_amd/csrc/sharded.hip; `pollen_amd.ShardedFlatGFA`) cuts inside a path
where no path boundary is near the even cut, and fixes unique depth up for the cut paths.

Nothing here computes depth on the host: `local_fn` is the HIP path in production
(DepthPlan.seg_depth); tests inject a stand-in to exercise the partition + reduce logic on CPU
under gloo.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np


def shard_paths(path_begin: np.ndarray, path_end: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Cut paths [0, P) into `world` contiguous groups of whole paths with near-equal step
    counts.  Returns [(lo, hi)] per rank (hi exclusive; groups may be empty).

    Greedy on the prefix sum: rank r ends at the first path whose cumulative step count
    reaches (r+1)/world of the total, which bounds every rank's load by ideal + one path."""
    P = int(len(path_begin))
    lens = (np.asarray(path_end, dtype=np.int64) - np.asarray(path_begin, dtype=np.int64))
    if (lens < 0).any():
        raise ValueError("reversed path span")
    # ends[k] = steps in paths [0, k); a cut at k puts ends[k] steps to its left
    ends = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(ends[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world  # (the even cut, in whole steps: flatgfa_sharded_create computes the same -- the two routes agree wherever that one cuts between paths)
        k = int(np.searchsorted(ends, target, side="left"))  # first k with ends[k] >= target
        k = min(k, P)
        if k > 0 and abs(int(ends[k - 1]) - target) <= abs(int(ends[k]) - target):
            k -= 1  # the cut just before is at least as close to the target
        cuts.append(min(max(k, cuts[-1]), P))
    cuts.append(P)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def local_slice(steps: np.ndarray, path_begin: np.ndarray, path_end: np.ndarray, lo: int, hi: int
                ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The sub-image rank-local kernels read: only the steps its paths span, with rebased spans.
    Works for arbitrary (even overlapping) spans by taking the covering range."""
    if hi <= lo:
        z = np.zeros(0, dtype=np.uint32)
        return z, z.copy(), z.copy()
    b = np.asarray(path_begin[lo:hi], dtype=np.uint32)
    e = np.asarray(path_end[lo:hi], dtype=np.uint32)
    s0, s1 = int(b.min()), int(e.max())
    return (np.ascontiguousarray(steps[s0:s1]), (b - np.uint32(s0)).astype(np.uint32),
            (e - np.uint32(s0)).astype(np.uint32))


class ShardedDepth:
    """seg_depth_with_uniq over a path-sharded graph: local HIP partials + one sum all-reduce.

    Steps are pipelined over `depth_of_pipeline` result buffers: `run()` enqueues the local kernels
    of a step into the next buffer and starts its all-reduce asynchronously (RCCL runs it on its
    own stream, after the kernels), so the collective of step i overlaps the kernels of step
    i + 1.  `finish()` waits for everything in flight; `depth` / `uniq` / `buf` refer to the most
    recent step and are complete after `finish()`.

    **Calls in flight.**  `local_fn` may be a LIST of K functions with a list of K `streams` (one
    DepthPlan of the same resident graph per stream: a plan belongs to one stream): a step into buffer j runs
    `local_fn[j % K]` on `streams[j % K]`, and the kernels of consecutive steps
    share the chip -- pass 2 of one call (bound by instruction issue) runs on the CUs that pass 1
    of the next (bound by the memory system) does not occupy, and neither kernel's tail leaves CUs
    idle.  Every step is still a whole query into its own buffer."""

    def __init__(self, n_segs: int, local_fn, device, with_uniq: bool = True, group=None,
                 depth_of_pipeline: int = 2, streams=None):
        import torch
        self.torch = torch
        self.n_segs = int(n_segs)
        self.local_fns = list(local_fn) if isinstance(local_fn, (list, tuple)) else [local_fn]
        self.streams = list(streams) if streams else None
        if self.streams is not None and len(self.streams) != len(self.local_fns):
            raise ValueError("one stream per local_fn")
        self.with_uniq = with_uniq
        self.group = group
        k = 2 if with_uniq else 1
        n_bufs = max(1, int(depth_of_pipeline))
        if len(self.local_fns) > 1:  # a buffer per call in flight, and then some for the collectives behind them
            n_bufs = -(-max(n_bufs, len(self.local_fns)) // len(self.local_fns)) * len(self.local_fns)
        # one fused buffer per step in flight, so that a single collective carries both vectors
        self.bufs = [torch.zeros(k * self.n_segs, dtype=torch.int32, device=device) for _ in range(n_bufs)]
        # The zero-fills run on torch's current stream; the buffers' first writers may run on side streams (`streams`) or
        # on a stream the prepared calls were resolved for: those wait for the fills here, once (no host wait).
        if self.streams is not None and torch.device(device).type == "cuda":
            cur = torch.cuda.current_stream(torch.device(device))
            for s in self.streams:
                s.wait_stream(cur)
        self.works = [None] * len(self.bufs)
        self.cur = 0
        self.step = 0
        self._prepared = None  # world 1: one prepared call per buffer (prepare())

    def prepare(self, plans) -> None:
        """World size 1 with DepthPlans (one per local_fn): resolve every buffer's call once (DepthPlan.seg_depth_call),
        so that run() is a single call through the C ABI -- a loop that keeps calls in flight has to enqueue
        two kernels in less time than the device takes to run them."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return
        k = len(self.local_fns)
        assert len(plans) == k
        self._prepared = [plans[j % k].seg_depth_call(b[: self.n_segs], b[self.n_segs:] if self.with_uniq else None,
                                                       self.streams[j % k] if self.streams is not None else None)
                          for j, b in enumerate(self.bufs)]

    @property
    def local_fn(self):
        return self.local_fns[0]

    @property
    def buf(self):
        return self.bufs[self.cur]

    @property
    def depth(self):
        return self.buf[: self.n_segs]

    @property
    def uniq(self):
        return self.buf[self.n_segs:] if self.with_uniq else None

    def _enqueue(self, j: int, which: int) -> None:
        import torch.distributed as dist
        if self.works[j] is not None:  # the collective that last used this buffer
            self.works[j].wait()
            self.works[j] = None
        b = self.bufs[j]
        self.local_fns[which](b[: self.n_segs], b[self.n_segs:] if self.with_uniq else None)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self.works[j] = dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def run(self) -> None:
        """One step: local partials into the next buffer, then its all-reduce (if world > 1)."""
        j = (self.cur + 1) % len(self.bufs)
        if self._prepared is not None:
            self._prepared[j]()
            self.cur = j
            self.step += 1
            return
        which = j % len(self.local_fns)  # (the buffers are a multiple of the functions: a buffer always meets the same plan and stream)
        if self.streams is not None:
            with self.torch.cuda.stream(self.streams[which]):
                self._enqueue(j, which)
        else:
            self._enqueue(j, which)
        self.cur = j
        self.step += 1

    def finish(self) -> None:
        """Wait (the current stream, for RCCL and for the side streams) for everything still in flight."""
        for k, w in enumerate(self.works):
            if w is not None:
                if self.streams is not None:
                    with self.torch.cuda.stream(self.streams[k % len(self.streams)]):
                        w.wait()
                else:
                    w.wait()
                self.works[k] = None
        if self.streams is not None:
            cur = self.torch.cuda.current_stream()
            for s in self.streams:
                cur.wait_stream(s)


def gather_path_depth(local_lengths, local_weighted, group=None) -> Tuple[np.ndarray, np.ndarray]:
    """path_depth over a path-sharded graph: every rank measures ITS paths
    against the all-reduced node depth (DepthPlan.path_sums over its local path ids gives the two
    integer sums of measure_path, ops/depth.rs:116-131), and since ranks own contiguous groups of
    whole paths the per-path results are disjoint: an all-gather in rank order concatenates them,
    no reduction.  Takes this rank's `lengths` and `weighted` sums (int64 tensors of u64 bits, or
    numpy arrays); returns (lengths u64[P], means f64[P]) for all paths in path order, the one f64
    division per path done exactly as depth.rs:129 does it."""
    import torch
    import torch.distributed as dist
    ln = np.ascontiguousarray(local_lengths.cpu().numpy() if hasattr(local_lengths, "cpu") else local_lengths).view(np.uint64)
    ws = np.ascontiguousarray(local_weighted.cpu().numpy() if hasattr(local_weighted, "cpu") else local_weighted).view(np.uint64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, (ln, ws), group=group)  # P integers per rank: tiny, host-side
        ln = np.concatenate([p[0] for p in parts]) if parts else ln
        ws = np.concatenate([p[1] for p in parts]) if parts else ws
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = ws.astype(np.float64) / ln.astype(np.float64)  # 0/0 -> NaN, as the reference prints it
    return ln, mean
