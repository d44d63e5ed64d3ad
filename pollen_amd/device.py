"""Device-level depth queries on caller-owned HBM buffers (Part 3 of include/flatgfa.h).

torch is used for what it is good at here -- device memory, streams, torch.distributed -- and
nothing else: the tensors below are plain int32 buffers whose ``data_ptr()`` goes straight
into the C ABI; every kernel that runs is a hand-written HIP kernel in libflatgfa.so.

Handles are u32 in the reference (flatgfa.rs:186-209); they are carried in torch.int32
tensors because u32 and i32 have the same bits and wrapping add, and int32 is what RCCL and
every torch op accept.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .flatgfa import FlatGFAError, _check


def _torch():
    import torch
    return torch


def _as_i32(a: np.ndarray):
    torch = _torch()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32))


class DeviceGraph:
    """The structure-of-arrays graph image resident in one GPU's HBM."""

    def __init__(self, steps: np.ndarray, path_begin: np.ndarray, path_end: np.ndarray, n_segs: int,
                 seg_len: Optional[np.ndarray] = None, device: str = "cuda:0"):
        torch = _torch()
        if not torch.cuda.is_available():
            raise FlatGFAError("no HIP device is visible; pollen_amd has no CPU fallback", -3)
        self.device = torch.device(device)
        self.n_steps = int(len(steps))
        self.n_paths = int(len(path_begin))
        self.n_segs = int(n_segs)
        self.h_path_begin = np.ascontiguousarray(path_begin, dtype=np.uint32)
        self.h_path_end = np.ascontiguousarray(path_end, dtype=np.uint32)
        with torch.cuda.device(self.device):
            self.steps = _as_i32(steps).to(self.device)
            self.path_begin = _as_i32(self.h_path_begin).to(self.device)
            self.path_end = _as_i32(self.h_path_end).to(self.device)
            self.seg_len = None if seg_len is None else _as_i32(seg_len).to(self.device)
            torch.cuda.synchronize(self.device)

    def c_struct(self) -> _lib.flatgfa_dev_graph_t:
        return _lib.flatgfa_dev_graph_t(
            self.steps.data_ptr() if self.n_steps else None, self.n_steps,
            self.path_begin.data_ptr() if self.n_paths else None,
            self.path_end.data_ptr() if self.n_paths else None, self.n_paths, self.n_segs,
            self.seg_len.data_ptr() if self.seg_len is not None and self.n_segs else None)


class DepthPlan:
    """A prepared depth query (flatgfa_dev_plan_t): launch plan + scratch HBM for one DeviceGraph.

    The plan is laid out for the step values it was made with: a caller that changes ``graph.steps`` (a mutable
    tensor) under a live plan calls :meth:`steps_changed` before the next query -- a path the plan found strictly
    monotone is counted without the per-path "seen" set, and nothing re-checks that per call (``FLATGFA_CHECK_NO_CLAIM=1``
    in the environment does, as a debugging aid: ``status()`` then raises with code -8).

    ``first=(depth_out, uniq_out)`` (int32 CUDA tensors of n_segs elements; uniq_out may be None): the query that sizes
    the plan's scratch writes the caller's buffers -- creating the plan is the first query (flatgfa_dev_plan_create_first);
    the tensors are complete when the constructor returns, and ``first_status`` is 0 or -2 (an id out of range)."""

    def __init__(self, graph: DeviceGraph, first=None):
        torch = _torch()
        self.graph = graph
        self.first_status = 0
        with torch.cuda.device(graph.device):
            g = graph.c_struct()
            hb = graph.h_path_begin.ctypes.data if graph.n_paths else None
            he = graph.h_path_end.ctypes.data if graph.n_paths else None
            if first is None:
                self._p = ctypes.c_void_p(_lib.lib().flatgfa_dev_plan_create(ctypes.byref(g), hb, he))
            else:
                d, u = first
                S = graph.n_segs
                for t in (d, u):
                    if t is not None:
                        assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous() and t.numel() == S
                torch.cuda.current_stream(graph.device).synchronize()  # (the plan's creation runs on the null stream: whatever filled the buffers is done first)
                st = ctypes.c_int(0)
                self._p = ctypes.c_void_p(_lib.lib().flatgfa_dev_plan_create_first(
                    ctypes.byref(g), hb, he, d.data_ptr() if S else None, u.data_ptr() if (u is not None and S) else None, ctypes.byref(st)))
                self.first_status = int(st.value)
        if not self._p.value:
            raise FlatGFAError("dev_plan_create", -2)

    def steps_changed(self) -> None:
        """The values of ``graph.steps`` were changed (not the spans): make the plan again from the steps as they are
        (flatgfa_dev_plan_steps_changed; waits for torch's current stream first)."""
        with _torch().cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_plan_steps_changed(self._p, self._stream()), "dev_plan_steps_changed")

    def close(self) -> None:
        if getattr(self, "_p", None) is not None and self._p.value:
            _lib.lib().flatgfa_dev_plan_destroy(self._p)
            self._p = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> int:
        return _torch().cuda.current_stream(self.graph.device).cuda_stream

    def seg_depth(self, depth_out, uniq_out=None) -> None:
        """Enqueue node depth (+ unique depth when `uniq_out` is given) on torch's current stream.
        Outputs are int32 CUDA tensors of n_segs elements (u32 bits)."""
        torch = _torch()
        S = self.graph.n_segs
        for t in (depth_out, uniq_out):
            if t is not None:
                assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous() and t.numel() == S
        with torch.cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_seg_depth(self._p, depth_out.data_ptr() if S else None,
                                                    uniq_out.data_ptr() if (uniq_out is not None and S) else None,
                                                    self._stream()), "dev_seg_depth")

    def seg_depth_call(self, depth_out, uniq_out=None, stream=None):
        """A prepared call: returns a function of no arguments that enqueues node depth (+ unique depth) into the given
        tensors on `stream` (a torch.cuda.Stream; None = the stream that is current NOW).  Everything is checked and
        resolved here, once; the function itself is one call through the C ABI -- what a loop that keeps several
        calls in flight wants (the host has 60 us per kernel launch to stay ahead of the device)."""
        torch = _torch()
        S = self.graph.n_segs
        for t in (depth_out, uniq_out):
            if t is not None:
                assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous() and t.numel() == S
        fn = _lib.lib().flatgfa_dev_seg_depth
        p = self._p
        d = ctypes.c_void_p(depth_out.data_ptr() if S else None)
        u = ctypes.c_void_p(uniq_out.data_ptr() if (uniq_out is not None and S) else None)
        with torch.cuda.device(self.graph.device):
            h = ctypes.c_void_p((stream if stream is not None else torch.cuda.current_stream(self.graph.device)).cuda_stream)
        keep = (depth_out, uniq_out, stream)  # (the tensors live as long as the call does)

        def call(_keep=keep):
            rc = fn(p, d, u, h)
            if rc:
                _check(rc, "dev_seg_depth")
        return call

    def path_sums(self, path_ids, depth, length_out, weighted_out) -> None:
        """Enqueue measure_path's integer sums for `path_ids` (int32 CUDA tensor); outputs are
        int64 CUDA tensors (u64 bits)."""
        torch = _torch()
        n = int(path_ids.numel())
        assert path_ids.dtype == torch.int32 and length_out.dtype == torch.int64 and weighted_out.dtype == torch.int64
        assert length_out.numel() == n and weighted_out.numel() == n
        with torch.cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_path_sums(self._p, path_ids.data_ptr() if n else None, n,
                                                    depth.data_ptr() if self.graph.n_segs else None,
                                                    length_out.data_ptr() if n else None,
                                                    weighted_out.data_ptr() if n else None, self._stream()),
                   "dev_path_sums")

    def path_depth_all(self, depth_out, length_out, weighted_out) -> None:
        """Enqueue node depth AND measure_path's two integer sums for every path of the graph
        (what `fgfa depth` needs) in one pass over the steps.  depth_out: int32[n_segs]; the sums:
        int64[n_paths] CUDA tensors (u64 bits), indexed by path."""
        torch = _torch()
        S, P = self.graph.n_segs, self.graph.n_paths
        assert depth_out.dtype == torch.int32 and depth_out.numel() == S and depth_out.is_contiguous()
        for t in (length_out, weighted_out):
            assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous() and t.numel() == P
        with torch.cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_path_depth_all(self._p, depth_out.data_ptr() if S else None,
                                                         length_out.data_ptr() if P else None,
                                                         weighted_out.data_ptr() if P else None, self._stream()),
                   "dev_path_depth_all")

    def path_overlaps(self, query_ids, touch_out) -> None:
        """Enqueue path-pair overlap (slow_odgi/overlap.py:6-14): touch_out[k * n_paths + j] = 1 iff path
        j touches path query_ids[k].  query_ids: int32[n_q], touch_out: uint8[n_q * n_paths], CUDA tensors."""
        torch = _torch()
        n = int(query_ids.numel())
        assert query_ids.dtype == torch.int32 and touch_out.dtype == torch.uint8 and touch_out.numel() == n * self.graph.n_paths
        with torch.cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_path_overlaps(self._p, query_ids.data_ptr() if n else None, n,
                                                        touch_out.data_ptr() if touch_out.numel() else None,
                                                        self._stream()), "dev_path_overlaps")

    def describe(self) -> str:
        """Which kernels this plan's calls run (the choices made, some by timing, when it was created)."""
        buf = ctypes.create_string_buffer(512)
        _lib.lib().flatgfa_dev_plan_describe(self._p, buf, 512)
        return buf.value.decode()

    def status(self) -> None:
        """Synchronize the current stream and raise if a kernel saw an out-of-range id."""
        with _torch().cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_status(self._p, self._stream()), "dev_status")


class DepthPipeline:
    """Calls in flight (flatgfa_dev_pipeline_t): K plans of one resident graph on K internal streams, taken in turn, so
    that pass 2 of one call shares the chip with pass 1 of the next.  `seg_depth` enqueues and returns; the buffers of a
    call may be read after `join()` (torch's current stream then waits for every call so far) or `status()`.
    As with a DepthPlan, the lanes' plans are laid out for the step values they were made with: after writing to
    ``graph.steps`` call :meth:`steps_changed`."""

    def __init__(self, graph: DeviceGraph, calls_in_flight: int = 2):
        torch = _torch()
        self.graph = graph
        self.calls_in_flight = int(calls_in_flight)
        with torch.cuda.device(graph.device):
            g = graph.c_struct()
            self._p = ctypes.c_void_p(_lib.lib().flatgfa_dev_pipeline_create(
                ctypes.byref(g), graph.h_path_begin.ctypes.data if graph.n_paths else None,
                graph.h_path_end.ctypes.data if graph.n_paths else None, self.calls_in_flight))
        if not self._p.value:
            raise FlatGFAError("dev_pipeline_create", -2)

    def close(self) -> None:
        if getattr(self, "_p", None) is not None and self._p.value:
            _lib.lib().flatgfa_dev_pipeline_destroy(self._p)
            self._p = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seg_depth(self, depth_out, uniq_out=None, after_current_stream: bool = True) -> None:
        """Enqueue node depth (+ unique depth) on the pipeline's next lane.  With `after_current_stream` the call first
        waits for what torch's current stream holds so far (whoever filled the inputs or read these buffers last)."""
        torch = _torch()
        S = self.graph.n_segs
        for t in (depth_out, uniq_out):
            if t is not None:
                assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous() and t.numel() == S
        with torch.cuda.device(self.graph.device):
            after = ctypes.c_void_p(torch.cuda.current_stream(self.graph.device).cuda_stream) if after_current_stream else ctypes.c_void_p(-1)
            _check(_lib.lib().flatgfa_dev_pipeline_seg_depth(self._p, depth_out.data_ptr() if S else None,
                                                             uniq_out.data_ptr() if (uniq_out is not None and S) else None, after),
                   "dev_pipeline_seg_depth")

    def path_depth_all(self, depth_out, length_out, weighted_out, after_current_stream: bool = True) -> None:
        """Enqueue node depth and measure_path's two sums for every path (what `fgfa depth` needs) on the next lane."""
        torch = _torch()
        S, P = self.graph.n_segs, self.graph.n_paths
        assert depth_out.dtype == torch.int32 and depth_out.numel() == S and depth_out.is_contiguous()
        for t in (length_out, weighted_out):
            assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous() and t.numel() == P
        with torch.cuda.device(self.graph.device):
            after = ctypes.c_void_p(torch.cuda.current_stream(self.graph.device).cuda_stream) if after_current_stream else ctypes.c_void_p(-1)
            _check(_lib.lib().flatgfa_dev_pipeline_path_depth_all(self._p, depth_out.data_ptr() if S else None, length_out.data_ptr() if P else None,
                                                                  weighted_out.data_ptr() if P else None, after), "dev_pipeline_path_depth_all")

    def steps_changed(self) -> None:
        """The values of ``graph.steps`` were changed: every lane's plan is made again (flatgfa_dev_pipeline_steps_changed)."""
        with _torch().cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_pipeline_steps_changed(self._p), "dev_pipeline_steps_changed")

    def join(self) -> None:
        """torch's current stream waits for every call enqueued so far (no host wait)."""
        torch = _torch()
        with torch.cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_pipeline_join(self._p, ctypes.c_void_p(torch.cuda.current_stream(self.graph.device).cuda_stream)),
                   "dev_pipeline_join")

    def status(self) -> None:
        """Wait for every lane and raise if a kernel saw an out-of-range id."""
        with _torch().cuda.device(self.graph.device):
            _check(_lib.lib().flatgfa_dev_pipeline_status(self._p), "dev_pipeline_status")

    def describe(self) -> str:
        buf = ctypes.create_string_buffer(1024)
        _lib.lib().flatgfa_dev_pipeline_describe(self._p, buf, 1024)
        return buf.value.decode()


def profile_enable(on: bool) -> None:
    _lib.lib().flatgfa_dev_profile_enable(1 if on else 0)


def profile_overhead_ms(n_workgroups: int = 256, lds_bytes: int = 142 * 1024, reps: int = 20) -> float:
    """What the event pair of a profiled kernel reads around a launch that does nothing."""
    import torch
    return float(_lib.lib().flatgfa_dev_profile_overhead_ms(n_workgroups, lds_bytes, reps, torch.cuda.current_stream().cuda_stream))


def profile_read(cap: int = 4096) -> List[Tuple[str, float]]:
    names = (ctypes.c_char_p * cap)()
    ms = (ctypes.c_float * cap)()
    n = _lib.lib().flatgfa_dev_profile_read(names, ms, cap)
    return [(names[i].decode(), float(ms[i])) for i in range(n)]
