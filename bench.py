#!/usr/bin/env python3
"""bench.py -- path-steps/sec of node depth (+ unique depth) on the 1M-segment / 100M-step
synthetic graph (BASELINE.json metric; configs[2] at N=1, configs[3] at N>1).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfgL|...] [--scaling strong|weak]
                  [--rotate K]

One "step" = one pass of the hot path (seg_depth_with_uniq, ops/depth.rs:15-39) over the rank's
resident graph image: run the HIP kernels, and -- for N > 1 -- one RCCL sum all-reduce of the
fused [depth | uniq] vector.  The timed region keeps `--in-flight` (3) calls in flight: at one GPU through the
library's pipeline (flatgfa_dev_pipeline_*: that many plans of the resident graph on as many internal streams, each
lane's pass 1 on part of the chip, so that pass 2 of one call runs beside pass 1 of another); with a collective per
step (N > 1) one plan per torch stream, the all-reduce of one step overlapping the kernels of the next
(pollen_amd/sharded.py).  Every step is a whole query into its own result buffer, all of them checked against the
oracle before anything is timed; the region ends when every kernel and every collective of its K steps has
finished.  Inputs are in HBM before the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU):
  --scaling strong (default)  BASELINE.json configs[3]: the SAME graph, its paths cut into N
                              contiguous groups of whole paths with equal step counts
                              (shard_paths / local_slice); value = the graph's steps / time.
  --scaling weak              every rank holds its own 1000 paths x 100k steps over the same
                              segments (the path set grows with the GPU count, which is when
                              sharding is warranted); value = all ranks' steps / time.
Before anything is timed the REDUCED vector is checked against the oracle on every rank.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel: its own algorithmic bytes
(k_scan reads every step once: 4N + 8P; k_accum writes the result vectors: 4Sk) over its average
duration from HIP events recorded around each launch on the launch stream, eighteen consecutive calls
that run one after the other on a plan of its own right behind the timed region (`frac`; `frac_cold`: the
same with no step kept in the Infinity Cache; `whole_call.timed_region_cold`: the region itself that way); `roofline.whole_call` is the whole call's algorithmic bytes
(4N + 8P + 4Sk) over the sum of its kernels, and -- `timed_region` -- over the region's ms_per_step.  `cpu_baseline` is the single-threaded C oracle (the reference's loop
is single-threaded) timed on this host.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (S, P, L, model)
    "cfgL": (1_000_000, 1000, 100_000, "pangenome"),
    "cfgL-uniform": (1_000_000, 1000, 100_000, "uniform"),
    "cfgL-chrom": (1_000_000, 1000, 100_000, "chromosome"),   # paths along the graph, half of them downwards, runs of 3.3
    "cfgL-short": (1_000_000, 100_000, 1000, "pangenome"),
    "cfgL-fewlong": (1_000_000, 100, 1_000_000, "pangenome"),
    "cfgL-4paths": (1_000_000, 4, 25_000_000, "pangenome"),   # fewer paths than pass 2 has waves per window
    "cfgL-medium": (1_000_000, 10_000, 10_000, "pangenome"),  # paths of ten blocks each
    "cfgL-32k": (1_000_000, 3125, 32_000, "pangenome"),       # mid-length paths
    "chrom-10k": (1_000_000, 10_000, 10_000, "chromosome"),   # ten thousand contigs of ten blocks each, half of them downwards
    "chrom-1k": (1_000_000, 100_000, 1000, "chromosome"),     # a hundred thousand short ones
    "tiny-paths": (1_000_000, 1_000_000, 100, "pangenome"),   # a million paths of a hundred steps
    "hap-1k": (1_000_000, 100_000, 1000, "haplotype"),        # a hundred thousand contigs that stay in order, half of them downwards: no segment twice, no claims
    "hap-10k": (1_000_000, 10_000, 10_000, "haplotype"),
    "hap-100": (1_000_000, 1_000_000, 100, "haplotype"),
    "cfgL-100kseg": (100_000, 1000, 100_000, "pangenome"),    # deep coverage of a small graph: 25 windows
    "cfgL-4Mseg": (4_000_000, 1000, 100_000, "pangenome"),
    "cfgL-16Mseg": (16_000_000, 1000, 100_000, "pangenome"),
    "16Mseg-tiny": (16_000_000, 10, 1000, "pangenome"),       # what a call costs when there is next to nothing to count
    "cfgL-64Mseg": (64_000_000, 1000, 100_000, "pangenome"),  # beyond the bucketed path's 16 M segments
    "cfgL-x16": (1_000_000, 16_000, 100_000, "pangenome"),    # 1.6 G steps = 6.4 GB of steps per GPU: the weak-scaling size
    "x16-16Mseg": (16_000_000, 16_000, 100_000, "pangenome"),  # the same on a graph of 16 M segments: a hundred steps per segment, as whole-genome graphs have
    "x16-16Mseg-chrom": (16_000_000, 16_000, 100_000, "chromosome"),  # ... with paths that run along the graph, as haplotypes do
    "x16-16Mseg-contigs": (16_000_000, 160_000, 10_000, "chromosome"),  # ... in contigs of ten thousand steps: more items per workgroup than a record's tag can name
    "x16-16Mseg-fewlong": (16_000_000, 16, 100_000_000, "chromosome"),  # ... in sixteen paths of a hundred million steps (each wraps around the graph six times)
    "chr-like": (16_000_000, 90, 10_000_000, "chromosome"),    # a chromosome graph as the HPRC ones are shaped: ninety haplotype paths of ten million steps each
    "chr-like-40M": (40_000_000, 90, 20_000_000, "chromosome"),  # ... on forty million segments: beyond one range of 4096-segment windows
    "chr-like-2k": (16_000_000, 2000, 500_000, "chromosome"),     # ... walked by two thousand paths of half a million steps
    "hap-chr20": (4_000_000, 2000, 100_000, "haplotype"),     # the size of one chromosome's graph (a few million segments, a couple of thousand contigs of ~100 k steps that stay in order): 200 M steps
    "rep-chr20": (4_000_000, 2000, 100_000, "repeats"),       # ... whose walks go back over 16-271 segments every 6400 steps (tandem duplications): not monotone as a whole
    "rep-16M": (16_000_000, 16_000, 100_000, "repeats"),
    "hap-16M": (16_000_000, 16_000, 100_000, "haplotype"),    # sixteen thousand haplotype walks of a hundred thousand steps that stay in their neighbourhood
    "hap-chr": (16_000_000, 90, 10_000_000, "haplotype"),      # ninety of ten million steps
    "cfgS": (10_000, 100, 10_000, "pangenome"),
    "cfgM": (100_000, 100, 100_000, "pangenome"),             # 10 M steps
}


def algorithmic_bytes(N, P, S, k):
    # each handle read once + path spans + k result vectors written once
    return 4 * N + 8 * P + 4 * S * k


def kernel_bytes(name, N, P, S, k):
    """Algorithmic bytes of one kernel of the call: the scan kernels share the step reads, pass 2
    writes the result vectors."""
    if name.startswith("k_accum"):
        return 4 * S * k
    return 4 * N + 8 * P


def git_head():
    """The commit this tree is at; on a GPU box (a snapshot without .git/) the one `make` stamped next to the library."""
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                             timeout=5).stdout.strip()
        if out:
            return out
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "pollen_amd", "lib", "HEAD")) as f:
            return f.read().strip() or None
    except OSError:
        return None


T_PROCESS0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfgL", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: shard ONE graph by path (strong), or one graph per rank (weak)")
    ap.add_argument("--rotate", type=int, default=3,
                    help="N = 1 extras: also time the loop cycling this many resident graph images (seeds 1..K), "
                         "so that no step finds its steps in the 256 MiB Infinity Cache; 0 = skip")
    ap.add_argument("--host", default="torch", choices=["torch", "c"],
                    help="who shards and reduces for N > 1: torch.distributed (one rank per GPU, RCCL through torch's nccl backend), or "
                         "the C ABI (flatgfa_sharded_*: rank 0 alone drives all N devices, RCCL inside libflatgfa.so; the other ranks only "
                         "keep the launcher's barrier)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="calls in flight: K plans of the rank's resident graph on K streams, called in turn, so that pass 2 of one "
                         "call (bound by instruction issue) shares the chip with pass 1 of the next (bound by the memory system); every "
                         "step is still a whole query into its own result buffer.  1 = strictly one call after the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cold", action="store_true",
                    help="skip roofline.frac_cold's second plan (FLATGFA_MALL_MB=0) and its calls: a kernel trace of the run then holds "
                         "the warm plan's launches alone (tools/profile_round.sh)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (depth-only, path depth, rotation, end-to-end, copy bandwidth)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import pollen_amd as pa
    from pollen_amd import device as dev
    from pollen_amd.sharded import ShardedDepth, local_slice, shard_paths

    if args.host == "c":
        return main_host_c(args, torch, dist, pa, dev)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # Test hooks (single-GPU box): FLATGFA_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # FLATGFA_BENCH_BACKEND=gloo replaces RCCL, so that the multi-rank control flow can be run there.
    if os.environ.get("FLATGFA_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("FLATGFA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_device = device if backend == "nccl" else torch.device("cpu")  # where tensors of a collective live
    scaling = args.scaling  # (N = 1: the two coincide, and the line says what the same command says at N > 1 -- "strong" by default, BASELINE.json configs[3])

    S, P, L, model = WORKLOADS[args.workload]
    N = P * L
    # ---- build the rank's graph and make it resident (not timed) ----
    strong = world > 1 and scaling == "strong"
    g = pa.synth(1 if strong else 1 + rank, S, P, L, model, False)
    steps, pb, pe, seg_len = g.soa()
    if strong:
        lo, hi = shard_paths(pb, pe, world)[rank]
        l_steps, l_pb, l_pe = local_slice(steps, pb, pe, lo, hi)
    else:
        lo, hi = 0, P
        l_steps, l_pb, l_pe = steps, pb, pe
    N_local, P_local = int((l_pe.astype(np.int64) - l_pb.astype(np.int64)).sum()), int(hi - lo)
    N_job = N if strong else world * N  # steps one step of the whole job walks
    graph = dev.DeviceGraph(l_steps, l_pb, l_pe, S, seg_len, device=str(device))
    plan = dev.DepthPlan(graph)
    in_flight = max(1, args.in_flight)

    class PipelinedDepth:
        """N = 1 with calls in flight: the library's own pipeline (flatgfa_dev_pipeline_*: K plans of the resident graph
        on K internal streams), one result buffer per lane.  Same interface as ShardedDepth."""

        def __init__(self, k):
            import ctypes
            self.pipe = dev.DepthPipeline(graph, k)
            self.bufs = [torch.zeros(2 * S, dtype=torch.int32, device=device) for _ in range(k)]
            torch.cuda.synchronize(device)  # (the zero-fills run on torch's current stream, the first writers on the pipeline's own: nothing else orders them)
            self.cur = 0
            fn = dev._lib.lib().flatgfa_dev_pipeline_seg_depth
            none = ctypes.c_void_p(-1)  # (nothing to wait for: the buffers are this loop's own, a lane's calls are in order)
            self._calls = [(lambda f=fn, p=self.pipe._p, d=ctypes.c_void_p(b[:S].data_ptr()), u=ctypes.c_void_p(b[S:].data_ptr()): f(p, d, u, none))
                           for b in self.bufs]

        @property
        def buf(self):
            return self.bufs[self.cur]

        def run(self):
            j = (self.cur + 1) % len(self.bufs)  # (call n goes to lane n mod K inside the library: warm-up and region keep in step with it)
            if self._calls[j]():
                raise SystemExit("flatgfa_dev_pipeline_seg_depth failed: " + dev._lib.last_error())
            self.cur = j

        def finish(self):
            self.pipe.join()

    plans = [plan]
    side = None
    if world == 1 and in_flight > 1:
        op = PipelinedDepth(in_flight)
        op.cur = len(op.bufs) - 1  # (the first call: buffer 0, lane 0)
    else:
        plans = [plan] + [dev.DepthPlan(graph) for _ in range(in_flight - 1)]  # (a plan belongs to one stream; they share the graph image and its claim on the Infinity Cache)
        side = [torch.cuda.Stream(device) for _ in plans] if in_flight > 1 else None
        op = ShardedDepth(S, [p.seg_depth for p in plans], device=device, with_uniq=True, streams=side)
        op.prepare(plans)   # (N = 1: every step is one call through the C ABI, resolved here)
    op1 = op if in_flight == 1 else ShardedDepth(S, plan.seg_depth, device=device, with_uniq=True)  # one call after the other, for the per-kernel samples
    if op1 is not op:
        op1.prepare([plan])

    def status_all():
        op.finish()
        if isinstance(op, PipelinedDepth):
            op.pipe.status()
            return
        for k, p in enumerate(plans):
            if side is not None:
                with torch.cuda.stream(side[k]):
                    p.status()
            else:
                p.status()

    def sync_all():
        op.finish()
        if op1 is not op:
            op1.finish()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(max(args.warmup, len(op.bufs))):  # (every result buffer written at least once: all of them are verified below)
        op.run()
    status_all()
    # What a plan makes behind its creation on a side stream (the per-block no-claim marks: three more reads of the steps, started by
    # its first call) is part of warming up, not of the steady state the region measures: describe() waits for it.
    for p_ in plans:
        p_.describe()
    if isinstance(op, PipelinedDepth):
        op.pipe.describe()
    sync_all()

    # ---- verification of what is being timed: the REDUCED vector against the oracle, on every rank ----
    # Before the region (what the warm-up wrote) AND behind it: every result buffer the timed calls wrote, the buffers of
    # the one-call-at-a-time loop and of the sampled pass, and those of the cold plan and the cold region are compared
    # with the oracle's vector again once they have been written for the last time (`verified_after`); a line is only
    # printed when all of them agree.
    verified = None
    want = None
    checked = []  # (what was compared with the oracle, and when: the line's `verified_after`)
    if not args.no_verify:
        from oracle import flatgfa_oracle as fo
        pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
        want_d, want_u = fo.seg_depth_with_uniq(pools)
        want = np.concatenate([want_d, want_u]).astype(np.int64)
        del pools
        if world > 1 and not strong:
            # every rank has its own graph: the expected reduced vector is the sum of the ranks' oracle vectors
            t = torch.from_numpy(want).to(coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            want = t.cpu().numpy()

    def check_bufs(bufs, what):
        """Every buffer of `bufs` ([depth | uniq], u32) against the oracle's vector, on every rank; exits when one differs."""
        if want is None:
            return None
        ok, first_bad = True, None
        for k, b in enumerate(bufs):
            got = b.cpu().numpy().view(np.uint32).astype(np.int64)
            if not bool((got == want).all()):
                ok = False
                if first_bad is None:
                    bad = np.nonzero(got != want)[0]
                    first_bad = (k, len(bad), bad[:4].tolist(), got[bad[:4]].tolist(), want[bad[:4]].tolist())
        if world > 1:
            t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            all_ok = bool(int(t.item()))
        else:
            all_ok = ok
        if not all_ok:
            detail = "" if first_bad is None else (f": buffer {first_bad[0]} differs in {first_bad[1]} of {len(want)} entries (first at {first_bad[2]}: "
                                                   f"got {first_bad[3]}, want {first_bad[4]})")
            raise SystemExit(f"rank {rank}: HIP result differs from the oracle {what}{detail}: refusing to report a number")
        checked.append({"what": what, "buffers": len(bufs)})
        return True

    verified = check_bufs(op.bufs, "behind the warm-up, before the timed region")  # (calls in flight: every lane's result, not only the last one's)

    # ---- timed region: exactly K steps ----
    # Kernel durations come from HIP events recorded around each launch on the launch stream.  A step
    # that carries its four event records takes 0.20 ms where the others take 0.15 (the records are
    # queue packets of their own, each with a completion signal), and `value` is the whole region's
    # throughput, so inside the region only every EVENT_EVERY-th step is sampled -- never step 0, the
    # first launch behind a full sync -- and SAMPLE_STEPS more steps, every one of them sampled, follow
    # right behind the region on the same stream (two unsampled steps first).  The roofline uses all of
    # these warm samples; both groups are also reported on their own.
    EVENT_EVERY, EVENT_AT, SAMPLE_STEPS = 10, 5, 16

    def timed_loop(run_step, n_steps, with_events):
        dev.profile_enable(False)
        dev.profile_read()
        sync_all()
        t0 = time.perf_counter()
        for i in range(n_steps):
            if with_events:
                dev.profile_enable(i % EVENT_EVERY == EVENT_AT)
            run_step(i)
        sync_all()
        t1 = time.perf_counter()
        dev.profile_enable(False)
        return t1 - t0, dev.profile_read()

    # Calls in flight (K > 1): the region is the pipelined loop and carries no event records -- a kernel's
    # duration next to another call's kernel on the same chip is not that kernel's own -- and every per-kernel
    # sample comes from the pass behind it, which runs one call after the other on the first plan alone.
    elapsed, kernels_region = timed_loop(lambda i: op.run(), args.steps, in_flight == 1)
    status_all()
    check_bufs(op.bufs, f"behind the timed region ({args.steps} steps, {in_flight} in flight)")  # (what the region's calls wrote, not the warm-up's)
    n_region_samples = len([i for i in range(args.steps) if i % EVENT_EVERY == EVENT_AT]) if in_flight == 1 else 0
    if in_flight > 1:
        SAMPLE_STEPS = 18
    serial_elapsed = None
    if in_flight > 1:  # the same loop with one call in flight (no event records either)
        for _ in range(2):
            op1.run()
        serial_elapsed, _ = timed_loop(lambda i: op1.run(), args.steps, False)
        plan.status()
        check_bufs(op1.bufs, "behind the one-call-at-a-time loop")
    # the sampled pass behind the region
    for _ in range(2):
        op1.run()
    dev.profile_enable(True)
    for _ in range(SAMPLE_STEPS):
        op1.run()
    sync_all()
    dev.profile_enable(False)
    kernels_post = dev.profile_read()
    kernels = kernels_region + kernels_post
    n_timed_steps = n_region_samples + SAMPLE_STEPS
    status_all()
    plan.status()
    check_bufs(op1.bufs, "behind the sampled pass (the calls whose kernels `roofline` prices)")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1: the all-reduce alone (8 MB, latency-bound), for reading the scaling numbers
    allreduce_ms = None
    if world > 1:
        sync_all()
        c0 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(op.buf, op=dist.ReduceOp.SUM)
        sync_all()
        allreduce_ms = (time.perf_counter() - c0) / 10 * 1e3
        # the buffer now holds garbage sums: one more real step leaves a valid result behind
        op.run()
        sync_all()

    # ---- per-kernel durations -> roofline for the dominant kernel ----
    per = {}
    for name, ms in kernels:
        per.setdefault(name, []).append(ms)
    kern_avg_ms = {k: float(np.mean(v)) for k, v in per.items()}
    dom = max(kern_avg_ms, key=lambda k: kern_avg_ms[k] * len(per[k])) if kern_avg_ms else None
    B_call = algorithmic_bytes(N_local, P_local, S, 2)
    device_ms_per_step = sum(kern_avg_ms[k] * len(per[k]) for k in per) / max(n_timed_steps, 1)
    roofline = None
    if dom:
        B_dom = kernel_bytes(dom, N_local, P_local, S, 2)
        # A kernel that is launched several times per call on different paths (k_scan_short: once for the paths read
        # forwards, once for the reversed copies) reads its share of the steps per launch; a plan with segment ranges
        # walks ALL steps once per range.  `achieved` is per launch: the launch's bytes over the launch's time.
        launches = len(per[dom]) / max(n_timed_steps, 1)
        m_ranges = re.search(r"ranges=(\d+)", plan.describe())
        shares = max(1.0, launches / max(1, int(m_ranges.group(1)) if m_ranges else 1))
        # ... and where several scan kernels share the paths out, each reads its own paths' steps (the plan says how many)
        m_steps = re.search(r" steps=(\d+)/(\d+)/(\d+)/(\d+)", plan.describe())
        cls = {"k_scan_short": 1, "k_scan_medium": 2, "k_scan_tiny": 3, "k_scan_dense": 0, "k_scan": 0}.get(dom.split("<")[0])
        if m_steps and cls is not None and not dom.startswith("k_accum"):
            B_dom = 4 * int(m_steps.group(1 + cls)) + 8 * P_local
        if dom.startswith("k_accum"):
            shares = max(1.0, launches)  # (pass 2: a launch per range of segments, each writing its range's counts)
        B_dom = int(B_dom / shares)
        achieved = B_dom / (kern_avg_ms[dom] * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel from the PMC counters: they need separate
        # rocprofv3 --pmc passes (tools/profile_round.sh), so the committed summary of the same
        # workload is quoted, with where it came from; null when it is for another workload.
        traffic, traffic_source = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
            if tj.get("workload") == args.workload and world == 1 and dom in tj["kernels"]:
                traffic = tj["kernels"][dom]["hbm_bytes"]
                traffic_source = {"file": "profiles/latest_traffic.json", "from": tj.get("source"), "commit": tj.get("commit")}
        except (OSError, ValueError, KeyError):
            pass
        whole = B_call / (device_ms_per_step * 1e-3) / 1e9 if device_ms_per_step else None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_source,
                    "kernel_avg_ms": round(kern_avg_ms[dom], 5), "algorithmic_bytes": B_dom,
                    "whole_call": {"algorithmic_bytes": B_call, "all_kernels_ms_per_step": round(device_ms_per_step, 5),
                                   "achieved": round(whole, 2) if whole else None,
                                   "frac": round(whole / HBM_PEAK_GBS, 5) if whole else None},
                    "kernels_avg_ms": {k: round(v, 5) for k, v in kern_avg_ms.items()},
                    "kernel_launches_per_step": {k: round(len(v) / max(n_timed_steps, 1), 2) for k, v in per.items()},
                    "plan_choice": plan.describe(),
                    "kernel_timing": f"HIP events around each launch: steps {EVENT_AT}, {EVENT_AT + EVENT_EVERY}, ... of the timed region "
                                     f"({n_region_samples} of {args.steps} steps; never step 0) plus {SAMPLE_STEPS} consecutive steps right "
                                     "behind it on the same stream; `achieved` uses the mean over all of them"}
        v = np.array(per[dom], dtype=np.float64)

        def stats(x):
            x = np.array(x, dtype=np.float64)
            return {"n": int(len(x)), "mean_ms": round(float(x.mean()), 5), "median_ms": round(float(np.median(x)), 5),
                    "min_ms": round(float(x.min()), 5), "max_ms": round(float(x.max()), 5)} if len(x) else {"n": 0}
        roofline["samples"] = dict(stats(v), in_region=stats([ms for n_, ms in kernels_region if n_ == dom]),
                                   behind_region=stats([ms for n_, ms in kernels_post if n_ == dom]),
                                   frac_at_median=round(B_dom / (float(np.median(v)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))
        roofline["n"] = int(len(v))
        # the whole call as the timed region ran it (K calls in flight): SURVEY.md 8(d)'s bytes over ms_per_step
        region_ms = elapsed / args.steps * 1e3
        roofline["whole_call"]["timed_region"] = {
            "calls_in_flight": in_flight, "ms_per_step": round(region_ms, 5),
            "achieved": round(B_call / (region_ms * 1e-3) / 1e9, 2), "frac": round(B_call / (region_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "ms_per_step_one_call_in_flight": round(serial_elapsed / args.steps * 1e3, 5) if serial_elapsed else round(region_ms, 5)}
        if in_flight > 1:
            roofline["kernel_timing"] = (f"HIP events around each launch of {SAMPLE_STEPS} consecutive steps right behind the timed region, one call "
                                         "after the other on a plan of its own (flatgfa_dev_plan_create: one pass-1 workgroup per CU); the region itself keeps "
                                         f"{in_flight} calls in flight through flatgfa_dev_pipeline_* -- whose lanes run pass 1 on fewer workgroups, "
                                         "config.pipeline: scan_workgroups -- and carries no event records; `achieved` uses the mean over all of them")
        # Every step from HBM: the plan keeps the first cache_resident_mb of the steps in the 256 MiB Infinity Cache from
        # call to call (they are read without the nt hint), which a loop over ONE resident graph profits from.  The same
        # graph through a plan made with FLATGFA_MALL_MB=0, sampled the same way, is what a call costs when nothing of
        # its steps is left in that cache (more graphs queried in turn than it holds: extras.rotate).
        m_res = re.search(r"cache_resident_mb=(\d+)", plan.describe())
        if world == 1 and dom.startswith("k_scan") and m_res and int(m_res.group(1)) > 0 and not args.no_cold:
            os.environ["FLATGFA_MALL_MB"] = "0"
            try:
                plan0 = dev.DepthPlan(graph)
            finally:
                del os.environ["FLATGFA_MALL_MB"]
            cold = torch.zeros(2 * S, dtype=torch.int32, device=device)
            for _ in range(3):
                plan0.seg_depth(cold[:S], cold[S:])
            plan0.status()
            plan0.describe()  # (waits for what the plan makes behind its creation: not part of the calls sampled below)
            torch.cuda.synchronize(device)
            dev.profile_enable(True)
            dev.profile_read()
            for _ in range(SAMPLE_STEPS):
                plan0.seg_depth(cold[:S], cold[S:])
            plan0.status()
            dev.profile_enable(False)
            per0 = {}
            for name, ms in dev.profile_read():
                per0.setdefault(name, []).append(ms)
            if dom in per0:
                cold_ms = float(np.mean(per0[dom]))
                roofline["frac_cold"] = round(B_dom / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                roofline["cold"] = {
                    "what": f"the same graph through a plan made with FLATGFA_MALL_MB=0 (no step kept in the Infinity Cache: `frac` has "
                            f"{m_res.group(1)} MB of them resident there from call to call), {len(per0[dom])} consecutive calls, HIP events",
                    "kernel_avg_ms": round(cold_ms, 5), "achieved": round(B_dom / (cold_ms * 1e-3) / 1e9, 2), "n": len(per0[dom]),
                    "kernels_avg_ms": {k: round(float(np.mean(v2)), 5) for k, v2 in per0.items()},
                    "seconds_into_process": round(time.perf_counter() - T_PROCESS0, 1),
                    "same_result": bool((cold.cpu().numpy() == op1.buf.cpu().numpy()).all()) if world == 1 else None}
            check_bufs([cold], "behind the cold plan's sampled calls (FLATGFA_MALL_MB=0)")
            plan0.close()
            del plan0, cold
            # ... and the timed region the same way: a pipeline whose lanes keep nothing of the steps in the Infinity Cache
            if in_flight > 1 and isinstance(op, PipelinedDepth):
                os.environ["FLATGFA_MALL_MB"] = "0"
                try:
                    op0 = PipelinedDepth(in_flight)
                finally:
                    del os.environ["FLATGFA_MALL_MB"]
                op0.cur = len(op0.bufs) - 1
                for _ in range(max(args.warmup, in_flight)):
                    op0.run()
                op0.finish()
                op0.pipe.describe()  # (every lane's background work done before the region)
                torch.cuda.synchronize()
                t0c = time.perf_counter()
                for _ in range(args.steps):
                    op0.run()
                op0.finish()
                torch.cuda.synchronize()
                cold_region_ms = (time.perf_counter() - t0c) / args.steps * 1e3
                op0.pipe.status()
                check_bufs(op0.bufs, f"behind the cold timed region ({args.steps} steps, {in_flight} in flight, FLATGFA_MALL_MB=0)")
                same = all(bool((b.cpu().numpy() == op1.buf.cpu().numpy()).all()) for b in op0.bufs)
                roofline["whole_call"]["timed_region_cold"] = {
                    "what": "the timed region again through a pipeline made with FLATGFA_MALL_MB=0: every step of every call from HBM",
                    "calls_in_flight": in_flight, "ms_per_step": round(cold_region_ms, 5),
                    "achieved": round(B_call / (cold_region_ms * 1e-3) / 1e9, 2), "frac": round(B_call / (cold_region_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "same_result": same}
                del op0
        # An event pair also contains the launch itself; what it reads around a kernel of k_scan's
        # shape that does nothing is reported beside the kernel times, not subtracted from them
        # (rocprofv3's dispatch durations, profiles/, do not contain it).
        ov = dev.profile_overhead_ms(256, 142 * 1024, 20)
        if ov >= 0:
            roofline["event_pair_around_an_empty_launch_ms"] = round(ov, 5)

    # ---- secondary measurements, rank 0 at N=1 only, outside the timed region ----
    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize(device)
            ts = []
            for _ in range(reps):
                c0 = time.perf_counter()
                fn()
                torch.cuda.synchronize(device)
                ts.append(time.perf_counter() - c0)
            return float(np.median(ts)) * 1e3

        # The first answer: the reference's consumers ask one depth query per graph (cli/cmds.rs:234-285; bench/config.toml:29-32
        # times a process per query), so what a resident image costs until its first verified result is a figure of its own:
        # flatgfa_dev_plan_create_first -- the query that sizes the plan's scratch writes the caller's buffers.
        if want is not None:
            fa_runs = []
            fa_d = torch.empty(S, dtype=torch.int32, device=device)
            fa_u = torch.empty(S, dtype=torch.int32, device=device)
            for _ in range(4):
                fa_d.fill_(-1)
                fa_u.fill_(-1)
                torch.cuda.synchronize(device)
                c0 = time.perf_counter()
                fplan = dev.DepthPlan(graph, first=(fa_d, fa_u))   # (the buffers are complete when this returns: no synchronize here)
                fa_runs.append((time.perf_counter() - c0) * 1e3)
                got = torch.cat([fa_d, fa_u]).cpu().numpy().view(np.uint32).astype(np.int64)
                if fplan.first_status != 0 or not bool((got == want).all()):
                    raise SystemExit("the first answer (flatgfa_dev_plan_create_first) differs from the oracle: refusing to report a number")
                fplan.close()
            fa_ms = float(np.median(fa_runs[1:]))
            extras["first_answer"] = {
                "what": "resident graph image -> seg_depth_with_uniq in the caller's device buffers through flatgfa_dev_plan_create_first (the plan's "
                        "sizing query IS the first query; the per-block no-claim marks are made behind it on a side stream), host wall clock, "
                        "every run checked against the oracle; median of the last three of four fresh plans",
                "ms": round(fa_ms, 4), "runs_ms": [round(x, 4) for x in fa_runs], "algorithmic_bytes": B_call,
                "achieved": round(B_call / (fa_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(B_call / (fa_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "steps_per_s": round(N / (fa_ms * 1e-3), 1),
                "bit_exact_vs_oracle": True,
                "steady_state_queries_it_costs": round(fa_ms / (serial_elapsed / args.steps * 1e3), 1) if serial_elapsed else None}
            del fa_d, fa_u
        # a2: depth only; a3: path depth of all paths (seg_depth + the per-path sums)
        d_only = torch.zeros(S, dtype=torch.int32, device=device)
        extras["seg_depth_only_ms"] = round(timed(lambda: plan.seg_depth(d_only, None)), 5)
        ids = torch.arange(P, dtype=torch.int32, device=device)
        len_out = torch.zeros(P, dtype=torch.int64, device=device)
        wsum_out = torch.zeros(P, dtype=torch.int64, device=device)

        def path_depth_two_walks():  # round 1: node depth, then a second walk of the steps with one gather each
            plan.seg_depth(d_only, None)
            plan.path_sums(ids, d_only, len_out, wsum_out)
        extras["path_depth_all_paths_ms"] = round(timed(lambda: plan.path_depth_all(d_only, len_out, wsum_out)), 5)
        extras["path_depth_all_paths_two_walks_ms"] = round(timed(path_depth_two_walks), 5)
        plan.status()

        def sampled(fn, calls=SAMPLE_STEPS):  # per-kernel HIP-event times of `calls` consecutive calls, one after the other
            fn()
            plan.status()
            dev.profile_enable(True)
            dev.profile_read()
            for _ in range(calls):
                fn()
            plan.status()
            dev.profile_enable(False)
            per_k = {}
            for name, ms in dev.profile_read():
                per_k.setdefault(name, []).append(ms)
            return {k: float(np.sum(v)) / calls for k, v in per_k.items()}, {k: len(v) / calls for k, v in per_k.items()}

        def hbm_row(bytes_, ms):
            return {"bound": "hbm", "algorithmic_bytes": int(bytes_), "ms": round(ms, 5), "achieved": round(bytes_ / (ms * 1e-3) / 1e9, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}

        # ---- a3: path_depth of all paths (depth.rs:88-131; what bench/config.toml:30 times) as a measurement row of its own ----
        # algorithmic bytes: every handle once, the spans, seg_len read and node depth written once, two u64 sums per path
        pd_per_call, pd_launches = sampled(lambda: plan.path_depth_all(d_only, len_out, wsum_out))
        pd_bytes = 4 * N + 8 * P + 8 * S + 16 * P
        pd_ms = sum(pd_per_call.values())
        pd_dom = max(pd_per_call, key=pd_per_call.get) if pd_per_call else None
        pd_row = {"what": f"path_depth of all {P} paths (ops/depth.rs:88-131): node depth and the two sums of measure_path per path in ONE walk of the "
                          "steps (flatgfa_dev_path_depth_all); the one f64 division per path is the host's",
                  "ms_per_call": extras["path_depth_all_paths_ms"], "steps_per_s": round(N / (extras["path_depth_all_paths_ms"] * 1e-3), 1),
                  "kernels_ms_per_call": {k: round(v, 5) for k, v in pd_per_call.items()}, "kernel_launches_per_call": pd_launches,
                  "roofline": dict(hbm_row(pd_bytes, pd_ms), bytes_are="4 N + 8 P + 8 S + 16 P (steps, spans, seg_len in and depth out, the sums)",
                                   ms_is="all kernels of a call, HIP events, one call after the other"),
                  "dominant_kernel": None if not pd_dom else dict(hbm_row(4 * N + 8 * P, pd_per_call[pd_dom]), kernel=pd_dom, bytes_are="its step reads, 4 N + 8 P")}
        if not args.no_cpu_baseline:
            from oracle import flatgfa_oracle as fo
            pools_a3 = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
            fo.path_depth(pools_a3)
            ts = []
            for _ in range(3):
                c0 = time.perf_counter()
                want_len, want_mean = fo.path_depth(pools_a3)
                ts.append(time.perf_counter() - c0)
            got_len, got_mean = g.path_depth()
            pd_row["cpu_baseline"] = {"value": round(N / float(np.median(ts)), 1), "unit": "path-steps/s", "cores": 1, "kind": "port",
                                      "sample": f"oracle_path_depth (depth_oracle.c: seg_depth, then measure_path per path) on the full {args.workload} "
                                                f"graph, median of {len(ts)} runs after 1 warm-up", "seconds_median": round(float(np.median(ts)), 4)}
            pd_row["bit_exact_vs_oracle"] = bool((got_len == want_len).all() and got_mean.tobytes() == want_mean.tobytes())
            del pools_a3
        if in_flight > 1:  # the same query with calls in flight (flatgfa_dev_pipeline_path_depth_all), every lane its own outputs
            ppipe = dev.DepthPipeline(graph, in_flight)
            outs = [(torch.zeros(S, dtype=torch.int32, device=device), torch.zeros(P, dtype=torch.int64, device=device),
                     torch.zeros(P, dtype=torch.int64, device=device)) for _ in range(in_flight)]
            torch.cuda.synchronize(device)  # (zero-fills on the current stream before first writers on the pipeline's own)
            for k in range(2 * in_flight):
                ppipe.path_depth_all(*outs[k % in_flight], after_current_stream=False)
            ppipe.status()
            torch.cuda.synchronize(device)
            c0 = time.perf_counter()
            for k in range(args.steps):
                ppipe.path_depth_all(*outs[k % in_flight], after_current_stream=False)
            ppipe.status()
            pip_ms = (time.perf_counter() - c0) / args.steps * 1e3
            same = all(bool((o[1].cpu().numpy() == len_out.cpu().numpy()).all() and (o[2].cpu().numpy() == wsum_out.cpu().numpy()).all()) for o in outs)
            pd_row["calls_in_flight"] = {"n": in_flight, "ms_per_call": round(pip_ms, 5), "steps_per_s": round(N / (pip_ms * 1e-3), 1),
                                         "frac": round(pd_bytes / (pip_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "same_sums_as_one_at_a_time": same}
            ppipe.close()
            del ppipe, outs
        extras["path_depth"] = pd_row
        # The benchmark's walk continues 90 % of its steps (runs of 10).  The same shape with paths
        # that run along the graph, every other one downwards, 70 % continuing (0.3 records per
        # step): what a chromosome graph looks like to the kernels.
        if args.workload == "cfgL":
            gc = pa.synth(1, S, P, L, "chromosome", False)
            cs, cb, ce, cl = gc.soa()
            cplan = dev.DepthPlan(dev.DeviceGraph(cs, cb, ce, S, cl, device=str(device)))
            cd = torch.zeros(S, dtype=torch.int32, device=device)
            cu = torch.zeros(S, dtype=torch.int32, device=device)
            cplan.seg_depth(cd, cu)
            cplan.status()
            cplan.describe()
            torch.cuda.synchronize(device)
            chrom_ms = timed(lambda: cplan.seg_depth(cd, cu))
            cplan.status()
            want_c = fo.seg_depth_with_uniq(fo.Pools(**{n: gc.pool(n) for n in fo.POOL_ORDER})) if not args.no_verify else None
            extras["chromosome_model"] = {
                "what": f"seg_depth_with_uniq on synth(seed=1, S={S}, P={P}, L={L}, model=chromosome)", "ms_per_call": round(chrom_ms, 5),
                "steps_per_s": round(N / (chrom_ms * 1e-3), 1),
                "bit_exact_vs_oracle": None if want_c is None else bool(
                    (cd.cpu().numpy().view(np.uint32) == want_c[0]).all() and (cu.cpu().numpy().view(np.uint32) == want_c[1]).all())}
            cplan.close()
            del gc, cs, cplan, cd, cu
        # The timed loop walks the same 400 MB image every step; MI355X has 256 MiB of Infinity
        # Cache and FETCH_SIZE counts its hits as fetches.  Cycle K resident images (> 1 GB): if
        # the cache helped, this is slower.
        if args.rotate >= 2:
            ops = [op1]  # (one call in flight in both loops)
            keep = []
            for seed in range(2, args.rotate + 1):
                gk = pa.synth(seed, S, P, L, model, False)
                sk, bk, ek, lk = gk.soa()
                grk = dev.DeviceGraph(sk, bk, ek, S, lk, device=str(device))
                plk = dev.DepthPlan(grk)
                ops.append(ShardedDepth(S, plk.seg_depth, device=device, with_uniq=True))
                keep.append((gk, grk, plk))
            for o in ops:
                o.run()
            for _, _, plk in keep:
                plk.describe()
            torch.cuda.synchronize(device)
            rot_elapsed, _ = timed_loop(lambda i: ops[i % len(ops)].run(), args.steps, False)
            same_elapsed, _ = timed_loop(lambda i: op1.run(), args.steps, False)
            for _, _, plk in keep:
                plk.status()
            extras["rotate"] = {"images": len(ops), "resident_step_bytes": len(ops) * 4 * N,
                                "ms_per_step_rotating": round(rot_elapsed / args.steps * 1e3, 5),
                                "ms_per_step_same_image": round(same_elapsed / args.steps * 1e3, 5),
                                "note": "no per-kernel events in either loop"}
            del ops, keep
        # device-to-device copy bandwidth of this box (read + write bytes), for the roofline's second denominator
        a = torch.empty(1 << 28, dtype=torch.int32, device=device)
        b = torch.empty_like(a)
        copy_ms = timed(lambda: b.copy_(a), reps=10)
        extras["d2d_copy_gbs"] = round(2 * a.numel() * 4 / (copy_ms * 1e-3) / 1e9, 1)
        del a, b
        if roofline:
            roofline["frac_of_measured_copy"] = round(roofline["achieved"] / extras["d2d_copy_gbs"], 5)
            if roofline.get("traffic"):
                # the bytes the kernel really moves (PMC: step reads + record writes, partial lines included) over its
                # duration, against what a device-to-device copy achieves on THIS box: how much of the memory system's
                # practical rate the kernel uses -- what is left is a matter of bytes, not of the kernel's schedule
                moved = roofline["traffic"] / (roofline["kernel_avg_ms"] * 1e-3) / 1e9
                roofline["traffic_gbs"] = round(moved, 1)
                roofline["traffic_frac_of_measured_copy"] = round(moved / extras["d2d_copy_gbs"], 5)
        # end to end through the host API on a fresh handle: .flatgfa mmap -> H2D -> kernels -> D2H -> TSV text
        import tempfile
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        fpath = os.path.join(tmpdir, f"bench_{os.getpid()}.flatgfa")
        try:
            g.write_flatgfa(fpath)
            runs = []
            for rep in range(2):  # the first run also pays for this process's first pinned staging buffers and worker threads
                if rep:
                    g2.close()
                c0 = time.perf_counter()
                g2 = pa.load(fpath)
                c1 = time.perf_counter()
                g2.to_device(local_rank)
                c2 = time.perf_counter()
                h2d_ms, plan_ms = g2.residency_ms()
                g2.seg_depth_with_uniq()
                c3 = time.perf_counter()
                text = g2.depth_table()
                c4 = time.perf_counter()
                runs.append({"load_ms": round((c1 - c0) * 1e3, 3), "h2d_and_plan_ms": round((c2 - c1) * 1e3, 3),
                             "h2d_ms": round(h2d_ms, 3), "plan_ms": round(plan_ms, 3),
                             "first_query_ms": round((c3 - c2) * 1e3, 3), "table_ms": round((c4 - c3) * 1e3, 3),
                             "total_ms": round((c4 - c0) * 1e3, 3), "steps_per_s": round(N / (c4 - c0), 1)})
            extras["end_to_end"] = dict(runs[1], what=".flatgfa mmap -> H2D -> seg_depth_with_uniq (kernels + D2H + widen to u64) -> "
                                        "depth table text; a fresh handle each time, second of two runs in this process",
                                        table_bytes=len(text), first_run=runs[0])
            # the same on the host alone: oracle compute + oracle emitter (one core, as the reference runs)
            from oracle import flatgfa_oracle as fo
            pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
            fo.fgfa_depth(pools, True)
            c5 = time.perf_counter()
            cpu_text = fo.fgfa_depth(pools, True)
            c6 = time.perf_counter()
            extras["cpu_end_to_end"] = {"what": "oracle seg_depth_with_uniq + oracle SegDepth::emit, one core, arrays in memory",
                                        "total_ms": round((c6 - c5) * 1e3, 3), "same_bytes": bool(cpu_text == text)}
            # The reference's own measurement (bench/config.toml:29-32, bench/bench.py:68-85): the whole
            # PROCESS `fgfa -i G.flatgfa depth` under hyperfine --warmup=1 --min-runs=3, output discarded --
            # exec to exit, so HIP start-up, code-object load and the first upload are all inside.  Beside it
            # the same command with -d, and the oracle's process for both (one core; mmap -> compute -> emit).
            def processes(cmd, runs=5):
                ts = []
                for k in range(runs + 1):  # the first run is the warm-up
                    c0 = time.perf_counter()
                    rc = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
                    c1 = time.perf_counter()
                    if rc != 0:
                        return {"error": f"exit code {rc}", "command": " ".join(cmd)}
                    if k:
                        ts.append((c1 - c0) * 1e3)
                return {"mean_ms": round(float(np.mean(ts)), 2), "stddev_ms": round(float(np.std(ts)), 2), "min_ms": round(min(ts), 2),
                        "max_ms": round(max(ts), 2), "runs": len(ts), "warmup_runs": 1}
            fgfa = os.path.join(ROOT, "pollen_amd", "bin", "fgfa")
            cpu_exe = fo.cpu_cli()
            extras["cli_process"] = {
                "what": "whole-process wall time, exec to exit, stdout discarded, 1 warm-up + 5 runs each (the reference times "
                        "`fgfa -i G.flatgfa depth` this way: bench/config.toml:29-32); graph file in " + tmpdir,
                "graph_file_bytes": os.path.getsize(fpath),
                "fgfa_depth": processes([fgfa, "-i", fpath, "depth"]),
                "fgfa_depth_d": processes([fgfa, "-i", fpath, "depth", "-d"]),
                "cpu_oracle_depth": processes([cpu_exe, fpath]),
                "cpu_oracle_depth_d": processes([cpu_exe, fpath, "-d"]),
                "cpu": "oracle/fgfa_depth_cpu.c: one core, mmap -> depth.rs loops -> emit into one buffer -> one write",
            }
            # BASELINE.json configs[4]: which paths share an oriented handle with which (all pairs)
            g2.path_overlaps([0])  # builds the per-path handle bitsets (once per resident graph)
            c5 = time.perf_counter()
            touch = g2.path_overlaps(list(range(P)))
            c6 = time.perf_counter()
            extras["overlap_all_pairs"] = {"pairs": int(P) * int(P), "touching": int(touch.sum()),
                                           "ms": round((c6 - c5) * 1e3, 3), "note": "host API call incl. D2H of the P x P byte matrix"}
            g2.close()
            # ... as a measurement row: on the device (inputs and the P x P result resident), a fresh plan's first call
            # (it builds the per-path bitmaps: k_coarse_bits, k_handle_bits) and the calls after it (k_pair_touch alone)
            if P * P <= (1 << 26):
                oplan = dev.DepthPlan(graph)
                q_all = torch.arange(P, dtype=torch.int32, device=device)
                t_all = torch.zeros(P * P, dtype=torch.uint8, device=device)
                torch.cuda.synchronize(device)
                dev.profile_enable(True)
                dev.profile_read()
                c5 = time.perf_counter()
                oplan.path_overlaps(q_all, t_all)
                oplan.status()
                first_ms = (time.perf_counter() - c5) * 1e3
                dev.profile_enable(False)
                build = {n_: ms for n_, ms in dev.profile_read()}
                c5 = time.perf_counter()
                for _ in range(5):
                    oplan.path_overlaps(q_all, t_all)
                oplan.status()
                steady_ms = (time.perf_counter() - c5) / 5 * 1e3
                dev.profile_enable(True)
                dev.profile_read()
                for _ in range(5):
                    oplan.path_overlaps(q_all, t_all)
                oplan.status()
                dev.profile_enable(False)
                pair_ms = float(np.mean([ms for n_, ms in dev.profile_read() if n_ == "k_pair_touch"] or [0.0]))
                words = (((S + 31) // 32) + 3) & ~3
                cwords = ((2 * S + 2047) // 2048 + 31) // 32
                bitset_bytes = P * 2 * words * 4
                # a pair reads both coarse bitmaps and, where they share a block, at least one block of both exact bitsets (one
                # word per lane) before it can stop; one result byte per pair
                pair_bytes = P * P * (2 * cwords * 4 + 1) + int(touch.sum()) * 2 * 256
                orow = {"what": f"all {P} x {P} path pairs on {args.workload} (slow_odgi/overlap.py:6-32; BASELINE.json configs[4]); inputs and the P x P byte "
                                "matrix resident in HBM", "pairs": int(P) * int(P), "touching": int(touch.sum()),
                        "first_call_ms": round(first_ms, 4), "ms_per_call": round(steady_ms, 5), "pairs_per_s": round(P * P / (steady_ms * 1e-3), 1),
                        "kernels_first_call_ms": {k: round(v, 5) for k, v in build.items()},
                        "roofline": {
                            "k_pair_touch": dict(hbm_row(pair_bytes, pair_ms) if pair_ms else {}, bytes_are="per pair both coarse bitmaps and one result byte; per "
                                                 "touching pair one 2048-handle block of both exact bitsets (the least it must see before it stops)"),
                            "k_coarse_bits": dict(hbm_row(4 * N + 8 * P + P * cwords * 4, build["k_coarse_bits"]) if build.get("k_coarse_bits") else {},
                                                  bytes_are="4 N + 8 P + the bitmaps out (once per plan)"),
                            "k_handle_bits": dict(hbm_row(4 * N + 8 * P + bitset_bytes, build["k_handle_bits"]) if build.get("k_handle_bits") else {},
                                                  bytes_are="4 N + 8 P + the exact bitsets out (once per plan; the kernel reads the steps once per orientation: "
                                                            "a window of one orientation's bits is what fits the LDS)")}}
                t_dev = t_all.cpu().numpy().reshape(P, P)
                orow["same_as_host_api"] = bool((t_dev == touch.reshape(P, P)).all())
                if not args.no_cpu_baseline:
                    pools_ov = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
                    c5 = time.perf_counter()
                    want_t = fo.path_touches(pools_ov, np.arange(P, dtype=np.uint32))
                    cpu_s = time.perf_counter() - c5
                    orow["cpu_baseline"] = {"value": round(P * P / cpu_s, 1), "unit": "path-pairs/s", "cores": 1, "kind": "port",
                                            "sample": f"oracle_path_touches (overlap_oracle.c: a handle bitset per path, then word-wise ANDs with early exit) on the "
                                                      f"full {args.workload} graph, all {P * P} pairs, one run (bitset build included)", "seconds": round(cpu_s, 3)}
                    orow["bit_exact_vs_oracle"] = bool((want_t == t_dev).all())
                    del pools_ov
                extras["overlap"] = orow
                oplan.close()
                del oplan, q_all, t_all
            # The same query where few pairs touch: every path is folded into its own band of the
            # segments (two band widths wide, so that neighbours can share handles).  All pairs on
            # cfg-L touch after a handful of probes; here nearly every pair is settled by the
            # coarse bitmaps and the rest walk a whole path against a query's exact bitset.
            if model == "pangenome" and P >= 4 and S >= 4 * P and N == P * L:
                band = S // P
                ids = (steps >> 1).reshape(P, L)
                folded = (np.arange(P, dtype=np.uint32)[:, None] * np.uint32(band) + ids % np.uint32(2 * band)) % np.uint32(S)
                bsteps = ((folded << 1) | (steps.reshape(P, L) & 1)).reshape(-1).astype(np.uint32)
                bgraph = dev.DeviceGraph(bsteps, pb, pe, S, seg_len, device=str(device))
                bplan = dev.DepthPlan(bgraph)
                q = torch.arange(P, dtype=torch.int32, device=device)
                t_out = torch.zeros(P * P, dtype=torch.uint8, device=device)
                bplan.path_overlaps(q, t_out)   # builds the coarse bitmaps (once per plan)
                bplan.status()
                dev.profile_enable(True)
                dev.profile_read()
                c5 = time.perf_counter()
                bplan.path_overlaps(q, t_out)
                bplan.status()
                c6 = time.perf_counter()
                dev.profile_enable(False)
                kern = {n: round(ms, 5) for n, ms in dev.profile_read()}
                t_np = t_out.cpu().numpy().reshape(P, P)
                cwords = (2 * S + 2047) // 2048 // 32 + 1
                extras["overlap_banded"] = {
                    "what": "all pairs; path p folded into segments [p*S/P, (p+2)*S/P)", "pairs": int(P) * int(P),
                    "touching": int(t_np.sum()), "symmetric": bool((t_np == t_np.T).all()), "ms": round((c6 - c5) * 1e3, 3),
                    "kernels_ms": kern,
                    "algorithmic_bytes": {"coarse_bitmap_reads": 2 * P * P * cwords * 4,
                                          "note": "plus 4 bytes per step probed for the pairs the coarse test cannot settle"}}
                bplan.close()
        finally:
            if os.path.exists(fpath):
                os.unlink(fpath)

    # The boxes of this pool run the cold step scan in one of two states (102 and 92-94 us at cfg-L), and a box changes from the
    # slow one to the fast one some tens of seconds into sustained work (the same library, alternating processes on one box:
    # tools/ab_wide.sh, profiles/NOTES.md R6.5) -- `frac_cold` above is sampled a few seconds into this process.  The same
    # sampling once more, at the end of the secondary measurements, says which state the box is in by then.
    if rank == 0 and world == 1 and roofline and roofline.get("cold") and extras is not None:
        os.environ["FLATGFA_MALL_MB"] = "0"
        try:
            plan0 = dev.DepthPlan(graph)
        finally:
            del os.environ["FLATGFA_MALL_MB"]
        cold = torch.zeros(2 * S, dtype=torch.int32, device=device)
        for _ in range(3):
            plan0.seg_depth(cold[:S], cold[S:])
        plan0.status()
        plan0.describe()
        torch.cuda.synchronize(device)
        dev.profile_enable(True)
        dev.profile_read()
        for _ in range(SAMPLE_STEPS):
            plan0.seg_depth(cold[:S], cold[S:])
        plan0.status()
        dev.profile_enable(False)
        late = [ms for n_, ms in dev.profile_read() if n_ == roofline["kernel"]]
        check_bufs([cold], "behind the cold plan's late sample")
        if late:
            late_ms = float(np.mean(late))
            roofline["cold"]["late"] = {"kernel_avg_ms": round(late_ms, 5), "n": len(late), "seconds_into_process": round(time.perf_counter() - T_PROCESS0, 1),
                                        "frac": round(roofline["algorithmic_bytes"] / (late_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        plan0.close()
        del plan0, cold

    # ---- CPU baseline: the oracle, one core, same arrays (rank 0, N=1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import flatgfa_oracle as fo
        pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
        fo.seg_depth_with_uniq(pools)  # warm-up: page in, build the .so
        times = []
        budget = time.perf_counter() + 25.0
        while len(times) < 5 and (len(times) < 2 or time.perf_counter() < budget):
            c0 = time.perf_counter()
            fo.seg_depth_with_uniq(pools)
            times.append(time.perf_counter() - c0)
        med = float(np.median(times))
        cpu = {"value": round(N / med, 1), "unit": "path-steps/s", "cores": 1, "kind": "port",
               "sample": f"full {args.workload} graph ({N} steps), seg_depth_with_uniq, median of {len(times)} runs "
                         f"after 1 warm-up, oracle/depth_oracle.c gcc -O3, host has {os.cpu_count()} logical cores",
               "seconds_median": round(med, 4)}
        # BASELINE.md section 3 also asks for a path-parallel run on all host cores (not the
        # reference's algorithm -- its loop is single-threaded -- so it is reported beside, not as, the baseline)
        nthr = max(1, min(os.cpu_count() or 1, 32))  # more threads only add reduction work
        if nthr > 1 and extras is not None:
            fo.seg_depth_with_uniq_mt(pools, nthr)
            mts = []
            for _ in range(3):
                c0 = time.perf_counter()
                fo.seg_depth_with_uniq_mt(pools, nthr)
                mts.append(time.perf_counter() - c0)
            extras["cpu_all_cores"] = {"value": round(N / float(np.median(mts)), 1), "unit": "path-steps/s",
                                       "cores": nthr, "kind": "port, path-parallel (pthreads)"}

    # how many ranks the collective really spans (an all-reduce of ones, on the data path's backend)
    ranks_seen = 1
    if world > 1:
        t = torch.ones(1, dtype=torch.int32, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks_seen = int(t.item())
    steps_all = [N_local]
    if world > 1:
        t = torch.zeros(world, dtype=torch.int64, device=coll_device)
        t[rank] = N_local
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        steps_all = [int(x) for x in t.cpu().tolist()]
    if rank == 0:
        value = N_job * args.steps / elapsed
        metric = f"path-steps/sec on `depth` ({S / 1e6:g}M seg / {N_job / 1e6:g}M step GFA)"
        if verified:
            metric += "; bit-exact vs flatgfa CPU"
        if world == 1:
            sharding = "none"
        elif strong:
            sharding = (f"one graph, paths cut into {world} contiguous groups of equal step counts; per step one sum "
                        "all-reduce of [depth|uniq] (8 bytes per segment), overlapping the next step's kernels")
        else:
            sharding = ("one graph per rank over the same segments; per step one sum all-reduce of [depth|uniq], "
                        "overlapping the next step's kernels")
        line = {
            "metric": metric,
            "value": round(value, 1), "unit": "path-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: seg_depth_with_uniq on synth(seed={'1' if strong or world == 1 else '1+rank'}, "
                                   f"S={S}, P={P}, L={L}, model={model})" + (" per GPU" if world > 1 and not strong else ""),
                       "segments": S, "paths_per_gpu": P_local if strong else P,
                       "steps_per_gpu": N_local, "steps_per_gpu_all_ranks": steps_all, "steps_per_job_step": N_job, "sharding": sharding,
                       "collective_bytes": 8 * S if world > 1 else 0, "calls_in_flight": in_flight,
                       "in_flight_through": ("flatgfa_dev_pipeline_* (C ABI)" if world == 1 and in_flight > 1 else "one plan per torch stream" if in_flight > 1 else None),
                       "pipeline": op.pipe.describe() if isinstance(op, PipelinedDepth) else None,
                       "ranks_seen": ranks_seen, "uses_rccl": bool(world > 1 and backend == "nccl")},
            "bit_exact_vs_oracle": verified,
            # one whole query with nothing else in flight (the loop of the same process that runs one call after the other);
            # `ms_per_step` is the throughput of `calls_in_flight` overlapped queries
            "one_call_ms": round(serial_elapsed / args.steps * 1e3, 5) if serial_elapsed else round(elapsed / args.steps * 1e3, 5),
            "verified_after": checked if verified else None,
            "roofline": roofline, "cpu_baseline": cpu, "commit": git_head(),
        }
        if allreduce_ms is not None:
            line["allreduce_ms"] = round(allreduce_ms, 5)
            line["allreduce_bytes"] = 8 * S
        if extras:
            line["extras"] = extras
        if cpu:
            line["speedup_vs_cpu_1core"] = round(value / cpu["value"], 2)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_host_c(args, torch, dist, pa, dev):
    """--host c: BASELINE.json configs[3] through the C ABI alone.  One process (rank 0) holds the
    graph and a flatgfa_sharded_t over all N devices: shards cut by libflatgfa.so, per-device plans
    and streams, ncclCommInitAll + ncclAllReduce inside the library.  A step = flatgfa_sharded_enqueue
    (local kernels + collective on every shard's stream); the timed region ends with
    flatgfa_sharded_sync.  Under torch.distributed.run the other ranks only take part in the barriers."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world not in (1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)  # (barriers only: the data path is rank 0's)
    S, P, L, model = WORKLOADS[args.workload]
    N = P * L
    line = None
    if rank == 0:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
        one_dev = os.environ.get("FLATGFA_BENCH_ONE_DEVICE") == "1"
        devices = [0] * args.gpus if one_dev else list(range(args.gpus))
        g = pa.synth(1, S, P, L, model, False)
        sh = pa.ShardedFlatGFA(g, args.gpus, devices=devices)
        lay = sh.layout()
        for _ in range(max(args.warmup, 1)):
            sh.enqueue(True)
        sh.sync()
        verified = None
        if not args.no_verify:
            from oracle import flatgfa_oracle as fo
            pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
            want_d, want_u = fo.seg_depth_with_uniq(pools)
            verified = True
            for i in range(args.gpus):  # every shard holds the reduced vectors
                d, u = sh.fetch(i)
                verified = verified and bool((d == want_d).all() and (u == want_u).all())
            if not verified:
                raise SystemExit("HIP result differs from the oracle: refusing to report a number")
        dev.profile_enable(False)
        dev.profile_read()
        sh.sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            sh.enqueue(True)
        sh.sync()
        elapsed = time.perf_counter() - t0
        # kernel durations of a few more steps, outside the timed region (the event records are process-wide)
        dev.profile_enable(True)
        for _ in range(3):
            sh.enqueue(True)
        sh.sync()
        dev.profile_enable(False)
        per = {}
        for name, ms in dev.profile_read():
            per.setdefault(name, []).append(ms)
        kern = {k: round(float(np.mean(v)), 5) for k, v in per.items()}
        dom = max(kern, key=lambda k: kern[k]) if kern else None
        N_local = max(x["step_end"] - x["step_begin"] for x in lay)
        roofline = None
        if dom:
            B_dom = kernel_bytes(dom, N_local, P // max(args.gpus, 1), S, 2)
            ach = B_dom / (kern[dom] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "kernel_avg_ms": kern[dom], "algorithmic_bytes": B_dom,
                        "kernels_avg_ms": kern, "kernel_timing": "HIP events around each launch of three steps after the timed region, all shards"}
        line = {
            "metric": f"path-steps/sec on `depth` ({S / 1e6:g}M seg / {N / 1e6:g}M step GFA)" + ("; bit-exact vs flatgfa CPU" if verified else ""),
            "value": round(N * args.steps / elapsed, 1), "unit": "path-steps/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: seg_depth_with_uniq on synth(seed=1, S={S}, P={P}, L={L}, model={model})",
                       "segments": S, "steps_per_job_step": N, "host": "c", "steps_per_gpu": N_local,
                       "steps_per_gpu_all_ranks": [x["step_end"] - x["step_begin"] for x in lay],
                       "collective_bytes": sh.collective_bytes(True) if args.gpus > 1 or lay[0]["rccl"] else 0,
                       "ranks_seen": sh.ranks_seen(), "uses_rccl": bool(lay[0]["rccl"]),
                       "sharding": f"flatgfa_sharded_* (C ABI, one process): {args.gpus} shards on devices {devices}, "
                                   f"{lay[0]['split_paths']} paths cut, exchange by " + ("RCCL ncclAllReduce inside libflatgfa.so" if lay[0]["rccl"] else "device-side adds (shards share a device)"),
                       "shards": [{k: x[k] for k in ("device", "step_begin", "step_end", "pieces")} for x in lay]},
            "bit_exact_vs_oracle": verified, "roofline": roofline, "cpu_baseline": None, "commit": git_head(),
        }
        sh.close()
    if world > 1:
        dist.barrier()
    if line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
