#!/usr/bin/env python3
"""bench.py -- path-steps/sec of node depth (+ unique depth) on the 1M-segment / 100M-step
synthetic graph (BASELINE.json metric; configs[2] at N=1).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfgL|cfgL-uniform|cfgL-short|cfgS]

One "step" = one pass of the hot path (seg_depth_with_uniq, ops/depth.rs:15-39) over the
rank's resident graph image: run the HIP kernels, and -- for N > 1 -- one RCCL sum all-reduce of
the fused [depth | uniq] vector.  Steps alternate between two result buffers, so the all-reduce
of one step overlaps the kernels of the next (pollen_amd/sharded.py); the timed region ends
when every kernel and every collective of its K steps has finished.  Inputs are in HBM before the timed
region starts.  For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
holds its own 1000 paths x 100k steps over the same 1M segments (weak scaling: the path set
grows with the GPU count, which is when sharding is warranted); value = all ranks' steps /
max-over-ranks time.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, from HIP events recorded
around each kernel launch on the launch stream inside the timed region; `cpu_baseline` is the
single-threaded C oracle (the reference's loop is single-threaded) timed on this host.
"""
import argparse
import json
import os
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
    "cfgL-short": (1_000_000, 100_000, 1000, "pangenome"),
    "cfgL-fewlong": (1_000_000, 100, 1_000_000, "pangenome"),
    "cfgL-medium": (1_000_000, 10_000, 10_000, "pangenome"),  # paths of ten blocks each
    "cfgL-4Mseg": (4_000_000, 1000, 100_000, "pangenome"),   # beyond one LDS bitset: four segment-range passes
    "cfgS": (10_000, 100, 10_000, "pangenome"),
}


def algorithmic_bytes(N, P, S, k):
    # each handle read once + path spans + k result vectors written once
    return 4 * N + 8 * P + 4 * S * k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfgL", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (depth-only, path depth, end-to-end, copy bandwidth)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import pollen_amd as pa
    from pollen_amd import device as dev
    from pollen_amd.sharded import ShardedDepth

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

    S, P, L, model = WORKLOADS[args.workload]
    N = P * L
    # ---- build the rank's graph and make it resident (not timed) ----
    g = pa.synth(1 + rank, S, P, L, model, False)
    steps, pb, pe, seg_len = g.soa()
    graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device=str(device))
    plan = dev.DepthPlan(graph)
    op = ShardedDepth(S, plan.seg_depth, device=device, with_uniq=True)

    def sync_all():
        op.finish()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        op.run()
    plan.status()
    sync_all()

    # ---- verification of what is being timed (rank-local partials vs the oracle; N=1: the result) ----
    verified = None
    if not args.no_verify and world == 1:
        from oracle import flatgfa_oracle as fo
        pools = fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})
        want_d, want_u = fo.seg_depth_with_uniq(pools)
        got = op.buf.cpu().numpy().view(np.uint32)
        verified = bool((got[:S] == want_d).all() and (got[S:] == want_u).all())
        if not verified:
            raise SystemExit("HIP result differs from the oracle: refusing to report a number")

    # ---- timed region: exactly K steps ----
    # Kernel durations come from HIP events recorded around each launch, inside this region, on
    # every EVENT_EVERY-th step: four event records per step cost ~8% of a 0.17 ms step, and the
    # value reported is the whole region's throughput.
    EVENT_EVERY = 4
    dev.profile_enable(False)
    dev.profile_read()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        dev.profile_enable(i % EVENT_EVERY == 0)
        op.run()
    sync_all()
    t1 = time.perf_counter()
    dev.profile_enable(False)
    kernels = dev.profile_read()
    n_timed_steps = len(range(0, args.steps, EVENT_EVERY))
    plan.status()

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations -> roofline for the dominant kernel ----
    per = {}
    for name, ms in kernels:
        per.setdefault(name, []).append(ms)
    kern_avg_ms = {k: float(np.mean(v)) for k, v in per.items()}
    dom = max(kern_avg_ms, key=lambda k: kern_avg_ms[k] * len(per[k])) if kern_avg_ms else None
    B = algorithmic_bytes(N, P, S, 2)
    device_ms_per_step = sum(kern_avg_ms[k] * len(per[k]) for k in per) / max(n_timed_steps, 1)
    roofline = None
    if dom:
        achieved = B / (kern_avg_ms[dom] * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel from the PMC counters: they need separate
        # rocprofv3 --pmc passes (tools/profile_round.sh), so the committed summary is quoted here.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
            if tj.get("workload") == args.workload and dom in tj["kernels"]:
                traffic = tj["kernels"][dom]["hbm_bytes"]
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel_avg_ms": round(kern_avg_ms[dom], 5), "algorithmic_bytes": B,
                    "all_kernels_ms_per_step": round(device_ms_per_step, 5),
                    "kernels_avg_ms": {k: round(v, 5) for k, v in kern_avg_ms.items()},
                    "kernel_timing": f"HIP events around each launch on steps 0, {EVENT_EVERY}, {2 * EVENT_EVERY}, ... "
                                     f"of the timed region ({n_timed_steps} of {args.steps} steps)"}

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

        # a2: depth only; a3: path depth of all paths (seg_depth + the per-path sums)
        d_only = torch.zeros(S, dtype=torch.int32, device=device)
        extras["seg_depth_only_ms"] = round(timed(lambda: plan.seg_depth(d_only, None)), 5)
        ids = torch.arange(P, dtype=torch.int32, device=device)
        len_out = torch.zeros(P, dtype=torch.int64, device=device)
        wsum_out = torch.zeros(P, dtype=torch.int64, device=device)

        def path_depth_all():
            plan.seg_depth(d_only, None)
            plan.path_sums(ids, d_only, len_out, wsum_out)
        extras["path_depth_all_paths_ms"] = round(timed(path_depth_all), 5)
        plan.status()
        # device-to-device copy bandwidth of this box (read + write bytes), for the roofline's second denominator
        a = torch.empty(1 << 28, dtype=torch.int32, device=device)
        b = torch.empty_like(a)
        copy_ms = timed(lambda: b.copy_(a), reps=10)
        extras["d2d_copy_gbs"] = round(2 * a.numel() * 4 / (copy_ms * 1e-3) / 1e9, 1)
        del a, b
        if roofline:
            roofline["frac_of_measured_copy"] = round(roofline["achieved"] / extras["d2d_copy_gbs"], 5)
        # end to end through the host API on a fresh handle: .flatgfa mmap -> H2D -> kernels -> D2H -> TSV text
        import tempfile
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        fpath = os.path.join(tmpdir, f"bench_{os.getpid()}.flatgfa")
        try:
            g.write_flatgfa(fpath)
            c0 = time.perf_counter()
            g2 = pa.load(fpath)
            c1 = time.perf_counter()
            g2.to_device(local_rank)
            c2 = time.perf_counter()
            g2.seg_depth_with_uniq()
            c3 = time.perf_counter()
            text = g2.depth_table()
            c4 = time.perf_counter()
            extras["end_to_end"] = {
                "what": ".flatgfa mmap -> H2D -> seg_depth_with_uniq (kernels + D2H + widen to u64) -> depth table text",
                "load_ms": round((c1 - c0) * 1e3, 3), "h2d_and_plan_ms": round((c2 - c1) * 1e3, 3),
                "first_query_ms": round((c3 - c2) * 1e3, 3), "table_ms": round((c4 - c3) * 1e3, 3),
                "table_bytes": len(text), "total_ms": round((c4 - c0) * 1e3, 3),
                "steps_per_s": round(N / (c4 - c0), 1)}
            # BASELINE.json configs[4]: which paths share an oriented handle with which (all pairs)
            g2.path_overlaps([0])  # builds the per-path handle bitsets (once per resident graph)
            c5 = time.perf_counter()
            touch = g2.path_overlaps(list(range(P)))
            c6 = time.perf_counter()
            extras["overlap_all_pairs"] = {"pairs": int(P) * int(P), "touching": int(touch.sum()),
                                           "ms": round((c6 - c5) * 1e3, 3), "note": "host API call incl. D2H of the P x P byte matrix"}
            g2.close()
        finally:
            if os.path.exists(fpath):
                os.unlink(fpath)

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

    if rank == 0:
        value = world * N * args.steps / elapsed
        line = {
            "metric": "path-steps/sec on `depth` (1M seg / 100M step GFA); bit-exact vs flatgfa CPU",
            "value": round(value, 1), "unit": "path-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: seg_depth_with_uniq on synth(seed=1+rank, S={S}, P={P}, L={L}, "
                                   f"model={model}) per GPU", "segments": S, "paths_per_gpu": P,
                       "steps_per_gpu": N, "sharding": "by path, one sum all-reduce of [depth|uniq] per step, overlapping the next step's kernels" if world > 1
                       else "none"},
            "bit_exact_vs_oracle": verified,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if extras:
            line["extras"] = extras
        if cpu:
            line["speedup_vs_cpu_1core"] = round(value / cpu["value"], 2)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
