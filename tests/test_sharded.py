"""Multi-GPU path sharding, exercised on CPU: the partition logic directly, and the
partial -> all-reduce combination under torch.distributed's gloo backend with world_size 2.
The per-rank partials come from the oracle here (test stand-in for the HIP kernels); the
partition, buffer fusion and collective are the product code in pollen_amd/sharded.py."""
import os
import socket

import numpy as np
import pytest

from oracle import flatgfa_oracle as fo
from oracle import synth
from pollen_amd.sharded import ShardedDepth, local_slice, shard_paths


def test_shard_paths_covers_and_balances():
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 4, 8):
        for P in (0, 1, 2, 7, 8, 100, 1000):
            lens = rng.integers(0, 5000, size=P)
            pb = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint32) if P else np.zeros(0, np.uint32)
            pe = (pb + lens).astype(np.uint32)
            cuts = shard_paths(pb, pe, world)
            assert len(cuts) == world and cuts[0][0] == 0 and cuts[-1][1] == P
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))          # contiguous, whole paths
            loads = [int(lens[lo:hi].sum()) for lo, hi in cuts]
            assert sum(loads) == int(lens.sum())
            if P:
                assert max(loads) <= lens.sum() / world + lens.max()          # ideal + one path


def test_shard_paths_equal_paths_split_evenly():
    pb = (np.arange(1000) * 100_000).astype(np.uint32)
    pe = pb + 100_000
    for world in (2, 4, 8):
        cuts = shard_paths(pb, pe, world)
        assert [hi - lo for lo, hi in cuts] == [1000 // world] * world


def test_c_route_cuts_are_shard_paths_cuts():
    """flatgfa_sharded_create's cut computation (flatgfa_shard_cuts: host only, no device) against the torch route's
    shard_paths: with whole paths the two are the same rule; allowed to cut inside a path, a cut is the nearest path
    boundary when that lies within an eighth of a shard's share of the even cut, and the even cut itself otherwise."""
    import pollen_amd as pa
    rng = np.random.default_rng(11)
    for world in (1, 2, 3, 4, 8):
        for P in (0, 1, 2, 7, 8, 100, 1000):
            for hi_len in (1, 5000, 1_000_000):
                lens = rng.integers(0, hi_len + 1, size=P).astype(np.uint64)
                ends = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
                total = int(ends[-1])
                pb = ends[:-1].astype(np.uint32)
                pe = ends[1:].astype(np.uint32)
                whole = pa.shard_cuts(lens, world, pa.SHARD_WHOLE_PATHS)
                want = [int(ends[lo]) for lo, _ in shard_paths(pb, pe, world)] + [total]
                assert whole.tolist() == want, (world, P, hi_len)
                cuts = pa.shard_cuts(lens, world)
                assert cuts[0] == 0 and cuts[-1] == total and (np.diff(cuts.astype(np.int64)) >= 0).all()
                for r in range(1, world):
                    target = total * r // world
                    near = min((int(e) for e in ends), key=lambda e: (abs(e - target), e))
                    if abs(near - target) * 8 * world <= total:
                        assert int(cuts[r]) == max(near, int(cuts[r - 1])), (world, P, hi_len, r)
                    else:
                        assert int(cuts[r]) == max(target, int(cuts[r - 1])), (world, P, hi_len, r)


def test_shard_cuts_rejects_bad_arguments():
    import pollen_amd as pa
    with pytest.raises(pa.FlatGFAError):
        pa.shard_cuts([1, 2, 3], 0)
    with pytest.raises(pa.FlatGFAError):
        pa.shard_cuts([1, 2, 3], 65)


def test_local_slice_rebases_spans():
    p = synth.pools(3, 200, 6, 50, "pangenome")
    pb, pe = p.paths["steps_start"], p.paths["steps_end"]
    s, b, e = local_slice(p.steps, pb, pe, 2, 5)
    assert len(s) == 150 and b.tolist() == [0, 50, 100] and e.tolist() == [50, 100, 150]
    assert (s == p.steps[100:250]).all()
    s, b, e = local_slice(p.steps, pb, pe, 3, 3)
    assert len(s) == 0 and len(b) == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, seed, S, P, L, model, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pools = synth.pools(seed, S, P, L, model)
        pb, pe = pools.paths["steps_start"], pools.paths["steps_end"]
        lo, hi = shard_paths(pb, pe, world)[rank]
        steps, b, e = local_slice(pools.steps, pb, pe, lo, hi)
        sub = fo.Pools(**{n: getattr(pools, n) for n in fo.POOL_ORDER})
        sub.steps = steps
        paths = np.zeros(hi - lo, dtype=fo.PATH_DT)
        paths["steps_start"], paths["steps_end"] = b, e
        sub.paths = paths

        def local_fn(depth, uniq):  # stand-in for DepthPlan.seg_depth on this rank's shard
            d, u = fo.seg_depth_with_uniq(sub)
            depth.copy_(torch.from_numpy(d.astype(np.uint32).view(np.int32)))
            uniq.copy_(torch.from_numpy(u.astype(np.uint32).view(np.int32)))

        op = ShardedDepth(S, local_fn, device="cpu", with_uniq=True)
        op.run()
        op.run()  # a second step must not accumulate
        op.run()  # a third one reuses the first buffer, whose collective must have finished
        op.finish()
        want_d, want_u = fo.seg_depth_with_uniq(pools)
        ok = (op.depth.numpy().view(np.uint32) == want_d).all() and (op.uniq.numpy().view(np.uint32) == want_u).all()
        # path depth: every rank measures its own paths against the reduced depth; results are concatenated
        from pollen_amd.sharded import gather_path_depth
        depth = op.depth.numpy().view(np.uint32).astype(np.uint64)
        seg_len = (pools.segs["seq_end"] - pools.segs["seq_start"]).astype(np.uint64)
        ids_all = pools.steps >> 1
        ln = np.array([seg_len[ids_all[pb[p]:pe[p]]].sum() for p in range(lo, hi)], dtype=np.uint64)
        ws = np.array([(depth[ids_all[pb[p]:pe[p]]] * seg_len[ids_all[pb[p]:pe[p]]]).sum() for p in range(lo, hi)], dtype=np.uint64)
        got_ln, got_mean = gather_path_depth(ln, ws)
        want_ln, want_mean = fo.path_depth(pools)
        ok = ok and (got_ln == want_ln).all() and got_mean.tobytes() == want_mean.tobytes()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [(1, 5000, 9, 700, "pangenome"), (2, 300, 2, 1000, "uniform"), (3, 64, 1, 10, "uniform")])
def test_gloo_world2_allreduce_matches_unsharded(shape):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, *shape, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True), (1, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_multirank_control_flow_on_one_gpu(scaling):
    """bench.py's N > 1 path end to end on a one-GPU box: two ranks share cuda:0, gloo stands in for
    RCCL.  The reduced vector is checked against the oracle inside bench.py (it refuses to print a
    line otherwise), for both ways of sharding."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FLATGFA_BENCH_ONE_DEVICE="1", FLATGFA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "cfgS", "--scaling", scaling],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["bit_exact_vs_oracle"] is True
    assert line["config"]["steps_per_job_step"] == (1_000_000 if scaling == "strong" else 2_000_000)
    assert line["allreduce_ms"] > 0 and line["roofline"]["kernel"]
    assert line["config"]["collective_bytes"] == 8 * 10_000
    assert line["config"]["steps_per_gpu_all_ranks"] == ([500_000, 500_000] if scaling == "strong" else [1_000_000, 1_000_000])


@pytest.mark.gpu
def test_bench_host_c_two_shards_on_one_gpu():
    """bench.py --host c under the launcher: rank 0 alone drives both shards through
    flatgfa_sharded_* (here on one device, so the exchange is the device-side add; on a node it is
    RCCL inside the library), rank 1 only keeps the barriers.  bench.py checks every shard's
    reduced vectors against the oracle before it prints a line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FLATGFA_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "cfgS", "--host", "c"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["bit_exact_vs_oracle"] is True
    assert line["config"]["host"] == "c" and len(line["config"]["shards"]) == 2
    assert sum(x["step_end"] - x["step_begin"] for x in line["config"]["shards"]) == 1_000_000
    # the same cut as the torch route's (shard_paths): both put the even cut at a path boundary here
    assert line["config"]["steps_per_gpu_all_ranks"] == [500_000, 500_000]
    assert line["config"]["collective_bytes"] == 8 * 10_000  # (no path cut: [depth | uniq] alone)


@pytest.mark.gpu
def test_bench_one_rank_under_the_launcher_is_the_plain_run():
    """`--gpus 1` launched through torch.distributed.run (how the driver's scaling run starts its N = 1 point)
    prints the same metric, config and workload as the plain `python bench.py`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "cfgS", "--no-extras", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    a = subprocess.run([sys.executable] + args, capture_output=True, text=True, env=env, timeout=600)
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port())] + args, capture_output=True, text=True, env=env, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-1000:], b.stderr[-1000:])
    la, lb = (json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]) for r in (a, b))
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "config", "bit_exact_vs_oracle"):
        assert la[k] == lb[k], k


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1000, 100, 8), (37, 5000, 4), (12, 20_000, 3), (9, 999, 2)])
def test_the_two_routes_cut_the_same_way_between_paths(shape):
    """shard_paths (one process per GPU) and flatgfa_sharded_create with whole paths (one process, the C ABI)
    put every cut at the same path boundary: the one nearest the even cut."""
    import pollen_amd as pa
    from pollen_amd.sharded import shard_paths
    P, L, n = shape
    g = pa.synth(3, 5000, P, L, "pangenome", False)
    _, pb, pe, _ = g.soa()
    want = shard_paths(pb, pe, n)
    with pa.ShardedFlatGFA(g, n, devices=[0] * n, flags=pa.SHARD_WHOLE_PATHS) as sh:
        lay = sh.layout()
    for r, (lo, hi) in enumerate(want):
        steps = int(pe[hi - 1]) - int(pb[lo]) if hi > lo else 0
        assert lay[r]["step_end"] - lay[r]["step_begin"] == steps and lay[r]["pieces"] == hi - lo, (r, lay[r], lo, hi)
