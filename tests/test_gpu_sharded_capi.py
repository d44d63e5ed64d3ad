"""flatgfa_sharded_* (include/flatgfa.h): one graph sharded over devices by one
process must give what the single-device calls give, bit for bit -- whole paths per shard, paths
cut between shards (unique depth counts a path once per segment however many of its pieces touch
it, depth.rs:30-34), more shards than paths, empty shards.  A one-GPU box runs the shards on
device 0 (the exchange is then a device-side add) and RCCL through a communicator of size one."""
import os

import numpy as np
import pytest

import pollen_amd as pa
from conftest import fixture_id, golden_gfas
from oracle import flatgfa_oracle as fo

pytestmark = pytest.mark.gpu


def pools_of(g):
    return fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})


def check_sharded(g, n_shards, flags=0, devices=None):
    pools = pools_of(g)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    want_len, want_mean = fo.path_depth(pools)
    devices = [0] * n_shards if devices is None else devices
    with pa.ShardedFlatGFA(g, n_shards, devices=devices, flags=flags) as sh:
        lay = sh.layout()
        d, u = sh.seg_depth_with_uniq()
        assert (d == want_d).all(), "depth"
        assert (u == want_u).all(), "uniq"
        assert (sh.seg_depth() == want_d).all(), "seg_depth"
        ln, mean = sh.path_depth()
        assert (ln == want_len).all(), "path lengths"
        assert mean.tobytes() == want_mean.tobytes(), "mean depth (bitwise, NaN included)"
        # the three-step form, twice (the second call must not see leftovers of the first), every shard holds the result
        for _ in range(2):
            sh.enqueue(True)
            sh.sync()
        for i in range(n_shards):
            d2, u2 = sh.fetch(i)
            assert (d2 == want_d).all() and (u2 == want_u).all(), f"shard {i} holds another vector"
        return lay, None


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_synthetic_graph_any_number_of_shards(n_shards):
    g = pa.synth(5, 30_000, 12, 20_000, "pangenome", True)
    lay, _ = check_sharded(g, n_shards)
    assert len(lay) == n_shards and all(not x["rccl"] for x in lay)  # (shards on one device exchange by adds)
    with pa.ShardedFlatGFA(g, n_shards, devices=[0] * n_shards) as sh:
        assert sh.ranks_seen() == n_shards
        # the cuts the handle made are the ones the host-only function computes
        lens = [20_000] * 12
        cuts = pa.shard_cuts(lens, n_shards)
        assert [x["step_begin"] for x in sh.layout()] == [int(c) for c in cuts[:-1]]
    # equal paths, twelve of them: 2 and 3 shards cut at path boundaries, 8 shards have to cut paths
    assert (lay[0]["split_paths"] > 0) == (n_shards == 8)


@pytest.mark.parametrize("model", ["pangenome", "chromosome", "uniform"])
def test_fewer_paths_than_shards(model):
    # two paths on five shards: every shard but one holds pieces only; a path's pieces revisit each other's segments
    g = pa.synth(9, 5_000, 2, 60_000, model, True)
    lay, _ = check_sharded(g, 5)
    assert lay[0]["split_paths"] == 2
    assert sum(x["pieces"] for x in lay) >= 5


def test_whole_paths_flag_never_cuts():
    g = pa.synth(9, 5_000, 2, 60_000, "pangenome", True)
    lay, _ = check_sharded(g, 5, flags=pa.SHARD_WHOLE_PATHS)
    assert lay[0]["split_paths"] == 0
    assert sorted(x["pieces"] for x in lay) == [0, 0, 0, 1, 1]


def test_uneven_paths_cut_where_no_boundary_is_near():
    # one path far longer than a shard's share among short ones
    rng = np.random.default_rng(3)
    S = 20_000
    lens = [200_000] + [int(x) for x in rng.integers(50, 4000, size=40)]
    ids = [((np.arange(n) * 7 + rng.integers(0, S)) % S).astype(np.uint32) for n in lens]
    lines = [f"S\t{i + 1}\tACGT" for i in range(S)]
    for k, walk in enumerate(ids):
        lines.append(f"P\tp{k}\t" + ",".join(f"{int(x) + 1}+" for x in walk) + "\t*")
    g = pa.parse_bytes(("\n".join(lines) + "\n").encode())
    lay, _ = check_sharded(g, 4)
    assert lay[0]["split_paths"] >= 1


@pytest.mark.parametrize("gfa", golden_gfas(), ids=fixture_id)
def test_fixtures_sharded_three_ways(gfa):
    # tiny graphs: most shards are empty, some paths are cut in the middle of a handful of steps
    g = pa.parse(gfa)
    check_sharded(g, 3)


def test_empty_graph_and_pathless_graph():
    for text in (b"", b"S\t1\tACGT\nS\t2\tGG\n"):
        g = pa.parse_bytes(text)
        with pa.ShardedFlatGFA(g, 2, devices=[0, 0]) as sh:
            d, u = sh.seg_depth_with_uniq()
            assert d.tolist() == [0] * g.segment_count and u.tolist() == [0] * g.segment_count
            assert sh.path_depth()[0].tolist() == []


def test_out_of_range_step_is_an_error_not_a_count():
    g = pa.synth(1, 2_000, 4, 5_000, "pangenome", False)
    steps, pb, pe, _ = g.soa()
    pools = pools_of(g)
    pools.steps = steps.copy()
    pools.steps[7_000] = (5_000 << 1)  # a handle naming a segment that is not there
    import tempfile  # (the bad graph gets to the library through the oracle's .flatgfa writer)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "bad.flatgfa")
        with open(path, "wb") as f:
            f.write(fo.dump_flatgfa(pools))
        with pa.load(path) as gb, pa.ShardedFlatGFA(gb, 2, devices=[0, 0]) as sh:
            with pytest.raises(pa.FlatGFAError) as ei:
                sh.seg_depth_with_uniq()
            assert ei.value.code == -2


def test_rccl_route_on_one_device(monkeypatch):
    # a communicator of size one: ncclCommInitAll + ncclAllReduce really run (librccl.so is loaded here, not before)
    monkeypatch.setenv("FLATGFA_SHARD_FORCE_RCCL", "1")
    g = pa.synth(2, 40_000, 20, 10_000, "pangenome", True)
    pools = pools_of(g)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    with pa.ShardedFlatGFA(g, 1, devices=[0]) as sh:
        assert sh.layout()[0]["rccl"]
        assert sh.ranks_seen() == 1  # (an all-reduce of ones over the communicator)
        for _ in range(3):
            d, u = sh.seg_depth_with_uniq()
            assert (d == want_d).all() and (u == want_u).all()
        ln, mean = sh.path_depth()
        want_len, want_mean = fo.path_depth(pools)
        assert (ln == want_len).all() and mean.tobytes() == want_mean.tobytes()
    with open("/proc/self/maps") as f:
        assert "librccl" in f.read()


def test_cfgL_sharded_eight_ways_on_one_device():
    # BASELINE.json configs[3]'s graph, eight shards (on one device here): 1000 paths of 100 k steps, cuts at path boundaries
    g = pa.synth(1, 1_000_000, 1000, 100_000, "pangenome", False)
    pools = pools_of(g)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    with pa.ShardedFlatGFA(g, 8, devices=[0] * 8) as sh:
        lay = sh.layout()
        assert [x["pieces"] for x in lay] == [125] * 8 and lay[0]["split_paths"] == 0
        d, u = sh.seg_depth_with_uniq()
        assert (d == want_d).all() and (u == want_u).all()


@pytest.mark.parametrize("n_shards,bits", [(2, 2), (4, 3), (8, 4), (13, 4), (20, 5)])
def test_touch_counters_travel_packed(n_shards, bits):
    """The cut paths' touch vectors share words in the collective: a path has at most n_shards pieces, so
    its count takes bits(n_shards) bits and 32 / bits of them fit a u32 -- eight shards: every cut path in
    ONE word per segment (12 bytes per segment on the wire, not 8 + 4 per cut path).  Three long paths over
    few segments: every shard holds pieces, pieces of one path revisit each other's segments, and one
    path is cut into more pieces than a narrower counter could count."""
    S = 3_000
    g = pa.synth(17, S, 3, 40_000, "pangenome", True)
    lay, _ = check_sharded(g, n_shards)
    K = lay[0]["split_paths"]
    assert K >= 1
    with pa.ShardedFlatGFA(g, n_shards, devices=[0] * n_shards) as sh:
        per_word = 32 // bits
        words = -(-K // per_word)
        assert sh.collective_bytes(True) == 4 * S * (2 + words)
        assert sh.collective_bytes(False) == 4 * S
        if n_shards == 8:
            assert words == 1
