"""Parity tests proper: the HIP path, called through the C ABI, against the oracle.

Bit-exact everywhere (integer work); the one f64 division per path must also match exactly
because both sides divide the same two integers.  Run with `-m gpu` on an MI355X.
"""
import os
import subprocess

import numpy as np
import pytest

import pollen_amd as pa
from conftest import GOLDEN, ROOT, fixture_id, golden_gfas
from oracle import flatgfa_oracle as fo
from oracle import synth

pytestmark = pytest.mark.gpu
FGFA = os.path.join(ROOT, "pollen_amd", "bin", "fgfa")


@pytest.fixture(params=["auto", "bucketed", "atomic", "tinycap", "pieces", "noshort", "handback", "parts3", "ranges", "dense",
                        "untagged", "untagged_pieces", "untagged_noshort", "groups3", "packed", "slots8", "claim_all", "wb11", "wgs96",
                        "owners", "owners8", "owners_pieces"])
def device_path(request, monkeypatch):
    """Runs a test once per device path: the default (up to 8 M steps the plan times the bucketed
    path against the atomic kernels on the graph at hand and keeps the faster), the bucketed path
    (short paths walked by single waves, k_scan_short; the rest by whole workgroups, k_scan), the simple global-atomic kernels,
    the bucketed path with 8-record buckets so that nearly every record takes the overflow route,
    long paths cut into 512-step pieces, k_scan alone (no wave-per-path kernels), and short paths
    sent to k_scan_short regardless of their run count (so that it has to hand some back), and
    three pass-2 workgroups per window whatever the graph's size (by default only small graphs
    share their windows out), and the segments cut into ranges of 40960 with one walk of the
    steps per range (by default only graphs beyond 16 M segments are; more than 64 ranges: the
    atomic kernels), and pass 1 without run detection (k_scan_dense: every step a record,
    partitioned by window in LDS; by default only for graphs with next to no runs).  All of those
    run tagged where the plan allows it (records that name their item, pass 2 walking whole
    sub-buckets; pieces of split paths claim in bitsets shared by a workgroup's waves); the
    "untagged" paths keep the directory walk (what plans with more split paths than pass 2 has
    bitsets for fall back to, and what path depth rides on).  The variables are read when a graph
    becomes resident."""
    monkeypatch.delenv("FLATGFA_DEPTH_PATH", raising=False)
    monkeypatch.delenv("FLATGFA_BUCKET_CAP", raising=False)
    monkeypatch.delenv("FLATGFA_PIECE_STEPS", raising=False)
    monkeypatch.delenv("FLATGFA_SHORT_MAX", raising=False)
    monkeypatch.delenv("FLATGFA_SHORT_ANY", raising=False)
    monkeypatch.delenv("FLATGFA_ACC_PARTS", raising=False)
    monkeypatch.delenv("FLATGFA_RANGE_SEGS", raising=False)
    monkeypatch.delenv("FLATGFA_DENSE", raising=False)
    monkeypatch.delenv("FLATGFA_BIG_GROUPS", raising=False)
    monkeypatch.delenv("FLATGFA_TAGGED", raising=False)
    monkeypatch.delenv("FLATGFA_NO_CLAIM", raising=False)
    monkeypatch.delenv("FLATGFA_SCAN_WGS", raising=False)
    monkeypatch.delenv("FLATGFA_PATH_GROUPS", raising=False)
    monkeypatch.delenv("FLATGFA_PACKED", raising=False)
    monkeypatch.delenv("FLATGFA_ACC_SLOTS", raising=False)
    monkeypatch.delenv("FLATGFA_WB", raising=False)
    monkeypatch.delenv("FLATGFA_ACC_OWN", raising=False)
    monkeypatch.delenv("FLATGFA_TAG_LIMIT", raising=False)
    if request.param.startswith("owners"):
        # the tagged walk that keeps track of which tag owns each of a wave's bitsets (k_accum<..., OWN>: by default only plans whose
        # sub-buckets hold sparse tags), on every plan: four bitsets, eight, and next to the shared ones of paths cut into pieces
        monkeypatch.setenv("FLATGFA_ACC_OWN", "1")
        monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
        if request.param == "owners8":
            monkeypatch.setenv("FLATGFA_ACC_SLOTS", "8")
        elif request.param == "owners_pieces":
            monkeypatch.setenv("FLATGFA_PIECE_STEPS", "4096")
    if request.param == "wb11":  # windows of 2048 segments (by default 4096, or 8192 beyond 4 M segments)
        monkeypatch.setenv("FLATGFA_WB", "11")
        monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    if request.param == "wgs96":  # pass 1 with 96 persistent workgroups, hence 96 sub-buckets per window (by default one per CU; a pipeline's lanes take 128 or 176)
        monkeypatch.setenv("FLATGFA_SCAN_WGS", "96")
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    if request.param == "claim_all":  # every item claims in a bitset, strictly monotone paths included (by default their records skip the claim: kTagNoClaim)
        monkeypatch.setenv("FLATGFA_NO_CLAIM", "0")
        monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    if request.param == "slots8":  # eight private bitsets per pass-2 wave wherever the plan allows them (by default only where a workgroup of pass 1 takes more than four items)
        monkeypatch.setenv("FLATGFA_ACC_SLOTS", "8")
        monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    if request.param == "packed":  # record buckets laid out to the count (by default only where the even layout would take gigabytes), every path an item of k_scan
        monkeypatch.setenv("FLATGFA_PACKED", "1")
        monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    if request.param == "groups3":  # the paths walked in three groups, the second and third adding to the first's counts (by default only plans with more items per workgroup than a tag can name)
        monkeypatch.setenv("FLATGFA_PATH_GROUPS", "3")
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    if request.param.startswith("untagged"):
        monkeypatch.setenv("FLATGFA_TAGGED", "0")
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    if request.param == "dense":
        monkeypatch.setenv("FLATGFA_DENSE", "1")
    if request.param in ("dense", "pieces", "ranges", "untagged_pieces"):  # pass 2's one-item shortcut on (elsewhere the plan times it)
        monkeypatch.setenv("FLATGFA_BIG_GROUPS", "1")
    if request.param == "ranges":
        monkeypatch.setenv("FLATGFA_RANGE_SEGS", "40960")
    if request.param == "parts3":
        monkeypatch.setenv("FLATGFA_ACC_PARTS", "3")
    if request.param in ("noshort", "untagged_noshort"):
        monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    if request.param == "handback":  # short paths go to k_scan_short whatever their run count: it hands back what does not fit
        monkeypatch.setenv("FLATGFA_SHORT_ANY", "1")
    if request.param in ("pieces", "untagged_pieces"):   # every path longer than 512 steps is scanned as several pieces + k_merge
        monkeypatch.setenv("FLATGFA_PIECE_STEPS", "512")
    if request.param == "bucketed":
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    if request.param == "atomic":
        monkeypatch.setenv("FLATGFA_DEPTH_PATH", "atomic")
    elif request.param == "tinycap":
        monkeypatch.setenv("FLATGFA_BUCKET_CAP", "8")
    return request.param


def read(path):
    with open(path, "rb") as f:
        return f.read()


def pools_of(g: pa.FlatGFA) -> fo.Pools:
    return fo.Pools(**{n: g.pool(n) for n in fo.POOL_ORDER})


def check_graph(g: pa.FlatGFA, pools: fo.Pools, path_ids=None):
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    d, u = g.seg_depth_with_uniq()
    assert d.dtype == np.uint64 and (d == want_d).all(), "depth"
    assert (u == want_u).all(), "uniq"
    assert (g.seg_depth() == want_d).all(), "seg_depth"
    want_len, want_mean = fo.path_depth(pools, path_ids)
    ln, mean = g.path_depth(path_ids)
    assert (ln == want_len).all(), "path lengths"
    assert mean.tobytes() == want_mean.tobytes(), "mean depth (bitwise, NaN included)"


@pytest.mark.parametrize("gfa", golden_gfas(), ids=fixture_id)
def test_fixtures_match_oracle_and_golden(gfa, device_path):
    g = pa.parse(gfa)
    pools = fo.parse_gfa(read(gfa))
    check_graph(g, pools)
    # the text boundary: byte-identical to slow_odgi's table and to the oracle's emitters
    assert g.depth_table() == read(gfa[:-4] + ".depth.tsv")
    assert g.path_depth_table() == fo.fgfa_depth(pools, False)


def test_known_answers():
    # flatgfa-sh/README.md:31-36,51-59,267-270 (stand-ins) and slow_odgi/README.md:144-178
    g = pa.parse(os.path.join(GOLDEN, "standin_note5.gfa"))
    assert g.depth_table() == b"#node.id\tdepth\tdepth.uniq\n1\t2\t2\n2\t0\t0\n3\t2\t2\n4\t2\t2\n"
    assert g.path_depth_table() == b"#path\tstart\tend\tmean.depth\n5\t0\t13\t2\n5-\t0\t13\t2\n"
    assert g.path_depth_table([b"zzz", b"5-"]) == b"#path\tstart\tend\tmean.depth\n5-\t0\t13\t2\n"
    assert g.path_depth_table([b"zzz"]) == b"#path\tstart\tend\tmean.depth\n"
    g = pa.parse(os.path.join(GOLDEN, "standin_k.gfa"))
    assert g.path_depth_table() == b"#path\tstart\tend\tmean.depth\nx\t0\t50\t1.9\ny\t0\t50\t1.9\n"
    g = pa.parse(os.path.join(GOLDEN, "kat_slow_odgi_readme.gfa"))
    d, u = g.seg_depth_with_uniq()
    assert d.tolist() == [2, 0, 4, 2] and u.tolist() == [2, 0, 3, 2]


def test_residency_timing_and_repeated_uploads(tmp_path):
    """to_device keeps its pinned staging buffers for the next graph (large step arrays go up in
    8 MB chunks from four threads): several graphs, in memory and file-mapped, one after the other
    and interleaved, each with the right answer; residency_ms is an error before, two durations after."""
    g0 = pa.parse(os.path.join(GOLDEN, "standin_note5.gfa"))
    with pytest.raises(pa.FlatGFAError):
        g0.residency_ms()
    graphs = []
    for seed, steps in ((3, 90_000), (4, 110_000), (5, 9_000)):   # 9, 11 and 0.9 M steps: the first two take the chunked route
        g = pa.synth(seed, 60_000, 100, steps, "pangenome", True)
        f = str(tmp_path / f"g{seed}.flatgfa")
        g.write_flatgfa(f)
        graphs += [g, pa.load(f)]
    for g in graphs[::2] + graphs[1::2]:
        g.to_device(0)
        h2d, plan = g.residency_ms()
        assert h2d > 0 and plan > 0
    for g in graphs:
        want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
        d, u = g.seg_depth_with_uniq()
        assert (d == want_d).all() and (u == want_u).all()
        g.close()


def test_cli_depth_is_byte_identical():
    for name in ("standin_note5", "ref_ex2", "edge_names_loops"):
        gfa = os.path.join(GOLDEN, name + ".gfa")
        pools = fo.parse_gfa(read(gfa))
        out = subprocess.run([FGFA, "-I", gfa, "depth", "-d"], capture_output=True, check=True).stdout
        assert out == read(os.path.join(GOLDEN, name + ".depth.tsv"))
        out = subprocess.run([FGFA, "-I", gfa, "depth"], capture_output=True, check=True).stdout
        assert out == fo.fgfa_depth(pools, False)
    gfa = os.path.join(GOLDEN, "edge_names_loops.gfa")
    out = subprocess.run([FGFA, "-I", gfa, "depth", "-r", "gamma", "-r", "nope", "-r", "alpha"],
                         capture_output=True, check=True).stdout
    assert out == fo.fgfa_depth(fo.parse_gfa(read(gfa)), False, [b"gamma", b"nope", b"alpha"])


def test_flatgfa_file_input(tmp_path):
    g = pa.synth(3, 5000, 40, 2000, "pangenome", True)
    f = tmp_path / "s.flatgfa"
    g.write_flatgfa(str(f))
    h = pa.load(str(f))  # zero-copy view of an unaligned file image
    check_graph(h, pools_of(g))
    out = subprocess.run([FGFA, "-i", str(f), "depth", "-d"], capture_output=True, check=True).stdout
    assert out == fo.fgfa_depth(pools_of(g), True)


EDGE_TEXTS = [
    b"",                                               # no segments, no paths
    b"S\t1\tACGT\n",                                   # segments but no paths
    b"S\t1\tACGT\nP\te\t\t*\n",                        # a path with zero steps (mean = NaN)
    b"S\t1\tACGT\nS\t2\tA\nP\te\t\t*\nP\tp\t2+,2-,2+\t*\nP\tq\t1+\t*\n",
    b"S\t7\tAC\nP\ta\t7+\t*\nP\ta\t7-\t*\n",           # duplicate path names are distinct paths
    b"S\t1\t*\nP\tp\t1+,1+\t*\n",
]


@pytest.mark.parametrize("text", EDGE_TEXTS, ids=range(len(EDGE_TEXTS)))
def test_edge_graphs(text, device_path):
    g = pa.parse_bytes(text)
    pools = fo.parse_gfa(text)
    check_graph(g, pools)
    assert g.depth_table() == fo.fgfa_depth(pools, True)
    assert g.path_depth_table() == fo.fgfa_depth(pools, False)


SHAPES = [
    # seed, S, P, L, model
    (1, 1, 1, 1, "pangenome"),
    (2, 1, 7, 1000, "pangenome"),          # every step on one segment: worst-case contention
    (3, 31, 3, 64, "uniform"),
    (4, 33, 5, 65, "pangenome"),
    (5, 1000, 1, 100_000, "pangenome"),    # one long path
    (6, 100_000, 2000, 50, "pangenome"),   # many short paths
    (7, 100_000, 3, 1, "uniform"),
    (8, 65_537, 257, 1023, "uniform"),
    (9, 1_048_576, 8, 20_000, "pangenome"),  # exactly one LDS window
    (10, 1_048_577, 8, 20_000, "uniform"),   # one segment past 256 windows: every path is a k_scan item
    (11, 2_500_000, 6, 30_000, "uniform"),   # 611 windows
    (12, 50_000, 3, 700_001, "pangenome"),   # few long paths: split into pieces by default
    (13, 3_000, 2, 400_000, "uniform"),      # pieces of one path revisit the same segments heavily
    (14, 200_000, 5000, 1000, "pangenome"),  # short paths, one block each (k_scan_short)
    (15, 50_000, 300, 2048, "uniform"),      # short paths with too many runs: handed back to k_scan
    (16, 70_000, 900, 1500, "pangenome"),    # two blocks per short path
    (17, 999, 4000, 17, "pangenome"),        # paths shorter than a lane chunk, every alignment
    (18, 1_500_000, 3000, 700, "pangenome"),   # short paths beyond 256 windows: 3000 k_scan items of less than a block
    (19, 1_200_000, 5, 300_000, "uniform"),    # long paths cut into pieces, beyond 256 windows
    (20, 400_000, 600, 9000, "pangenome"),     # medium paths: single waves with the larger hash set
    (21, 1_300_000, 90, 15_000, "pangenome"),  # medium-length paths beyond 256 windows
    (22, 90_000, 40, 6000, "uniform"),         # too many runs for the medium kernel: k_scan
    (23, 5_000_000, 7, 60_000, "pangenome"),   # more than 4 M segments: 8192-segment windows in both passes
    (24, 9_000_001, 40, 3000, "uniform"),      # 8192-segment windows, short paths as k_scan items, a ragged last window
    (25, 70_000, 3, 1_048_576 + 1040, "pangenome"),  # long paths whose pieces end in partial blocks and odd tails
    (27, 200_000, 12, 40_000, "chromosome"),   # paths along the graph, every other one downwards (k_scan's step -1 runs)
    (28, 1_500_000, 7, 250_000, "chromosome"),
    (29, 30_000, 200, 900, "chromosome"),      # the wave-per-path kernels see downward paths as runs of one
    (31, 300_000, 40, 60_000, "haplotype"),    # haplotype walks: strictly monotone but for those that wrap around the last segment (records that skip pass 2's claim)
    (32, 2_000_000, 9, 500_000, "haplotype"),  # ... long ones, cut into pieces that are put together again by the plan
    (33, 400_000, 50, 50_000, "repeats"),      # haplotype walks that go back over a few segments now and then: no-claim records block by block
    (26, 17_000_000, 3, 4000, "uniform"),      # beyond 2048 windows of 8192 segments: two ranges, two walks of the steps
    (30, 40_000_000, 5, 60_000, "chromosome"), # three ranges; runs that straddle a range boundary are split between the walks
]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"S{s[1]}_P{s[2]}_L{s[3]}_{s[4]}" for s in SHAPES])
def test_synthetic_shapes(shape, device_path):
    seed, S, P, L, model = shape
    g = pa.synth(seed, S, P, L, model, False)
    pools = pools_of(g)
    ids = None if P <= 64 else np.arange(P - 1, -1, -7, dtype=np.uint32)  # reversed, strided subset
    check_graph(g, pools, ids)


def test_arbitrary_and_overlapping_spans(tmp_path, device_path):
    # The Path type allows any spans: overlapping, nested, out of order, gaps.
    pools = synth.pools(12, 500, 4, 1000, "pangenome")
    pools.paths["steps_start"] = [100, 0, 3500, 100]
    pools.paths["steps_end"] = [2100, 4000, 3500, 2100]
    f = tmp_path / "spans.flatgfa"
    f.write_bytes(fo.dump_flatgfa(pools))
    check_graph(pa.load(str(f)), pools, np.array([3, 3, 0, 2, 1], dtype=np.uint32))


def test_out_of_range_ids_are_errors(tmp_path, device_path):
    pools = synth.pools(13, 100, 2, 50, "pangenome")
    bad = fo.Pools(**{n: getattr(pools, n).copy() for n in fo.POOL_ORDER})
    bad.steps[77] = (100 << 1) | 1  # segment id == n_segs
    f = tmp_path / "badseg.flatgfa"
    f.write_bytes(fo.dump_flatgfa(bad))
    g = pa.load(str(f))
    with pytest.raises(pa.FlatGFAError) as e:
        g.seg_depth_with_uniq()
    assert e.value.code == -2
    with pytest.raises(pa.FlatGFAError):
        g.seg_depth()
    bad = fo.Pools(**{n: getattr(pools, n).copy() for n in fo.POOL_ORDER})
    bad.paths["steps_end"][1] = 101  # span past the steps pool
    f = tmp_path / "badspan.flatgfa"
    f.write_bytes(fo.dump_flatgfa(bad))
    with pytest.raises(pa.FlatGFAError):
        pa.load(str(f)).seg_depth()
    with pytest.raises(pa.FlatGFAError):
        pa.synth(1, 10, 2, 5).path_depth([0, 2])  # path id out of range


def test_cfgS_matches_slow_odgi_golden():
    # BASELINE.json configs[1]: 10k segments / 1M steps on one MI355X, bit-exact vs slow_odgi
    g = pa.synth(1, 10_000, 100, 10_000, "pangenome", True)
    assert g.depth_table() == read(os.path.join(GOLDEN, "synth_cfgS.depth.tsv"))
    check_graph(g, pools_of(g))


@pytest.mark.parametrize("name,cfg", [
    ("synth_short", (7, 8_000, 600, 800, "pangenome")),    # k_scan_short
    ("synth_long", (9, 12_000, 8, 70_000, "pangenome")),   # k_scan
    ("synth_uniform", (11, 6_000, 40, 3_000, "uniform")),  # hardly any runs
    ("synth_chrom", (13, 15_000, 12, 50_000, "chromosome")),  # k_scan's downward runs
])
def test_more_synthetic_graphs_match_slow_odgi_goldens(name, cfg, device_path):
    g = pa.synth(*cfg, True)
    assert g.depth_table() == read(os.path.join(GOLDEN, name + ".depth.tsv"))


@pytest.mark.parametrize("model", ["pangenome", "uniform", "chromosome", "haplotype", "repeats"])
def test_cfgL_full_size(model):
    # BASELINE.json configs[2]: 1M segments / 100M steps.  Checked against the oracle (a few
    # seconds of CPU) and through size-independent properties.
    S, P, L = 1_000_000, 1000, 100_000
    g = pa.synth(1, S, P, L, model, False)
    d, u = g.seg_depth_with_uniq()
    assert int(d.sum()) == P * L                       # every step counted once
    assert (u <= d).all() and (u <= P).all() and ((d > 0) == (u > 0)).all()
    pools = pools_of(g)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    assert (d == want_d).all() and (u == want_u).all()
    ln, mean = g.path_depth()
    want_ln, want_mean = fo.path_depth(pools)
    assert (ln == want_ln).all() and mean.tobytes() == want_mean.tobytes()


def test_repeat_calls_are_idempotent(device_path):
    g = pa.synth(21, 20_000, 50, 4000, "pangenome", False)
    a = g.seg_depth_with_uniq()
    for _ in range(3):
        b = g.seg_depth_with_uniq()
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    assert (g.seg_depth() == a[0]).all()


@pytest.mark.parametrize("n_segs", [70_000, 1_100_000, 2_200_000])
def test_runs_that_end_at_the_last_segment(n_segs, device_path):
    """Runs ending exactly at the last segment (of the graph, hence of the last segment-range
    pass), called repeatedly: nothing a call leaves behind in the scratch may leak into the next
    one (found by tools/fuzz_gpu.py: the overflow route's -1 past the last segment did)."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    rng = np.random.default_rng(5)
    S = n_segs
    lens = [5000, 30_000, 700, 90_000, 1500, 12_000]
    ids = []
    for L in lens:  # walks that wrap around S, so that many runs end at S - 1 and start at 0
        start = S - int(rng.integers(1, L))
        ids.append((start + np.arange(L)) % S)
    steps = (np.concatenate(ids).astype(np.uint32) << 1)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - np.array(lens, dtype=np.uint32)).astype(np.uint32)
    paths = np.zeros(len(lens), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, steps, np.zeros(S, dtype=fo.SEG_DT)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for _ in range(3):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()
        assert (u.cpu().numpy().view(np.uint32) == want_u).all()
        plan.seg_depth(d, None)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()


@pytest.mark.parametrize("n_segs,density", [(60_000, 0.5), (300_000, 0.97), (1_300_000, 0.7), (5_000_000, 0.6)])
def test_paths_that_run_along_the_graph(n_segs, density, device_path):
    """Paths shaped like real chromosome walks rather than the benchmark's random one: each visits
    a random subset of the segments in order -- upwards, downwards (a contig on the reverse
    strand: k_scan finds its runs with step -1), or switching direction a few times -- and one
    walks a stretch twice.  A wave's runs then fall into one window, downward runs are emitted
    from their low end, and a piece that runs both ways is scanned in its majority direction."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    rng = np.random.default_rng(n_segs)
    S = n_segs
    walks = []
    for p in range(9):
        lo = int(rng.integers(0, S // 3))
        hi = int(rng.integers(2 * S // 3, S))
        ids = lo + np.nonzero(rng.random(hi - lo) < density)[0]
        if p % 3 == 1:
            ids = ids[::-1]
        elif p % 3 == 2:  # up, down, up, down over different stretches
            cuts = np.sort(rng.integers(0, len(ids), size=3))
            parts = np.split(ids, cuts)
            ids = np.concatenate([part[::-1] if k % 2 else part for k, part in enumerate(parts)])
        if p == 4:
            ids = np.concatenate([ids, ids[: len(ids) // 3]])  # revisits, downwards
        walks.append(ids.astype(np.uint32))
    walks.append(np.arange(S - 1, S - 1 - min(S, 3000), -1, dtype=np.uint32))  # down from the last segment
    walks.append(np.arange(min(S, 2500) - 1, -1, -1, dtype=np.uint32))         # down to segment 0
    lens = np.array([len(x) for x in walks], dtype=np.uint32)
    steps = (np.concatenate(walks) << 1) | rng.integers(0, 2, size=int(lens.sum())).astype(np.uint32)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - lens).astype(np.uint32)
    paths = np.zeros(len(lens), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, steps, np.zeros(S, dtype=fo.SEG_DT)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for _ in range(2):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()
        assert (u.cpu().numpy().view(np.uint32) == want_u).all()
    plan.seg_depth(d, None)
    plan.status()
    assert (d.cpu().numpy().view(np.uint32) == want_d).all()


def test_linearity_over_path_subsets():
    # depth and uniq are sums of per-path contributions: computing two disjoint path groups
    # separately and adding must equal the whole (this is what multi-GPU sharding relies on).
    from pollen_amd.device import DepthPlan, DeviceGraph
    from pollen_amd.sharded import local_slice, shard_paths
    import torch
    g = pa.synth(22, 30_000, 37, 3000, "pangenome", False)
    steps, pb, pe, _ = g.soa()
    whole_d, whole_u = g.seg_depth_with_uniq()
    acc = torch.zeros(2 * 30_000, dtype=torch.int32, device="cuda:0")
    for lo, hi in shard_paths(pb, pe, 3):
        s, b, e = local_slice(steps, pb, pe, lo, hi)
        plan = DepthPlan(DeviceGraph(s, b, e, 30_000))
        part = torch.zeros_like(acc)
        plan.seg_depth(part[:30_000], part[30_000:])
        plan.status()
        acc += part
    got = acc.cpu().numpy().view(np.uint32)
    assert (got[:30_000] == whole_d).all() and (got[30_000:] == whole_u).all()


def test_device_path_depth_all_matches_two_walks(device_path):
    """flatgfa_dev_path_depth_all (sums formed in pass 2 from the run records) against
    seg_depth + path_sums (a second walk of the steps) and against the oracle, on a graph with
    short, medium, long and split paths."""
    import torch
    from pollen_amd import device as dev
    g = pa.synth(12, 60_000, 40, 30_000, "pangenome", True)
    pools = pools_of(g)
    steps, pb, pe, seg_len = g.soa()
    # mix in short paths: cut the first ten paths into pieces of 700 steps
    nb, ne = [], []
    for p in range(len(pb)):
        if p < 10:
            for s in range(int(pb[p]), int(pe[p]), 700):
                nb.append(s)
                ne.append(min(s + 700, int(pe[p])))
        else:
            nb.append(int(pb[p]))
            ne.append(int(pe[p]))
    nb, ne = np.array(nb, np.uint32), np.array(ne, np.uint32)
    S, P = len(seg_len), len(nb)
    graph = dev.DeviceGraph(steps, nb, ne, S, seg_len)
    plan = dev.DepthPlan(graph)
    d = torch.zeros(S, dtype=torch.int32, device="cuda")
    ln = torch.zeros(P, dtype=torch.int64, device="cuda")
    ws = torch.zeros(P, dtype=torch.int64, device="cuda")
    plan.path_depth_all(d, ln, ws)
    plan.status()
    d2 = torch.zeros_like(d)
    ln2, ws2 = torch.zeros_like(ln), torch.zeros_like(ws)
    plan.seg_depth(d2, None)
    plan.status()   # (with 8-record buckets the call is only complete after this)
    plan.path_sums(torch.arange(P, dtype=torch.int32, device="cuda"), d2, ln2, ws2)
    plan.status()
    assert torch.equal(d, d2) and torch.equal(ln, ln2) and torch.equal(ws, ws2)
    want_d, _ = fo.seg_depth_with_uniq(pools)
    assert (d.cpu().numpy().view(np.uint32) == want_d).all()
    ids_all = steps >> 1
    lens = seg_len.astype(np.uint64)
    want_ln = np.array([lens[ids_all[b:e]].sum() for b, e in zip(nb, ne)], dtype=np.uint64)
    want_ws = np.array([(want_d[ids_all[b:e]].astype(np.uint64) * lens[ids_all[b:e]]).sum() for b, e in zip(nb, ne)], dtype=np.uint64)
    assert (ln.cpu().numpy().view(np.uint64) == want_ln).all() and (ws.cpu().numpy().view(np.uint64) == want_ws).all()
    plan.close()


@pytest.mark.parametrize("n_paths,steps", [(6000, 7), (900, 150), (40, 9000)])
def test_tagged_walk_many_items_per_step(n_paths, steps, monkeypatch):
    """Pass 2 of a tagged call keeps four bitsets per wave.  With every path an item of k_scan and a
    handful of steps per path, one 64-record step of a sub-bucket spans dozens of items (the
    general route: rounds of four tags, lowest first), and the waves of pass 1 run items ahead of
    each other (its gate: four for a wave with records to append).  All paths crowd into a few
    windows, revisit each other's segments and their own."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")       # every path is an item of k_scan
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.delenv("FLATGFA_TAGGED", raising=False)
    rng = np.random.default_rng(n_paths)
    S = 3 * 4096 + 100
    walks = []
    for p in range(n_paths):
        start = int(rng.integers(0, S))
        jumps = rng.choice([1, 1, 1, 1, 2, -1, -3, 5, 4096, -4000], size=steps)
        ids = (start + np.cumsum(jumps)) % S
        walks.append(ids.astype(np.uint32))
    lens = np.array([len(x) for x in walks], dtype=np.uint32)
    stp = (np.concatenate(walks) << 1) | rng.integers(0, 2, size=int(lens.sum())).astype(np.uint32)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - lens).astype(np.uint32)
    paths = np.zeros(len(lens), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, stp, np.zeros(S, dtype=fo.SEG_DT)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    plan = DepthPlan(DeviceGraph(stp, pb, pe, S))
    assert "pass2=tagged" in plan.describe()
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for _ in range(3):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()
        assert (u.cpu().numpy().view(np.uint32) == want_u).all()
    plan.close()


@pytest.mark.parametrize("always", [False, True])
@pytest.mark.parametrize("tagged", [True, False])
def test_wave_per_path_kernels_alone(always, tagged, monkeypatch):
    """When every path goes to the wave-per-path kernels, k_scan has nothing to walk and nothing can
    be handed back to it (the lists are made from exact run counts): its launch is left out and
    pass 2 takes every record as an earlier one.  FLATGFA_SCAN_ALWAYS keeps the launch (its
    workgroups only save the cursors); both routes, with and without unique depth, several calls."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.delenv("FLATGFA_SHORT_MAX", raising=False)
    monkeypatch.delenv("FLATGFA_SHORT_ANY", raising=False)
    if always:
        monkeypatch.setenv("FLATGFA_SCAN_ALWAYS", "1")
    else:
        monkeypatch.delenv("FLATGFA_SCAN_ALWAYS", raising=False)
    if tagged:
        monkeypatch.delenv("FLATGFA_TAGGED", raising=False)
    else:
        monkeypatch.setenv("FLATGFA_TAGGED", "0")
    g = pa.synth(11, 50_000, 3000, 700, "pangenome", False)   # short paths only (k_scan_short)
    g2 = pa.synth(12, 50_000, 300, 4000, "pangenome", False)   # more runs than a short path may have: the medium variant
    g3 = pa.synth(13, 50_000, 900, 1500, "chromosome", False)  # half of them walked from reversed copies
    for gr in (g, g2, g3):
        steps, pb, pe, _ = gr.soa()
        pools = pools_of(gr)
        want_d, want_u = fo.seg_depth_with_uniq(pools)
        S = gr.segment_count
        plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
        if gr is g:
            assert " items=0 " in plan.describe() and "pass1=k_scan_short " in plan.describe(), plan.describe()
        d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        for k in range(4):
            plan.seg_depth(d, u if k % 2 == 0 else None)
            plan.status()
            assert (d.cpu().numpy().view(np.uint32) == want_d).all()
            if k % 2 == 0:
                assert (u.cpu().numpy().view(np.uint32) == want_u).all()
        plan.close()


@pytest.mark.parametrize("tagged", [True, False])
def test_more_windows_than_the_directory_allows(tagged, monkeypatch):
    """A tagged plan may have 4096 windows per range (k_scan keeps one LDS table per window then,
    not two); a plan that cannot be tagged is cut into ranges of at most 2048.  17 M segments:
    2076 windows of 8192."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for v in ("FLATGFA_RANGE_SEGS", "FLATGFA_MAX_WINDOWS", "FLATGFA_SHORT_MAX", "FLATGFA_WB", "FLATGFA_PIECE_STEPS"):
        monkeypatch.delenv(v, raising=False)
    if tagged:
        monkeypatch.delenv("FLATGFA_TAGGED", raising=False)
    else:
        monkeypatch.setenv("FLATGFA_TAGGED", "0")
    S = 17_000_000
    g = pa.synth(21, S, 1000, 10_000, "pangenome", False)
    steps, pb, pe, _ = g.soa()
    want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
    text = plan.describe()
    assert ("windows=2076x8192 ranges=1" in text and "pass2=tagged" in text) if tagged else ("ranges=2" in text and "pass2=directory" in text), text
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for k in range(3):
        plan.seg_depth(d, u if k != 1 else None)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()
        if k != 1:
            assert (u.cpu().numpy().view(np.uint32) == want_u).all()
    plan.close()


@pytest.mark.parametrize("n_segs,ranges", [(17_000_000, 1), (34_000_000, 2)])
def test_bucket_arrays_beyond_four_gigabytes(n_segs, ranges, monkeypatch):
    """A plan whose (windows x sub-buckets x capacity) reaches 2^30 records addresses its buckets
    with 64-bit offsets (builds of k_scan of their own, whole-graph and ranged): what a graph of
    many windows whose paths run along it needs -- most sub-buckets empty, the others deep.  Forced
    here with a capacity of 2048 on 2076 windows per range (4.4 GB of scratch each)."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_BUCKET_CAP", "2048")
    for v in ("FLATGFA_RANGE_SEGS", "FLATGFA_MAX_WINDOWS", "FLATGFA_SHORT_MAX", "FLATGFA_WB", "FLATGFA_PIECE_STEPS", "FLATGFA_TAGGED", "FLATGFA_BUCKET_GB"):
        monkeypatch.delenv(v, raising=False)
    g = pa.synth(23, n_segs, 1000, 10_000, "chromosome", False)
    steps, pb, pe, _ = g.soa()
    want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
    plan = DepthPlan(DeviceGraph(steps, pb, pe, n_segs))
    text = plan.describe()
    assert f"x8192 ranges={ranges} " in text and "pass2=tagged" in text and "bucket_cap=2048" in text, text
    d = torch.zeros(n_segs, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(n_segs, dtype=torch.int32, device="cuda:0")
    for k in range(2):
        plan.seg_depth(d, u if k == 0 else None)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()
        if k == 0:
            assert (u.cpu().numpy().view(np.uint32) == want_u).all()
    plan.close()


@pytest.mark.parametrize("mode", ["kernel_bound", "plan_headroom"])
def test_a_workgroup_never_takes_more_items_than_it_has_tags(mode, monkeypatch):
    """A tagged k_scan deals its items to the workgroups as they get to them, and an item's private
    tag is its ordinal in its workgroup: a few long whole paths keep some workgroups busy while the
    others take the many tiny ones -- more of them than the mean a plan used to be accepted on.
    FLATGFA_TAG_LIMIT shrinks the tag space so that a small graph gets there.  `kernel_bound`: the plan
    is accepted on the mean alone (FLATGFA_TAG_MEAN_ONLY) and k_scan's workgroups stop taking items
    at the limit; `plan_headroom`: the plan plays the deal through and walks the paths in groups (or
    keeps the directory) instead.  Either way the counts are exact."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")       # every path is an item of k_scan
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for v in ("FLATGFA_TAGGED", "FLATGFA_PATH_GROUPS", "FLATGFA_PIECE_STEPS"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("FLATGFA_PIECE_STEPS", "100000000")  # the long paths stay whole
    monkeypatch.setenv("FLATGFA_TAG_LIMIT", "8")
    if mode == "kernel_bound":
        monkeypatch.setenv("FLATGFA_TAG_MEAN_ONLY", "1")
    else:
        monkeypatch.delenv("FLATGFA_TAG_MEAN_ONLY", raising=False)
    S = 150_000
    g = pa.synth(31, S, 10, 200_000, "pangenome", False)
    steps, _, _, _ = g.soa()
    # 200 long paths' worth of workgroups would be too many for a quick test: six long paths of 200 k
    # steps, then 1600 paths of 500 steps (256 workgroups: 6.3 items each on average, limit 8)
    lens = np.array([200_000] * 6 + [500] * 1600, dtype=np.uint32)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - lens).astype(np.uint32)
    paths = np.zeros(len(lens), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, steps, np.zeros(S, dtype=fo.SEG_DT)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
    text = plan.describe()
    if mode == "kernel_bound":
        assert "pass2=tagged" in text and "path_groups" not in text, text
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for _ in range(4):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all(), text
        assert (u.cpu().numpy().view(np.uint32) == want_u).all(), text
    plan.close()


@pytest.mark.parametrize("n_paths,max_len", [(5000, 128), (20000, 40), (700, 64), (300, 129)])
def test_tiny_paths_are_held_whole_by_a_wave(n_paths, max_len, monkeypatch):
    """k_scan_tiny: paths of at most 128 steps, a wave holding one whole (two steps per lane), first visits
    through a per-wave set of ids, records cut where the first-visit flag changes and at window
    boundaries.  Paths of every length from 1 up, walks that revisit their own segments, run down the ids,
    sit on a window boundary (4095 | 4096) and on the last segment; mixed with a few longer paths.
    FLATGFA_NO_TINY sends the same graph through k_scan_short: both must give the oracle's counts."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for v in ("FLATGFA_TAGGED", "FLATGFA_SHORT_MAX", "FLATGFA_PIECE_STEPS", "FLATGFA_NO_TINY"):
        monkeypatch.delenv(v, raising=False)
    rng = np.random.default_rng(n_paths + max_len)
    S = 3 * 4096 + 77
    walks = []
    for p in range(n_paths):
        n = int(rng.integers(1, max_len + 1))
        start = int(rng.choice([rng.integers(0, S), 4090, 8190, S - 3, 0]))
        jumps = rng.choice([1, 1, 1, 1, 1, 2, -1, -1, -3, 0, 5, 4096, -4000], size=n)
        walks.append(((start + np.cumsum(jumps)) % S).astype(np.uint32))
    for n in (3000, 9000):  # ... and two that are not tiny
        walks.append(((int(rng.integers(0, S)) + np.cumsum(rng.choice([1, 1, 1, 2, -1], size=n))) % S).astype(np.uint32))
    lens = np.array([len(x) for x in walks], dtype=np.uint32)
    stp = (np.concatenate(walks) << 1) | rng.integers(0, 2, size=int(lens.sum())).astype(np.uint32)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - lens).astype(np.uint32)
    paths = np.zeros(len(lens), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, stp, np.zeros(S, dtype=fo.SEG_DT)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    for no_tiny in (False, True):
        if no_tiny:
            monkeypatch.setenv("FLATGFA_NO_TINY", "1")
        plan = DepthPlan(DeviceGraph(stp, pb, pe, S))
        text = plan.describe()
        assert ("k_scan_tiny" in text) == (not no_tiny), text
        d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        for k in range(3):
            plan.seg_depth(d, u if k != 1 else None)
            plan.status()
            assert (d.cpu().numpy().view(np.uint32) == want_d).all(), text
            if k != 1:
                assert (u.cpu().numpy().view(np.uint32) == want_u).all(), text
        plan.close()


def test_packed_buckets_and_steps_that_change_behind_the_plan(monkeypatch):
    """Packed record buckets have exactly the room the counted call needed.  The same steps: every call
    fits (and the plan says what it took).  Other steps behind the plan's back -- runs broken up, so
    that more records are made than were counted: the records that do not fit go to the sink, the
    call is flagged, and flatgfa_dev_status completes it through the atomic kernels: the counts are
    those of the steps as they are now."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_PACKED", "1")
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    for v in ("FLATGFA_TAGGED", "FLATGFA_PIECE_STEPS", "FLATGFA_BUCKET_CAP"):
        monkeypatch.delenv(v, raising=False)
    S = 1_300_000
    g = pa.synth(41, S, 300, 20_000, "chromosome", False)
    steps, pb, pe, _ = g.soa()
    pools = pools_of(g)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    graph = DeviceGraph(steps, pb, pe, S)
    plan = DepthPlan(graph)
    text = plan.describe()
    assert "buckets=packed" in text and "pass2=tagged" in text, text
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for k in range(3):
        plan.seg_depth(d, u if k != 1 else None)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()
        if k != 1:
            assert (u.cpu().numpy().view(np.uint32) == want_u).all()
    # every other step of the first hundred paths now jumps somewhere else: many more runs than were counted
    rng = np.random.default_rng(5)
    changed = steps.copy()
    idx = np.arange(0, 100 * 20_000, 2)
    changed[idx] = (rng.integers(0, S, size=len(idx)).astype(np.uint32) << 1)
    graph.steps.copy_(torch.from_numpy(changed.view(np.int32)))
    torch.cuda.synchronize()
    pools.steps = changed
    want_d2, want_u2 = fo.seg_depth_with_uniq(pools)
    plan.seg_depth(d, u)
    plan.status()   # completes the call
    assert (d.cpu().numpy().view(np.uint32) == want_d2).all() and (u.cpu().numpy().view(np.uint32) == want_u2).all()
    plan.seg_depth(d, u)  # and later calls take the atomic kernels right away
    plan.status()
    assert (d.cpu().numpy().view(np.uint32) == want_d2).all() and (u.cpu().numpy().view(np.uint32) == want_u2).all()
    plan.close()


@pytest.mark.parametrize("cont,want", [(0.0, "even"), (0.05, "packed"), (0.5, "packed")])
def test_packed_buckets_and_blocks_of_nearly_all_starts(cont, want, monkeypatch):
    """A packed call's run queues are 88 entries shorter than a block can have starts (two LDS tables for up to
    4096 windows).  A block whose starts do not fit behind what is queued is appended after a drain down
    to one entry (ids that continue a run five times in a hundred: 970 starts per block); blocks without any
    run at all (more than 1005 starts) are flagged by the counting call, and the plan keeps the even layout."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_PACKED", "1")
    monkeypatch.setenv("FLATGFA_DENSE", "0")      # (pass 1 by runs whatever the run count: the plan would otherwise partition)
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    for v in ("FLATGFA_TAGGED", "FLATGFA_PIECE_STEPS", "FLATGFA_BUCKET_CAP"):
        monkeypatch.delenv(v, raising=False)
    rng = np.random.default_rng(int(cont * 100) + 3)
    S, P, L = 1_200_000, 40, 50_000
    ids = rng.integers(0, S, size=P * L).astype(np.int64)
    keep = rng.random(P * L) < cont
    keep[::L] = False
    run = np.arange(P * L)
    last_start = np.maximum.accumulate(np.where(~keep, run, 0))
    ids = (ids[last_start] + (run - last_start)) % S   # a step that continues follows the one before by +1
    stp = (ids.astype(np.uint32) << 1) | rng.integers(0, 2, size=P * L).astype(np.uint32)
    pe = (np.arange(1, P + 1) * L).astype(np.uint32)
    pb = (pe - L).astype(np.uint32)
    paths = np.zeros(P, dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, stp, np.zeros(S, dtype=fo.SEG_DT)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    plan = DepthPlan(DeviceGraph(stp, pb, pe, S))
    text = plan.describe()
    assert f"buckets={want}" in text, text
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for k in range(3):
        plan.seg_depth(d, u if k != 1 else None)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all(), text
        if k != 1:
            assert (u.cpu().numpy().view(np.uint32) == want_u).all(), text
    plan.close()


def test_steps_kept_in_the_infinity_cache_are_budgeted_per_device(monkeypatch):
    """k_scan reads the first cache_resident_mb of a plan's steps without the nt hint (they stay in the 256 MiB
    Infinity Cache between calls).  The amount is what the call's other traffic leaves of the cache and what the
    device's other plans have not claimed; a destroyed plan gives its share back; FLATGFA_MALL_MB pins it.
    Whatever the amount, the counts are the same (it is a cache policy of the step loads, nothing else)."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for v in ("FLATGFA_MALL_MB", "FLATGFA_TAGGED", "FLATGFA_SHORT_MAX", "FLATGFA_PIECE_STEPS"):
        monkeypatch.delenv(v, raising=False)
    S = 1_000_000
    g = pa.synth(51, S, 250, 100_000, "pangenome", False)  # 25 M steps = 100 MB
    steps, pb, pe, _ = g.soa()
    want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
    graph = DeviceGraph(steps, pb, pe, S)

    def resident(plan):
        return int(re.search(r"cache_resident_mb=(\d+)", plan.describe()).group(1))

    def check(plan):
        d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        for _ in range(2):
            plan.seg_depth(d, u)
            plan.status()
            assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()

    a = DepthPlan(graph)
    ra = resident(a)
    assert 90 <= ra <= 100, a.describe()       # all of its steps: they fit what a call's traffic leaves
    b = DepthPlan(graph)                       # a second plan over the SAME step array (two calls in flight): the same stretch, one claim
    assert resident(b) == ra, b.describe()
    graph2 = DeviceGraph(steps, pb, pe, S)     # another resident array: what the first left of the device's budget (nothing below 32 MB)
    c = DepthPlan(graph2)
    rc = resident(c)
    assert rc <= 160 - ra and rc in (0, *range(32, 161)), c.describe()
    check(a)
    check(b)
    check(c)
    a.close()
    d2 = DepthPlan(graph)                      # b still holds the shared claim
    assert resident(d2) == ra, d2.describe()
    d2.close()
    b.close()
    c.close()
    e = DepthPlan(graph2)
    assert resident(e) >= 90, e.describe()     # every share is free again
    check(e)
    e.close()
    monkeypatch.setenv("FLATGFA_MALL_MB", "0")
    z = DepthPlan(graph)
    assert resident(z) == 0
    check(z)
    z.close()
    monkeypatch.setenv("FLATGFA_MALL_MB", "37")
    p = DepthPlan(graph)
    assert resident(p) == 37
    check(p)
    p.close()


def test_plan_says_how_many_steps_each_scan_kernel_walks(monkeypatch):
    """`steps=a/b/c/d` of a plan's description: the steps walked by k_scan / k_scan_short / k_scan_medium / k_scan_tiny
    (bench.py prices a kernel's launches against the bytes of its own paths).  Every step is in exactly one class."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for v in ("FLATGFA_SHORT_MAX", "FLATGFA_NO_TINY", "FLATGFA_TAGGED", "FLATGFA_WB", "FLATGFA_PACKED"):
        monkeypatch.delenv(v, raising=False)
    S = 300_000
    rng = np.random.default_rng(11)
    lens = np.concatenate([rng.integers(1, 100, 500), rng.integers(200, 2000, 300), rng.integers(3000, 9000, 60), rng.integers(40_000, 90_000, 12)])
    rng.shuffle(lens)
    starts = rng.integers(0, S - 100_000, len(lens))
    ids = np.concatenate([np.arange(s, s + n) for s, n in zip(starts, lens)]).astype(np.uint32)
    steps = (ids << 1).astype(np.uint32)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - lens).astype(np.uint32)
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
    text = plan.describe()
    m = re.search(r" steps=(\d+)/(\d+)/(\d+)/(\d+)", text)
    assert m, text
    by = [int(x) for x in m.groups()]
    assert sum(by) == len(steps), text
    assert by[3] == int(lens[lens <= 128].sum()) and by[0] >= int(lens[lens >= 40_000].sum()), text
    assert all(by), text  # (runs of consecutive ids: the paths of 200 .. 9000 steps have few runs, so single waves walk them)
    import torch
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    plan.seg_depth(d, u)
    plan.status()
    want = np.bincount(ids, minlength=S)
    assert (d.cpu().numpy() == want).all()
    seen = np.zeros(S, dtype=np.int64)
    for s, n in zip(starts, lens):
        seen[s:s + n] += 1
    assert (u.cpu().numpy() == seen).all()


def test_medium_paths_whose_blocks_hold_more_runs_than_the_queue(monkeypatch):
    """Paths of 2600 steps in runs of two (1300 runs: wave-per-path "medium" paths): a block of 1024 steps starts 512
    runs, more than a wave's queue has room for, so k_scan_medium queues it sixteen lanes at a time (entries must lie
    in the order of their positions: a run ends where the next entry starts).  Every other path walks downwards
    (read from its reversed copy), and some revisit their own segments."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for v in ("FLATGFA_SHORT_MAX", "FLATGFA_NO_TINY", "FLATGFA_TAGGED", "FLATGFA_WB", "FLATGFA_PACKED", "FLATGFA_DENSE"):
        monkeypatch.delenv(v, raising=False)
    S = 200_000
    rng = np.random.default_rng(23)
    paths = []
    for p in range(300):
        starts = rng.integers(0, S - 2, 1300)
        if p % 5 == 0:
            starts[650:] = starts[:650]  # the second half walks the first half's segments again
        ids = np.stack([starts, starts + 1], axis=1).reshape(-1)
        if p % 2:
            ids = ids[::-1]
        paths.append(ids.astype(np.uint32))
    lens = np.array([len(x) for x in paths])
    ids = np.concatenate(paths)
    steps = ((ids << 1) | (rng.integers(0, 2, len(ids)).astype(np.uint32))).astype(np.uint32)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - lens).astype(np.uint32)
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
    text = plan.describe()
    assert int(re.search(r"medium_paths=(\d+)", text).group(1)) == 300, text
    want_d = np.bincount(ids, minlength=S)
    want_u = np.zeros(S, dtype=np.int64)
    for x in paths:
        want_u[np.unique(x)] += 1
    for with_u in (True, False):
        d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        plan.seg_depth(d, u if with_u else None)
        plan.status()
        assert (d.cpu().numpy() == want_d).all()
        if with_u:
            assert (u.cpu().numpy() == want_u).all()


@pytest.mark.parametrize("shape", [(300_000, 40, 30_000, "uniform"), (200_000, 300, 5_000, "pangenome"), (123_457, 7, 200_001, "uniform")])
def test_dense_pass1_on_segment_ranges(shape, monkeypatch):
    """k_scan_dense on a plan cut into segment ranges: a walk per range leaves out the steps outside it, so every phase takes
    its general, predicated form (full tiles of plans without ranges take the plain one), tiles pipelined two deep all the same."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DENSE", "1")
    monkeypatch.setenv("FLATGFA_RANGE_SEGS", "40960")
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_BIG_GROUPS", "1")
    for v in ("FLATGFA_SHORT_MAX", "FLATGFA_TAGGED", "FLATGFA_WB", "FLATGFA_PACKED"):
        monkeypatch.delenv(v, raising=False)
    S, P, L, model = shape
    g = pa.synth(3, S, P, L, model, False)
    steps, pb, pe, sl = g.soa()
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S, sl))
    assert "k_scan_dense" in plan.describe() and "ranges=1 " not in plan.describe(), plan.describe()
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    plan.seg_depth(d, u)
    plan.status()
    want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
    assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()


def _monotone_graph(S, kinds, seed=0):
    """Paths given as lists of segment ids per kind: 'up' a strictly increasing subset, 'down' strictly decreasing,
    'up2' increasing in steps of two (no run longer than one step), 'loop' increasing with one stretch walked twice,
    'flat' increasing but for one repeated id, 'wrap' increasing past the last segment and on from the first."""
    rng = np.random.default_rng(seed)
    walks = []
    for k, (kind, n) in enumerate(kinds):
        lo = int(rng.integers(0, max(1, S // 4)))
        ids = lo + np.nonzero(rng.random(S - lo) < 0.6)[0][:n]
        if kind == "down":
            ids = ids[::-1]
        elif kind == "up2":
            ids = (lo + 2 * np.arange(min(n, (S - lo) // 2)))
        elif kind == "loop":
            ids = np.concatenate([ids, ids[len(ids) // 3: len(ids) // 3 + 50], ids[-1:] ])
        elif kind == "flat":
            ids = np.concatenate([ids[: len(ids) // 2], ids[len(ids) // 2 - 1:]])
        elif kind == "wrap":
            ids = np.concatenate([ids[len(ids) // 2:], ids[: len(ids) // 2]])
        walks.append(ids.astype(np.uint32))
    lens = np.array([len(x) for x in walks], dtype=np.uint32)
    steps = (np.concatenate(walks) << 1) | rng.integers(0, 2, size=int(lens.sum())).astype(np.uint32)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - lens).astype(np.uint32)
    paths = np.zeros(len(lens), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, steps, np.zeros(S, dtype=fo.SEG_DT)
    return steps, pb, pe, pools


@pytest.mark.parametrize("pieces", [0, 4096])
@pytest.mark.parametrize("n_segs", [200_000, 5_000_000])
def test_strictly_monotone_paths_skip_the_claim(n_segs, pieces, monkeypatch):
    """A path that walks the segment ids strictly one way never meets a segment twice: depth.rs:30-34's `seen` test is
    always true for it, its records carry the no-claim tag and pass 2 applies them without a bitset.  The plan must find
    exactly the paths that qualify -- upwards, downwards, in steps of two, cut into pieces -- and none that revisits
    (a loop, one repeated id, a wrap around the last segment), alone or mixed in one workgroup's sub-buckets."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    monkeypatch.delenv("FLATGFA_NO_CLAIM", raising=False)
    if pieces:
        monkeypatch.setenv("FLATGFA_PIECE_STEPS", str(pieces))
    else:
        monkeypatch.delenv("FLATGFA_PIECE_STEPS", raising=False)
    kinds = [("up", 30_000), ("down", 30_000), ("loop", 20_000), ("up2", 9000), ("flat", 10_000), ("wrap", 12_000), ("up", 64), ("down", 1),
             ("up", 50_000), ("loop", 300), ("down", 17_000)] * 3
    steps, pb, pe, pools = _monotone_graph(n_segs, kinds, seed=n_segs + pieces)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    plan = DepthPlan(DeviceGraph(steps, pb, pe, n_segs))
    desc = plan.describe()
    assert "pass2=tagged" in desc, desc
    n_items = int(re.search(r" items=(\d+)", desc).group(1))
    n_noclaim = int(re.search(r"no_claim_items=(\d+)", desc).group(1))
    mono = sum(1 for k, _ in kinds if k in ("up", "down", "up2"))
    assert 0 < n_noclaim < n_items, desc
    if n_items == len(kinds):  # (no path was cut: exactly the paths that qualify)
        assert n_noclaim == mono, desc
    d = torch.zeros(n_segs, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(n_segs, dtype=torch.int32, device="cuda:0")
    for _ in range(2):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()
        assert (u.cpu().numpy().view(np.uint32) == want_u).all()
    # all of them monotone: nothing but depth updates in pass 2, and uniq == depth where each segment is met by distinct paths
    steps2, pb2, pe2, pools2 = _monotone_graph(n_segs, [("up", 40_000), ("down", 40_000), ("up2", 20_000)] * 8, seed=7)
    want_d2, want_u2 = fo.seg_depth_with_uniq(pools2)
    plan2 = DepthPlan(DeviceGraph(steps2, pb2, pe2, n_segs))
    m = re.search(r" items=(\d+) no_claim_items=(\d+)", plan2.describe())
    assert m and m.group(1) == m.group(2), plan2.describe()
    plan2.seg_depth(d, u)
    plan2.status()
    assert (d.cpu().numpy().view(np.uint32) == want_d2).all() and (u.cpu().numpy().view(np.uint32) == want_u2).all()
    assert (want_d2 == want_u2).all()


def _walks_graph(S, walks, seed=0):
    rng = np.random.default_rng(seed)
    lens = np.array([len(x) for x in walks], dtype=np.uint32)
    steps = (np.concatenate(walks).astype(np.uint32) << 1) | rng.integers(0, 2, size=int(lens.sum())).astype(np.uint32)
    pe = np.cumsum(lens).astype(np.uint32)
    pb = (pe - lens).astype(np.uint32)
    paths = np.zeros(len(lens), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, steps, np.zeros(S, dtype=fo.SEG_DT)
    return steps, pb, pe, pools


@pytest.mark.parametrize("no_claim", ["", "0"])
def test_strictly_monotone_short_paths_skip_the_claim(no_claim, monkeypatch):
    """The same for the paths single waves walk (k_scan_tiny, k_scan_short, its medium build): the plan lists the
    paths that walk the ids strictly one way in a stretch of their own, and the kernels neither probe nor wipe the
    per-path set for them.  Tiny, short and medium paths, upwards and downwards (the latter read from reversed
    copies), next to paths of the same lengths that do meet a segment twice -- a stretch walked again, one id
    repeated in place -- whose first visits must still be found; FLATGFA_NO_CLAIM=0 lists none and counts the same."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for v in ("FLATGFA_SHORT_MAX", "FLATGFA_NO_TINY", "FLATGFA_SHORT_ANY"):
        monkeypatch.delenv(v, raising=False)
    if no_claim:
        monkeypatch.setenv("FLATGFA_NO_CLAIM", no_claim)
    else:
        monkeypatch.delenv("FLATGFA_NO_CLAIM", raising=False)
    S = 600_000
    rng = np.random.default_rng(5)
    shapes = [("up", 100, 0.6), ("down", 128, 0.6), ("up2", 60, 1.0), ("loop", 90, 0.6), ("flat", 70, 0.6), ("up", 1, 1.0), ("up", 2, 0.5),
              ("up", 1500, 0.6), ("down", 2000, 0.7), ("loop", 1200, 0.6), ("flat", 900, 0.6), ("up2", 700, 1.0), ("down", 129, 0.9),
              ("up", 8000, 0.93), ("down", 9000, 0.95), ("loop", 7000, 0.93), ("flat", 6000, 0.95), ("up", 2049, 0.9)]
    walks, n_mono = [], 0
    for rep in range(24):
        for kind, n, dens in shapes:
            lo = int(rng.integers(0, S - 3 * n - 64))
            ids = lo + (np.nonzero(rng.random(int(1.2 * n / dens) + 64) < dens)[0][:n] if kind != "up2" else 2 * np.arange(n))
            assert len(ids) == n
            if kind == "down":
                ids = ids[::-1]
            elif kind == "loop":
                ids = np.concatenate([ids, ids[n // 3: n // 3 + 20]])
            elif kind == "flat":
                ids = np.concatenate([ids[: n // 2], ids[n // 2 - 1:]])
            n_mono += kind in ("up", "down", "up2")
            walks.append(ids)
    steps, pb, pe, pools = _walks_graph(S, walks, seed=11)
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
    desc = plan.describe()
    assert "k_scan_tiny" in desc and "k_scan_short" in desc and "k_scan_medium" in desc, desc
    got = int(re.search(r"no_claim_paths=(\d+)", desc).group(1))
    if no_claim:
        assert got == 0, desc
    elif re.search(r" items=0 ", desc):  # (every path went to a wave-per-path kernel: exactly the ones that qualify)
        assert got == n_mono, desc
    else:
        assert 0 < got <= n_mono, desc
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for _ in range(2):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all()
        assert (u.cpu().numpy().view(np.uint32) == want_u).all()
    plan.seg_depth(d, None)
    plan.status()
    assert (d.cpu().numpy().view(np.uint32) == want_d).all()


@pytest.mark.parametrize("shape", [(300_000, 40, 60_000), (2_000_000, 9, 500_000), (5_000_000, 600, 30_000)])
def test_stretches_that_need_no_claim_in_paths_that_are_not_monotone(shape, monkeypatch):
    """A path that goes back over a few segments now and then (the `repeats` model: a tandem duplication every 6400 steps) is not
    monotone as a whole, but most windows it enters once and walks one way: the plan marks the 16-step chunks that lie in such
    windows only (k_visit_bits, k_chunk_flags), and k_scan's blocks made of them carry the no-claim tag.  Whole paths, paths cut
    into pieces, 4096- and 8192-segment windows; the counts are the oracle's with the marks and without (FLATGFA_NO_CLAIM=0)."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    S, P, L = shape
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    g = pa.synth(21, S, P, L, "repeats", False)
    steps, pb, pe, _ = g.soa()
    wd, wu = fo.seg_depth_with_uniq(pools_of(g))
    assert (wd != wu).any()  # (the walks do meet segments twice)
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for nc in ("", "0"):
        if nc:
            monkeypatch.setenv("FLATGFA_NO_CLAIM", nc)
        else:
            monkeypatch.delenv("FLATGFA_NO_CLAIM", raising=False)
        plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
        desc = plan.describe()
        assert "pass2=tagged" in desc, desc
        m = re.search(r"no_claim_chunks=(\d+)", desc)
        if nc:
            assert m is None, desc
        else:
            assert m and int(m.group(1)) > P * L // 16 // 2, desc  # (more than half of all chunks: what it takes for a plan to use them)
        for _ in range(2):
            plan.seg_depth(d, u)
            plan.status()
            assert (d.cpu().numpy().view(np.uint32) == wd).all(), desc
            assert (u.cpu().numpy().view(np.uint32) == wu).all(), desc


def test_no_claim_marks_where_spans_overlap(monkeypatch):
    """The per-block no-claim marks are bits of the step array, and spans may overlap (pool.rs:80-124 allows it): a chunk two paths
    walk must qualify for both.  Path A walks a stretch once; path B walks the same steps and then, in a copy of them, half of
    them again -- B meets those windows twice, A does not, and neither may go without claims there."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    monkeypatch.delenv("FLATGFA_NO_CLAIM", raising=False)
    monkeypatch.setenv("FLATGFA_NO_CLAIM_BLOCKS_MIN", "0")  # (by default a plan uses the marks when half of its blocks' chunks have one)
    S, L = 1_000_000, 160_000
    g = pa.synth(5, S, 6, L, "repeats", False)
    st, _, _, _ = g.soa()
    walks = [st[i * L:(i + 1) * L] for i in range(6)]
    steps = np.concatenate([walks[0], walks[0], walks[1], walks[2], walks[2], walks[3], walks[4], walks[5]]).astype(np.uint32)
    pb = np.array([0, 0, 2 * L, 3 * L, 3 * L + L // 3, 5 * L, 6 * L, 7 * L, 5 * L + 77], dtype=np.uint32)
    pe = np.array([L, L + L // 2, 3 * L, 4 * L, 5 * L, 6 * L, 7 * L, 8 * L, 8 * L - 5], dtype=np.uint32)
    paths = np.zeros(len(pb), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, steps, np.zeros(S, dtype=fo.SEG_DT)
    wd, wu = fo.seg_depth_with_uniq(pools)
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
    desc = plan.describe()
    assert re.search(r"no_claim_chunks=\d+", desc), desc
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for _ in range(2):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == wd).all(), desc
        assert (u.cpu().numpy().view(np.uint32) == wu).all(), desc


def test_small_windows_where_workgroups_take_dozens_of_items(monkeypatch):
    """A graph beyond 4 M segments gets 8192-segment windows -- unless a pass-1 workgroup takes so many claiming items
    that pass 2 is better off with eight private bitsets per wave, which only 4096-segment windows leave LDS for
    (fast_plan_create; NOTES R5.12).  The plan's description says which it took; the counts are the oracle's either way."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_SCAN_WGS", "64")
    for v in ("FLATGFA_WB", "FLATGFA_RANGE_SEGS", "FLATGFA_ACC_SLOTS"):
        monkeypatch.delenv(v, raising=False)
    S = 4_200_000
    for P, L, want in [(1500, 40_000, "x4096"), (640, 80_000, "x8192")]:
        g = pa.synth(9, S, P, L, "chromosome", False)
        steps, pb, pe, _ = g.soa()
        wd, wu = fo.seg_depth_with_uniq(pools_of(g))
        plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
        desc = plan.describe()
        assert "pass2=tagged" in desc and want in desc, desc
        d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == wd).all() and (u.cpu().numpy().view(np.uint32) == wu).all(), desc


def test_wrong_answer_switches_are_not_in_the_product_library(monkeypatch):
    """FLATGFA_DEBUG_SKIP / FLATGFA_ACC_SKIP leave parts of the kernels' work out (measurements; results are then wrong
    by construction): they exist in measurement builds only (-DFGFA_MEASURE, tools/variants.sh).  The product
    library must not read them."""
    monkeypatch.setenv("FLATGFA_DEBUG_SKIP", "1")
    monkeypatch.setenv("FLATGFA_ACC_SKIP", "448")
    monkeypatch.setenv("FLATGFA_ACC_PAIR", "1")
    monkeypatch.setenv("FLATGFA_ACC_SMALL", "1")
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for shape in [(3, 300_000, 60, 50_000, "pangenome"), (4, 50_000, 2000, 900, "chromosome")]:
        g = pa.synth(*shape, False)
        check_graph(g, pools_of(g))
    with open(os.path.join(ROOT, "pollen_amd", "lib", "libflatgfa.so"), "rb") as f:
        blob = f.read()
    for name in (b"FLATGFA_DEBUG_SKIP", b"FLATGFA_ACC_SKIP", b"FLATGFA_ACC_PAIR", b"FLATGFA_ACC_SMALL", b"k_accum_pair", b"k_accum_small"):
        assert name not in blob, name


def test_more_than_two_to_the_31_steps():
    """The format allows 2^32 - 1 steps (u32 spans, pool.rs:80-86).  Three thousand million steps on a million segments
    (12 GB of handles): every step index in the upper half of the u32 range, bit for bit against the oracle -- node depth,
    unique depth, and the sums of path depth for a few paths that lie beyond step 2^31."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    S, P, L = 1_000_000, 30_000, 100_000
    assert 2**31 < P * L < 2**32
    g = pa.synth(77, S, P, L, "pangenome", False)
    pools = pools_of(g)
    g.close()  # (one copy of the 12 GB on the host is enough)
    steps = pools.steps
    pb = np.ascontiguousarray(pools.paths["steps_start"], dtype=np.uint32)
    pe = np.ascontiguousarray(pools.paths["steps_end"], dtype=np.uint32)
    seg_len = pools.seg_lens()
    assert int(pe[-1]) == P * L and int(pb[-1]) > 2**31
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    assert int(want_d.sum()) == P * L
    graph = DeviceGraph(steps, pb, pe, S, seg_len)
    plan = DepthPlan(graph)
    assert "path=bucketed" in plan.describe(), plan.describe()
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for _ in range(2):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all(), "depth"
        assert (u.cpu().numpy().view(np.uint32) == want_u).all(), "uniq"
    ids = np.array([P - 1, P - 2, 21_475, 21_474, 0], dtype=np.uint32)  # (path 21 475 starts just beyond step 2^31)
    want_len, want_mean = fo.path_depth(pools, ids)
    ln = torch.zeros(len(ids), dtype=torch.int64, device="cuda:0")
    ws = torch.zeros(len(ids), dtype=torch.int64, device="cuda:0")
    plan.path_sums(torch.from_numpy(ids.view(np.int32)).to("cuda:0"), d, ln, ws)
    plan.status()
    got_len = ln.cpu().numpy().view(np.uint64)
    got_mean = ws.cpu().numpy().view(np.uint64).astype(np.float64) / got_len.astype(np.float64)
    assert (got_len == want_len).all() and got_mean.tobytes() == want_mean.tobytes()
    plan.close()


def test_more_steps_than_a_u32_can_index_is_an_error(tmp_path):
    """One step more than the format's u32 spans can index (pool.rs:80-86): FLATGFA_ERR_TOO_LARGE when the graph is
    made resident, not a truncated walk.  The file is sparse: a table of contents that announces 2^32 steps, and a hole."""
    from oracle.flatgfa_oracle import MAGIC, POOL_DTYPES, POOL_ORDER, SEG_DT
    n_steps = 2**32
    seg = np.zeros(1, dtype=SEG_DT)
    lens = {n: 0 for n in POOL_ORDER}
    lens["segs"], lens["steps"] = 1, n_steps
    toc = np.uint64(MAGIC).tobytes() + b"".join(np.array([lens[n], lens[n]], dtype="<u8").tobytes() for n in POOL_ORDER)
    path = tmp_path / "too_large.flatgfa"
    with open(path, "wb") as f:
        f.write(toc + seg.tobytes())
        f.truncate(len(toc) + sum(lens[n] * POOL_DTYPES[n].itemsize for n in POOL_ORDER))
    assert os.stat(path).st_blocks * 512 < (1 << 24), "the file system does not keep the file sparse"
    g = pa.load(str(path))
    assert g.segment_count == 1 and g.path_count == 0
    with pytest.raises(pa.FlatGFAError) as ei:
        g.to_device(0)
    assert ei.value.code == -6  # FLATGFA_ERR_TOO_LARGE
    with pytest.raises(pa.FlatGFAError) as ei:
        g.seg_depth_with_uniq()
    assert ei.value.code == -6
    g.close()


@pytest.mark.parametrize("calls_in_flight", [1, 2, 3])
def test_calls_in_flight_through_the_pipeline(calls_in_flight, monkeypatch):
    """flatgfa_dev_pipeline_*: K plans of one resident graph on K internal streams, taken in turn (pass 2 of one call beside
    pass 1 of the next).  Every call is a whole query: each lane's result, of many calls enqueued without waiting, is the
    oracle's; join() orders the caller's stream behind them; an out-of-range step is an error from status(), not a count."""
    from pollen_amd.device import DepthPipeline, DeviceGraph
    import torch
    for v in ("FLATGFA_DEPTH_PATH", "FLATGFA_TAGGED", "FLATGFA_SHORT_MAX", "FLATGFA_PIECE_STEPS"):
        monkeypatch.delenv(v, raising=False)
    S = 700_000
    g = pa.synth(61, S, 400, 30_000, "chromosome", False)
    steps, pb, pe, _ = g.soa()
    want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
    graph = DeviceGraph(steps, pb, pe, S)
    pipe = DepthPipeline(graph, calls_in_flight)
    assert f"calls_in_flight={calls_in_flight} " in pipe.describe() and "path=bucketed" in pipe.describe(), pipe.describe()
    bufs = [torch.full((2 * S,), -3, dtype=torch.int32, device="cuda:0") for _ in range(calls_in_flight)]
    for n in range(5 * calls_in_flight + 1):
        b = bufs[n % calls_in_flight]
        pipe.seg_depth(b[:S], b[S:], after_current_stream=(n % 2 == 0))
    pipe.join()  # torch's current stream now waits for all of them: the copies below are ordered behind the kernels
    got = [b.cpu().numpy().view(np.uint32) for b in bufs]
    pipe.status()
    for k, x in enumerate(got):
        assert (x[:S] == want_d).all() and (x[S:] == want_u).all(), f"lane {k}"
    # depth only, and the buffers reused
    for n in range(calls_in_flight):
        bufs[n][:S].fill_(7)
        pipe.seg_depth(bufs[n][:S], None)
    pipe.status()
    for b in bufs:
        assert (b[:S].cpu().numpy().view(np.uint32) == want_d).all()
    # path depth of all paths (what `fgfa depth` prints) through the lanes
    _, _, _, seg_len = g.soa()
    graph2 = DeviceGraph(steps, pb, pe, S, seg_len)
    pipe2 = DepthPipeline(graph2, calls_in_flight)
    P = len(pb)
    outs = [(torch.zeros(S, dtype=torch.int32, device="cuda:0"), torch.zeros(P, dtype=torch.int64, device="cuda:0"),
             torch.zeros(P, dtype=torch.int64, device="cuda:0")) for _ in range(calls_in_flight)]
    for n in range(3 * calls_in_flight):
        pipe2.path_depth_all(*outs[n % calls_in_flight])
    pipe2.status()
    want_len, want_mean = fo.path_depth(pools_of(g))
    for o in outs:
        assert (o[0].cpu().numpy().view(np.uint32) == want_d).all()
        got_len = o[1].cpu().numpy().view(np.uint64)
        got_mean = o[2].cpu().numpy().view(np.uint64).astype(np.float64) / got_len.astype(np.float64)
        assert (got_len == want_len).all() and got_mean.tobytes() == want_mean.tobytes()
    pipe2.close()
    # a step that names a segment beyond the graph: an error, on whichever lane meets it
    bad = steps.copy()
    bad[int(pb[3]) + 17] = np.uint32((S + 5) << 1)
    graph.steps.copy_(torch.from_numpy(bad.view(np.int32)))
    torch.cuda.synchronize()
    pipe.seg_depth(bufs[0][:S], bufs[0][S:])
    with pytest.raises(pa.FlatGFAError) as ei:
        pipe.status()
    assert ei.value.code == -2  # FLATGFA_ERR_BOUNDS
    pipe.close()


# ---- the randomised differential check (tools/fuzz_gpu.py) as part of the suite: a seeded slice of it ----
@pytest.mark.parametrize("part", range(4))
def test_fuzz_slice(part):
    """Twenty-five seeded cases each of tools/fuzz_gpu.py: random graphs (segment counts from 1 to 9 M, mixes of tiny, short,
    medium and long paths, monotone walks, spans with gaps and duplicates) through the forty-odd forced device
    configurations of its ENVS list in turn -- node depth, unique depth, depth only, path sums of a subset and path depth
    of all paths against the C oracle, bit for bit.  The four parts together walk cases 0..99 (every configuration twice)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tools", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    saved = {k: os.environ.get(k) for k in fz.ENV_KEYS}
    rng = np.random.default_rng(2026_06 + part)
    failed = []
    try:
        for case in range(25 * part, 25 * part + 25):
            ok, what = fz.run_case(rng, case, verbose=False)
            if not ok:
                failed.append(what)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert not failed, failed


# ---- step values changed behind a live plan (include/flatgfa.h: flatgfa_dev_plan_steps_changed, FLATGFA_CHECK_NO_CLAIM) ----
def _pools_of_arrays(steps, pb, pe, S):
    paths = np.zeros(len(pb), dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, np.ascontiguousarray(steps, dtype=np.uint32), np.zeros(S, dtype=fo.SEG_DT)
    return pools


def _graph_with_facts(S, seed):
    """Long monotone walks (no-claim items), short monotone ones up and down (wave-per-path lists, reversed copies), a few
    walks that revisit, and walks with a tandem repeat now and then (per-block marks)."""
    kinds = [("up", 60_000), ("down", 45_000), ("loop", 30_000), ("up", 1500), ("down", 1200), ("down", 900), ("up", 90), ("flat", 5000),
             ("up", 70_000), ("down", 1800)] * 3
    return _monotone_graph(S, kinds, seed=seed)


def _break_a_fact(steps, pb, pe, which):
    """One step changed so that a path the plan took for strictly monotone meets a segment twice."""
    s = steps.copy()
    lens = pe.astype(np.int64) - pb.astype(np.int64)
    if which == "long":       # the first path (60 000 steps, upwards): a step in its middle repeats the one five before it
        k = int(pb[0]) + int(lens[0]) // 2
    elif which == "short_up":  # a short one read forwards
        p = int(np.nonzero((lens > 1000) & (lens < 2000))[0][0])
        k = int(pb[p]) + int(lens[p]) // 3
    else:                      # a short one read from its reversed copy
        p = int(np.nonzero((lens > 1000) & (lens < 2000))[0][1])
        k = int(pb[p]) + int(lens[p]) // 3
    s[k] = s[k - 5]
    return s


@pytest.mark.parametrize("which", ["long", "short_up", "short_down"])
def test_steps_changed_remakes_the_plan(which, monkeypatch):
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.delenv("FLATGFA_CHECK_NO_CLAIM", raising=False)
    S = 300_000
    steps, pb, pe, pools = _graph_with_facts(S, seed=3)
    graph = DeviceGraph(steps, pb, pe, S)
    plan = DepthPlan(graph)
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    plan.seg_depth(d, u)
    plan.status()
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()
    import re
    desc = plan.describe()
    assert re.search(r"no_claim_items=[1-9]", desc) and re.search(r"no_claim_paths=[1-9]", desc), desc   # (both kinds of facts are in play)
    changed = _break_a_fact(steps, pb, pe, which)
    assert (changed != steps).sum() == 1
    graph.steps.copy_(torch.from_numpy(changed.view(np.int32)))   # behind the plan's back ...
    plan.steps_changed()                                          # ... and said so
    want_d2, want_u2 = fo.seg_depth_with_uniq(_pools_of_arrays(changed, pb, pe, S))
    assert (want_u2 != want_u).any()                              # (the change does move unique depth)
    for _ in range(2):
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d2).all()
        assert (u.cpu().numpy().view(np.uint32) == want_u2).all()
    # path depth rides on the same plan
    ln = torch.zeros(len(pb), dtype=torch.int64, device="cuda:0")
    ws = torch.zeros(len(pb), dtype=torch.int64, device="cuda:0")
    graph2 = DeviceGraph(changed, pb, pe, S, np.ones(S, dtype=np.uint32))
    plan2 = DepthPlan(graph2)
    plan2.path_depth_all(d, ln, ws)
    plan2.status()
    assert (d.cpu().numpy().view(np.uint32) == want_d2).all()


@pytest.mark.parametrize("which", ["long", "short_up", "short_down"])
def test_check_mode_finds_steps_changed_behind_the_plan(which, monkeypatch):
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_CHECK_NO_CLAIM", "1")
    S = 300_000
    steps, pb, pe, pools = _graph_with_facts(S, seed=5)
    graph = DeviceGraph(steps, pb, pe, S)
    plan = DepthPlan(graph)
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    for _ in range(2):                      # the checks pass on the steps the plan was made with
        plan.seg_depth(d, u)
        plan.status()
        assert (u.cpu().numpy().view(np.uint32) == want_u).all()
    changed = _break_a_fact(steps, pb, pe, which)
    graph.steps.copy_(torch.from_numpy(changed.view(np.int32)))
    plan.seg_depth(d, u)
    with pytest.raises(pa.FlatGFAError) as err:
        plan.status()
    assert err.value.code == -8, err.value    # FLATGFA_ERR_STALE_PLAN
    plan.steps_changed()
    want_d2, want_u2 = fo.seg_depth_with_uniq(_pools_of_arrays(changed, pb, pe, S))
    plan.seg_depth(d, u)
    plan.status()
    assert (d.cpu().numpy().view(np.uint32) == want_d2).all() and (u.cpu().numpy().view(np.uint32) == want_u2).all()


def test_check_mode_finds_a_stale_block_mark(monkeypatch):
    """The `repeats` model's walks go back over a few segments now and then: most blocks carry the no-claim mark.  A step changed
    inside a marked block so that its path enters a window twice takes the mark's ground away."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    monkeypatch.setenv("FLATGFA_CHECK_NO_CLAIM", "1")
    g = pa.synth(3, 600_000, 24, 100_000, "repeats", False)
    steps, pb, pe, _ = g.soa()
    S = 600_000
    graph = DeviceGraph(steps, pb, pe, S)
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    plan = DepthPlan(graph, first=(d, u))
    want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
    assert plan.first_status == 0
    assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()   # call 1: no marks yet
    desc = plan.describe()
    m = re.search(r"no_claim_chunks=(\d+)", desc)
    assert m and int(m.group(1)) > 0, desc
    plan.seg_depth(d, u)                                                                                              # call 2: with the marks
    plan.status()
    assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()
    # a step in the middle of path 3 takes the id of the step 40 000 before it: a second entry into that window
    changed = steps.copy()
    k = int(pb[3]) + 70_000
    changed[k] = changed[k - 40_000]
    graph.steps.copy_(torch.from_numpy(changed.view(np.int32)))
    plan.seg_depth(d, u)
    with pytest.raises(pa.FlatGFAError) as err:
        plan.status()
    assert err.value.code == -8
    plan.steps_changed()
    want_d2, want_u2 = fo.seg_depth_with_uniq(_pools_of_arrays(changed, pb, pe, S))
    plan.seg_depth(d, u)
    plan.status()
    assert (d.cpu().numpy().view(np.uint32) == want_d2).all() and (u.cpu().numpy().view(np.uint32) == want_u2).all()


# ---- the first answer: creating the plan is the first query (flatgfa_dev_plan_create_first) ----
@pytest.mark.parametrize("shape", [(1, 100_000, 100, 10_000, "pangenome"), (2, 1_000_000, 300, 40_000, "pangenome"), (3, 600_000, 24, 100_000, "repeats"),
                                   (4, 400_000, 2000, 900, "haplotype"), (5, 10_000, 100, 10_000, "pangenome"), (6, 5_000_000, 40, 200_000, "chromosome"),
                                   (7, 50_000, 30, 30_000, "uniform")], ids=lambda s: f"S{s[1]}_P{s[2]}_L{s[3]}_{s[4]}")
def test_first_answer_is_the_plans_sizing_query(shape, device_path):
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    seed, S, P, L, model = shape
    g = pa.synth(seed, S, P, L, model, False)
    steps, pb, pe, _ = g.soa()
    want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
    graph = DeviceGraph(steps, pb, pe, S)
    d = torch.full((S,), -3, dtype=torch.int32, device="cuda:0")     # (the outputs start as garbage)
    u = torch.full((S,), -5, dtype=torch.int32, device="cuda:0")
    plan = DepthPlan(graph, first=(d, u))
    assert plan.first_status == 0
    assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()
    d2 = torch.full((S,), -7, dtype=torch.int32, device="cuda:0")
    plan2 = DepthPlan(graph, first=(d2, None))                          # node depth alone
    assert (d2.cpu().numpy().view(np.uint32) == want_d).all()
    for p in (plan, plan2):                                             # ... and the plans are plans like any other
        d.fill_(-1)
        u.fill_(-1)
        p.seg_depth(d, u)
        p.status()
        assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()


def test_first_answer_reports_an_id_out_of_range():
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    S = 50_000
    g = pa.synth(9, S, 20, 20_000, "pangenome", False)
    steps, pb, pe, _ = g.soa()
    steps = steps.copy()
    steps[12345] = (S + 7) << 1
    graph = DeviceGraph(steps, pb, pe, S)
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    plan = DepthPlan(graph, first=(d, u))
    assert plan.first_status == -2                                      # FLATGFA_ERR_BOUNDS
    plan.seg_depth(d, u)
    with pytest.raises(pa.FlatGFAError):
        plan.status()


def test_wide_windows_on_graphs_of_a_few_million_segments(monkeypatch):
    """Two to four million segments are two to four 4096-segment windows per CU, and pass 2 pays a fixed part per window: a plan whose
    pass 2 needs nothing the wide windows have no LDS for (four bitsets per wave suffice, no split paths, no wave-per-path records) takes
    8192-segment windows there; one whose workgroups take more than four items each keeps 4096 (and its eight bitsets per wave)."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    for k in ("FLATGFA_WB", "FLATGFA_RANGE_SEGS", "FLATGFA_DEPTH_PATH", "FLATGFA_PACKED"):
        monkeypatch.delenv(k, raising=False)
    for S, P, L, model, want_w in ((3_000_000, 600, 50_000, "pangenome", 8192), (2_500_000, 2400, 20_000, "haplotype", 4096), (1_500_000, 500, 40_000, "pangenome", 4096)):
        g = pa.synth(5, S, P, L, model, False)
        steps, pb, pe, _ = g.soa()
        want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
        plan = DepthPlan(DeviceGraph(steps, pb, pe, S))
        desc = plan.describe()
        assert int(re.search(r"windows=\d+x(\d+)", desc).group(1)) == want_w, desc
        d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
        for _ in range(2):
            plan.seg_depth(d, u)
            plan.status()
            assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all(), desc


def test_counting_kernel_in_pieces_finds_the_same_facts(monkeypatch):
    """The plan-time counting kernel (runs per path, which paths are monotone, how an unsplit item runs) takes a path in one piece or in
    several that add their counts up (a graph of few long paths: `cfgL-4paths`); what the plan then is -- which kernel walks how many paths,
    how many items carry the no-claim tag: the description -- must not depend on it, nor may a count."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    for k in ("FLATGFA_SHORT_MAX", "FLATGFA_DEPTH_PATH", "FLATGFA_COUNT_PIECES", "FLATGFA_PACKED", "FLATGFA_NO_CLAIM"):
        monkeypatch.delenv(k, raising=False)
    shapes = [(40_000, 7, 90_001, "pangenome"), (40_000, 12, 33_333, "haplotype"), (40_000, 12, 33_333, "repeats"), (9_000, 300, 777, "chromosome"),
              (9_000, 3000, 97, "haplotype"), (300_000, 3, 400_003, "haplotype")]
    for S, P, L, model in shapes:
        g = pa.synth(11, S, P, L, model, False)
        steps, pb, pe, _ = g.soa()
        want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
        graph = DeviceGraph(steps, pb, pe, S)
        seen = {}
        for pieces in (None, "1", "2", "5", "64"):
            if pieces is None:
                monkeypatch.delenv("FLATGFA_COUNT_PIECES", raising=False)
            else:
                monkeypatch.setenv("FLATGFA_COUNT_PIECES", pieces)
            monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
            d = torch.full((S,), -3, dtype=torch.int32, device="cuda:0")
            u = torch.full((S,), -3, dtype=torch.int32, device="cuda:0")
            plan = DepthPlan(graph, first=(d, u))
            # (what the sizing and timing runs decide -- which pass 1 was faster, how deep the fullest sub-bucket got with the items dealt
            # as they came -- is not the counting kernel's business)
            seen[pieces] = re.sub(r" (pass1|bucket_cap|scratch_mb)=\S+|\(one-item shortcut\)| bitset_owners=tracked", "", plan.describe())
            assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all(), (pieces, seen[pieces])
            plan.seg_depth(d, u)
            plan.status()
            assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all(), (pieces, seen[pieces])
            plan.close()
        assert len(set(seen.values())) == 1, seen


@pytest.mark.parametrize("shape", [(300_000, 40, 60_000), (2_000_000, 9, 500_000), (3_000_000, 600, 30_000)])
def test_packed_buckets_take_their_marks_behind_the_first_answer(shape, monkeypatch):
    """A plan with packed buckets has its per-block no-claim marks made behind its first answer like any other (a mark rides on the
    ids of a whole block, and a block's first step starts a run in every build of k_scan: the marked build makes exactly the records
    the counting call laid the buckets out for).  The first answer comes from the unmarked build, the calls after `describe()` (which
    waits for the marks) from the marked one: all exact, and none of them ran out of room -- the plan is still bucketed and packed."""
    import re
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    S, P, L = shape
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    monkeypatch.setenv("FLATGFA_PACKED", "1")
    monkeypatch.delenv("FLATGFA_NO_CLAIM", raising=False)
    g = pa.synth(23, S, P, L, "repeats", False)
    steps, pb, pe, _ = g.soa()
    wd, wu = fo.seg_depth_with_uniq(pools_of(g))
    d = torch.full((S,), -1, dtype=torch.int32, device="cuda:0")
    u = torch.full((S,), -1, dtype=torch.int32, device="cuda:0")
    plan = DepthPlan(DeviceGraph(steps, pb, pe, S), first=(d, u))
    assert (d.cpu().numpy().view(np.uint32) == wd).all() and (u.cpu().numpy().view(np.uint32) == wu).all()
    plan.seg_depth(d, u)   # (starts the marks' job)
    plan.status()
    assert (d.cpu().numpy().view(np.uint32) == wd).all() and (u.cpu().numpy().view(np.uint32) == wu).all()
    desc = plan.describe()
    assert "buckets=packed" in desc and "path=bucketed" in desc and re.search(r"no_claim_chunks=\d+", desc), desc
    for k in range(3):
        d.fill_(-5)
        u.fill_(-5)
        plan.seg_depth(d, u if k != 1 else None)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == wd).all(), desc
        if k != 1:
            assert (u.cpu().numpy().view(np.uint32) == wu).all(), desc
    after = plan.describe()
    assert "buckets=packed" in after and "path=bucketed" in after, after
    plan.close()


def test_a_plan_asks_the_counting_call_before_it_makes_an_even_layout_of_a_gigabyte(monkeypatch):
    """An even layout gives every sub-bucket the room of the fullest.  Where the first estimate is past half the limit the plan runs
    the counting call first and packs its buckets when the even layout with headroom would pass the limit -- contigs that run along
    the graph fill a few sub-buckets of a window and leave the others empty -- and keeps the even layout when it would not (random
    walks fill them all alike).  FLATGFA_PACKED_ASK=1 makes small graphs ask."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    for k in ("FLATGFA_PACKED", "FLATGFA_WB", "FLATGFA_RANGE_SEGS", "FLATGFA_NO_CLAIM"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    monkeypatch.setenv("FLATGFA_SHORT_MAX", "0")
    monkeypatch.setenv("FLATGFA_PACKED_ASK", "1")
    seen = {}
    for name, (S, P, L, model) in {"walks": (1_000_000, 300, 20_000, "pangenome"), "contigs": (4_000_000, 1000, 50_000, "haplotype")}.items():
        g = pa.synth(31, S, P, L, model, False)
        steps, pb, pe, _ = g.soa()
        wd, wu = fo.seg_depth_with_uniq(pools_of(g))
        d = torch.full((S,), -1, dtype=torch.int32, device="cuda:0")
        u = torch.full((S,), -1, dtype=torch.int32, device="cuda:0")
        plan = DepthPlan(DeviceGraph(steps, pb, pe, S), first=(d, u))
        assert (d.cpu().numpy().view(np.uint32) == wd).all() and (u.cpu().numpy().view(np.uint32) == wu).all()
        seen[name] = plan.describe()
        assert "path=bucketed" in seen[name], seen
        for _ in range(2):
            plan.seg_depth(d, u)
            plan.status()
            assert (d.cpu().numpy().view(np.uint32) == wd).all() and (u.cpu().numpy().view(np.uint32) == wu).all(), seen
        assert "path=bucketed" in plan.describe()
        plan.close()
    assert "buckets=even" in seen["walks"], seen
    # (the contigs' even layout with headroom stays far below 2 GB at this size: asked, counted, and kept even as well -- the rule's other
    # branch is what `hap-chr20` takes, tools/plan_facts.py; here both must simply be right)
    assert "buckets=" in seen["contigs"], seen


@pytest.mark.parametrize("L", [1500, 70_001])
def test_one_step_back_wherever_it_lies_keeps_a_path_from_going_without_claims(L, monkeypatch):
    """What the plan-time counting kernel says of a path decides whether its records claim their segments: a path it takes for strictly
    monotone goes without, and a single revisit it overlooked would count a segment's unique depth twice.  Paths that walk their own
    stretch of ids upwards (or downwards) with ONE step back -- at the first steps, the last, around every boundary the kernel has
    (a lane's four steps, a wave's 256, a workgroup's 1024 and 4096, the cuts between the pieces of a path) -- between paths that
    are strictly monotone; spans that start at odd offsets, a pool that ends in a partial quad.  One piece per path and several."""
    from pollen_amd.device import DepthPlan, DeviceGraph
    import torch
    monkeypatch.setenv("FLATGFA_DEPTH_PATH", "bucketed")
    for k in ("FLATGFA_SHORT_MAX", "FLATGFA_PACKED", "FLATGFA_NO_CLAIM", "FLATGFA_COUNT_PIECES"):
        monkeypatch.delenv(k, raising=False)
    spots = [2, 3, 4, 5, 6, 7, 8, 9, 252, 253, 254, 255, 256, 257, 258, 259, 1020, 1021, 1022, 1023, 1024, 1025, 1026, 1027, L - 3, L - 2, L - 1]
    if L > 5000:
        spots += [4092, 4093, 4094, 4095, 4096, 4097, 4098, 4099, 8189, 8190, 8191, 8192, 8193]
        for pieces in (2, 3, 5, 8, 64):   # (8: what the plan takes by itself for this many paths of this length)
            for k in (range(1, pieces) if pieces < 64 else (1, 32, 63)):
                c = L * k // pieces
                spots += [c - 5, c - 4, c - 3, c - 2, c - 1, c, c + 1, c + 2, c + 3, c + 4]
    spots = sorted({q for q in spots if 2 <= q < L})
    chunks, pb, pe = [], [], []
    at = 0
    P = 0
    rng = np.random.default_rng(L)
    def add(ids, q=0, mod=None):
        nonlocal at, P
        # (steps no path walks: the next span starts at an odd offset -- with `mod`, at the one that puts the path's step q there in its quad)
        n_junk = 1 + P % 4 if mod is None else 1 + (mod - (at + 1 + q)) % 4
        junk = rng.integers(0, 50, size=n_junk).astype(np.uint32)
        chunks.append(junk << 1)
        at += len(junk)
        chunks.append((ids.astype(np.uint32) << 1) | (P & 1))
        pb.append(at)
        at += len(ids)
        pe.append(at)
        P += 1
    S = 300_000                     # (up to 2^20 segments the plan runs the counting kernel over every path)
    back = 0
    for q in spots:
        # upwards with the step back first in its quad (where a lane's, a wave's, a piece's stretch begins) and elsewhere in it; downwards
        for down, mod in ((False, 0), (False, 1 + q % 3), (True, None)):
            ids = np.arange(L, dtype=np.int64)
            ids[q:] -= 2            # step q goes back to the id of step q - 2, and the walk goes on from there
            ids += (P * 7919) % (S - L - 8)
            add(ids[::-1] if down else ids, q, mod)
            back += 1
        if q % 7 == 0:              # strictly monotone neighbours (these do go without claims)
            add(np.arange(L, dtype=np.int64) + (P * 7919) % (S - L - 8))
            add((np.arange(L, dtype=np.int64) + (P * 7919) % (S - L - 8))[::-1])
    steps = np.concatenate(chunks).astype(np.uint32)
    if len(steps) % 4 == 0:         # (the pool ends in a partial quad, right behind the last path)
        steps = steps[:-1]
        pe[-1] -= 1
    pb = np.array(pb, dtype=np.uint32)
    pe = np.array(pe, dtype=np.uint32)
    paths = np.zeros(P, dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, steps, np.zeros(S, dtype=fo.SEG_DT)
    wd, wu = fo.seg_depth_with_uniq(pools)
    assert back <= int((wd.astype(np.int64) - wu.astype(np.int64)).sum()) <= 2 * back   # (a path with a step back meets one or two segments twice)
    graph = DeviceGraph(steps, pb, pe, S)
    for pieces in (None, "1", "2", "3", "5", "64"):
        if pieces is None:
            monkeypatch.delenv("FLATGFA_COUNT_PIECES", raising=False)
        else:
            monkeypatch.setenv("FLATGFA_COUNT_PIECES", pieces)
        d = torch.full((S,), -3, dtype=torch.int32, device="cuda:0")
        u = torch.full((S,), -3, dtype=torch.int32, device="cuda:0")
        plan = DepthPlan(graph, first=(d, u))
        desc = plan.describe()
        bad = np.flatnonzero(u.cpu().numpy().view(np.uint32) != wu)
        assert (d.cpu().numpy().view(np.uint32) == wd).all() and len(bad) == 0, (pieces, bad[:8], desc)
        assert "no_claim_items=0" not in desc or "no_claim_paths=0" not in desc, desc   # (the monotone neighbours were found)
        plan.seg_depth(d, u)
        plan.status()
        assert (d.cpu().numpy().view(np.uint32) == wd).all() and (u.cpu().numpy().view(np.uint32) == wu).all(), (pieces, desc)
        plan.close()


def test_a_plan_that_goes_leaves_its_buckets_for_the_next_and_release_gives_them_back():
    """flatgfa_dev_release_scratch (include/flatgfa.h): the bucket array of a destroyed plan stays with the library (device memory in use
    does not drop), the next plan of the same graph takes it (no growth over ten plans made and dropped), and the release gives it back."""
    from pollen_amd.device import DepthPlan, DeviceGraph, _lib
    import torch
    S = 1_000_000
    g = pa.synth(9, S, 200, 50_000, "pangenome", False)
    steps, pb, pe, _ = g.soa()
    want_d, want_u = fo.seg_depth_with_uniq(pools_of(g))
    graph = DeviceGraph(steps, pb, pe, S)
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    def used():
        torch.cuda.synchronize()
        f, t = torch.cuda.mem_get_info()
        return (t - f) / 2**20
    _lib.lib().flatgfa_dev_release_scratch()
    base = used()
    seen = []
    for _ in range(10):
        plan = DepthPlan(graph, first=(d, u))
        assert (d.cpu().numpy().view(np.uint32) == want_d).all() and (u.cpu().numpy().view(np.uint32) == want_u).all()
        plan.close()
        seen.append(used())
    assert seen[0] > base + 16, (base, seen)          # (the array stayed: tens of megabytes at this size)
    assert max(seen) - min(seen) < 16, seen            # (and was taken again, not added to)
    _lib.lib().flatgfa_dev_release_scratch()
    assert used() < seen[-1] - 16, (base, seen, used())
