"""Oracle pins for the rows next to the depth path: path-pair overlap, subset-paths
node depth, window / BED interval depth."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, fixture_id, golden_gfas
from oracle import flatgfa_oracle as fo


def read(path):
    with open(path, "rb") as f:
        return f.read()


@pytest.mark.parametrize("gfa", golden_gfas(), ids=fixture_id)
def test_overlap_matches_slow_odgi(gfa):
    # slow_odgi/slow_odgi/overlap.py:17-32 with every path as a query (make_golden.py)
    pools = fo.parse_gfa(read(gfa))
    names = [pools.path_name(i) for i in range(len(pools.paths))]
    assert fo.overlap_table(pools, names) == read(gfa[:-4] + ".overlap.tsv")


def test_overlap_is_on_oriented_handles():
    # slow_odgi/README.md depth example: y = {1+,3-} touches x = {1+,3+,4+} (via 1+) but NOT z = {3+,4+}
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, "kat_slow_odgi_readme.gfa")))
    t = fo.path_touches(pools, [0, 1, 2])
    assert t.tolist() == [[0, 1, 1], [1, 0, 0], [1, 0, 0]]


@pytest.mark.parametrize("name", ["kat_slow_odgi_readme", "ref_ex1", "ref_ex2", "edge_names_loops"])
def test_subset_depth_matches_slow_odgi(name):
    # slow_odgi depth --paths FILE (depth.py:12); goldens from make_golden.py
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, name + ".gfa")))
    want = read(os.path.join(GOLDEN, name + ".depth_subset.tsv"))
    ids = [fo.find_path(pools, ln.strip().encode()) for ln in open(os.path.join(GOLDEN, name + ".subset.paths")) if ln.strip()]
    d, u = fo.seg_depth_subset(pools, ids)
    assert fo.emit_seg_depth(pools, d, u) == want


def test_window_depth_known_answer():
    # flatgfa-sh/README.md:282-294 (windows.sh on note5; stand-in fixture)
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, "standin_note5.gfa")))
    want = b"5\t0\t4\t2\n5\t4\t8\t2\n5\t8\t12\t2\n5\t12\t13\t2\n"
    assert fo.window_depth_table(pools, b"5", 4) == want
    assert fo.bed_depth_table(pools, b"#path\tstart\tend\n5\t0\t4\n5\t4\t8\n5\t8\t12\n5\t12\t13\n") == want


def test_interval_depth_weights_partial_overlaps():
    # two segments of depth 2 and 1 (lengths 4 and 2): window [2,6) = (2*4*(2/4))/4 + (1*2*(2/2))/4 = 1.5
    pools = fo.parse_gfa(b"S\t1\tAAAA\nS\t2\tCC\nP\tp\t1+,2+\t*\nP\tq\t1+\t*\n")
    got = fo.interval_depth(pools, 0, [0, 2, 0], [4, 6, 6])
    assert got.tolist()[:2] == [2.0, 1.5]
    assert fo.window_depth_table(pools, b"p", 4) == b"p\t0\t4\t2\np\t4\t6\t1\n"


# Hand-computed answers for assign_depths (window_depth.rs:116-147) on tests/golden/kat_window_depth.gfa:
# segments 1 (4 bp), 2 (2 bp), 3 (3 bp) with node depths 2, 1, 3; path x = 1+,2+,3+ (9 bp), y = 1+,3+ (7 bp).
# A window's mean depth is the sum over the segments it meets of depth*len * (overlap/len) / window length.
WINDOW_KATS = [
    # x, windows of 3: [0,3) = 8*(3/4)/3 = 2;  [3,6) = 8*(1/4)/3 + 2*(2/2)/3 = 4/3;  [6,9) = 9/3 = 3
    ("window", b"x", 3, b"x\t0\t3\t2\nx\t3\t6\t1.3333\nx\t6\t9\t3\n"),
    # x, windows of 5: [0,5) = 8/5 + 2*(1/2)/5 = 1.8;  [5,9) = 2*(1/2)/4 + 9/4 = 2.5
    ("window", b"x", 5, b"x\t0\t5\t1.8\nx\t5\t9\t2.5\n"),
    # y, windows of 4: [0,4) = 8/4 = 2;  [4,7) = 9/3 = 3
    ("window", b"y", 4, b"y\t0\t4\t2\ny\t4\t7\t3\n"),
    # a BED interval that cuts two segments: x [2,7) = 8*(2/4)/5 + 2/5 + 9*(1/3)/5 = 0.8 + 0.4 + 0.6 = 1.8;
    # and the rest of the path: x [7,9) = 9*(2/3)/2 = 3   (intervals are sorted and disjoint, window_depth.rs:110-115)
    ("bed", b"x\t2\t7\nx\t7\t9\n", None, b"x\t2\t7\t1.8\nx\t7\t9\t3\n"),
]


def test_window_depth_hand_computed():
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, "kat_window_depth.gfa")))
    d, _ = fo.seg_depth_with_uniq(pools)
    assert d.tolist() == [2, 1, 3]
    for kind, a, b, want in WINDOW_KATS:
        got = fo.window_depth_table(pools, a, b) if kind == "window" else fo.bed_depth_table(pools, a)
        assert got == want, (kind, a, b)
    # the f64 value behind "1.3333", accumulated in the reference's order
    assert fo.interval_depth(pools, 0, [3], [6]).tolist() == [8 * 0.25 / 3 + 2 * 1.0 / 3]


def test_bed_parser_quirks():
    assert fo.parse_bed(b"#h\np\t1\t5\nq\t7 9\n") == [(b"p", 1, 5), (b"q", 7, 9)]   # any one byte separates start/end
    assert fo.parse_bed(b"p\t1\t5") == []                                             # unterminated last line dropped
    for bad in (b"\n", b"p\n", b"p\tx\t1\n", b"p\t1\n", b"p\t1\t\n"):
        with pytest.raises(fo.ParseError):
            fo.parse_bed(bad)


# ---- f2 tied to rows the reference pins (tests/f2_pins.py says how): slow_odgi's golden node depths and path depth ----
import f2_pins  # noqa: E402


def _pinned_pools(name):
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, name + ".gfa")))
    return pools, f2_pins.golden_node_depth(name)


@pytest.mark.parametrize("name", f2_pins.PINNED_GRAPHS)
def test_one_base_windows_read_the_golden_node_depth(name):
    # (b) window_depth.rs:135-137 with windows of one base: the emitted column is the golden depth of the covering segment
    pools, node_depth = _pinned_pools(name)
    for pid in range(len(pools.paths)):
        layout = f2_pins.path_layout(pools, pid, node_depth)
        want = f2_pins.per_base_depth(layout)
        if not want:
            continue
        nm = pools.path_name(pid)
        if fo.find_path(pools, nm) != pid:   # (a duplicate name: the CLI route finds the first one only)
            continue
        rows = fo.window_depth_table(pools, nm, 1).splitlines()
        assert len(rows) == len(want)
        for i, row in enumerate(rows):
            assert row == b"%s\t%d\t%d\t%d" % (nm, i, i + 1, want[i]), (name, pid, i)


@pytest.mark.parametrize("name", f2_pins.PINNED_GRAPHS)
def test_intervals_that_cut_segments_match_the_golden_depths_summed_in_reference_order(name):
    # (c) assign_depths' f64 order restated in plain Python floats over the golden depths: bitwise
    pools, node_depth = _pinned_pools(name)
    for pid in range(len(pools.paths)):
        layout = f2_pins.path_layout(pools, pid, node_depth)
        total = sum(n for n, _ in layout)
        if total == 0:
            continue
        for seed in range(3):
            st, en = f2_pins.cut_points(total, 100 * pid + seed)
            got = fo.interval_depth(pools, pid, st, en)
            want = f2_pins.expected_intervals(layout, st, en)
            assert got.tobytes() == want.tobytes(), (name, pid, seed)
        # intervals need not cover the path: gaps, and an interval that ends where a segment does
        if total >= 4:
            st = np.array([1, total // 2], dtype=np.uint64)
            en = np.array([total // 2 - 1 if total // 2 - 1 > 1 else 2, total - 1], dtype=np.uint64)
            assert fo.interval_depth(pools, pid, st, en).tobytes() == f2_pins.expected_intervals(layout, st, en).tobytes()


@pytest.mark.parametrize("name", f2_pins.PINNED_GRAPHS)
def test_whole_path_window_agrees_with_path_depth(name):
    # (a), any length: one window over the whole path and measure_path's mean are the same rational number
    pools, node_depth = _pinned_pools(name)
    ln, mean = fo.path_depth(pools)
    for pid in range(len(pools.paths)):
        layout = f2_pins.path_layout(pools, pid, node_depth)
        total, want_mean = f2_pins.mean_depth(layout)
        assert int(ln[pid]) == total
        if total == 0:
            continue
        assert mean[pid] == want_mean                      # a3 against the golden depths (one division: exact agreement)
        w = fo.interval_depth(pools, pid, [0], [total])[0]
        assert abs(w - mean[pid]) <= 8 * np.spacing(mean[pid]), (name, pid, w, mean[pid])


@pytest.mark.parametrize("seed", range(6))
def test_whole_path_window_equals_path_depth_bitwise_where_the_length_is_a_power_of_two(seed):
    # (a), L = 2^k: every term depth * len / L and every partial sum is exact, so the two f64 values are identical
    pools = fo.parse_gfa(f2_pins.pow2_gfa(seed, log2_len=7 + seed))
    ln, mean = fo.path_depth(pools)
    for pid in range(len(pools.paths)):
        assert int(ln[pid]) == 1 << (7 + seed)
        w = fo.interval_depth(pools, pid, [0], [int(ln[pid])])
        assert w.tobytes() == mean[pid:pid + 1].tobytes(), (seed, pid)
        nm = pools.path_name(pid)
        assert fo.window_depth_table(pools, nm, int(ln[pid])) == b"%s\t0\t%d\t%s\n" % (nm, int(ln[pid]), fo.format_float(float(mean[pid]), 4).encode())


def test_f2_pins_on_a_synthetic_graph_with_a_slow_odgi_golden():
    # the same three ties on a graph of 15 k segments whose node depths slow_odgi wrote (synth_chrom.depth.tsv)
    from oracle import synth
    pools = synth.pools(seed=13, S=15_000, P=12, L=50_000, model="chromosome")
    node_depth = f2_pins.golden_node_depth("synth_chrom")
    for pid in (0, 5, 11):
        layout = f2_pins.path_layout(pools, pid, node_depth)
        total, want_mean = f2_pins.mean_depth(layout)
        ln, mean = fo.path_depth(pools, [pid])
        assert int(ln[0]) == total and mean[0] == want_mean
        st, en = f2_pins.cut_points(total, pid, n_cuts=200)
        assert fo.interval_depth(pools, pid, st, en).tobytes() == f2_pins.expected_intervals(layout, st, en).tobytes()
        # one-base windows over the first 3000 bases
        st1 = np.arange(3000, dtype=np.uint64)
        got = fo.interval_depth(pools, pid, st1, st1 + 1)
        want = np.array(f2_pins.per_base_depth(layout)[:3000], dtype=np.float64)
        assert [fo.format_float(float(x), 4) for x in got] == [fo.format_float(float(x), 4) for x in want]
