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
