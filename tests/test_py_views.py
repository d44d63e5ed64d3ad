"""The reference's Python-binding tests (flatgfa-py/test/test_flatgfa.py), restated against the
ctypes mirror: same fixture (tiny.gfa = tests/golden/ref_tiny.gfa), same assertions."""
import os

import pytest

import pollen_amd as flatgfa
from conftest import GOLDEN

TEST_GFA = os.path.join(GOLDEN, "ref_tiny.gfa")


@pytest.fixture
def gfa():
    with open(TEST_GFA, "rb") as f:
        return flatgfa.parse_bytes(f.read())


def test_segs(gfa):
    assert len(gfa.segments) == 4
    seg = gfa.segments[0]
    assert seg.name == 1
    assert seg.sequence() == b"CAAATAAG"
    assert len(seg) == 8
    seg = list(gfa.segments)[2]
    assert seg.name == 3
    assert str(seg) == "S\t3\tTTG"


def test_segs_find(gfa):
    seg = gfa.segments.find(3)
    assert seg.id == 2
    assert seg.sequence() == b"TTG"
    assert gfa.segments.find(99) is None


def test_paths(gfa):
    assert len(gfa.paths) == 2
    assert len(list(gfa.paths)) == 2
    path = gfa.paths[0]
    assert path.name == "one"
    assert str(path) == "P\tone\t1+,2+,4-\t*"


def test_paths_find(gfa):
    path = gfa.paths.find("two")
    assert path.id == 1
    assert path.name == "two"
    assert gfa.paths.find("three") is None


def test_path_steps(gfa):
    path = gfa.paths[1]
    assert len(path) == 4
    assert len(list(path)) == 4
    step = path[0]
    assert step.segment.name == 1
    assert step.is_forward
    assert str(step) == "1+"


def test_links(gfa):
    assert len(gfa.links) == 4
    assert len(list(gfa.links)) == 4
    link = gfa.links[1]
    assert link.from_.segment.name == 2
    assert link.from_.is_forward
    assert link.to.segment.name == 4
    assert not link.to.is_forward
    assert str(link) == "L\t2\t+\t4\t-\t0M"


def test_gfa_str(gfa):
    with open(TEST_GFA, "r") as f:
        assert str(gfa) == f.read()


def test_read_write_gfa(gfa, tmp_path):
    gfa_path = str(tmp_path / "tiny.gfa")
    gfa.write_gfa(gfa_path)
    with open(TEST_GFA, "rb") as orig_f, open(gfa_path, "rb") as written_f:
        assert orig_f.read() == written_f.read()
    new_gfa = flatgfa.parse(gfa_path)
    assert len(new_gfa.segments) == len(gfa.segments)


def test_read_write_flatgfa(gfa, tmp_path):
    flatgfa_path = str(tmp_path / "tiny.flatgfa")
    gfa.write_flatgfa(flatgfa_path)
    new_gfa = flatgfa.load(flatgfa_path)
    assert len(new_gfa.segments) == len(gfa.segments)
    assert str(new_gfa) == str(gfa)


def test_eq(gfa):
    assert gfa.segments[0] == gfa.segments[0]
    assert gfa.segments[0] != gfa.segments[1]
    assert gfa.paths[0] == gfa.paths[0]
    assert gfa.paths[0] != gfa.paths[1]
    assert gfa.links[0] == gfa.links[0]
    assert gfa.links[0] != gfa.links[1]
    assert gfa.links[1].from_ == gfa.links[2].from_
    assert gfa.links[1].from_ != gfa.links[1].to


def test_hash(gfa):
    d = {gfa.segments[0]: "foo", gfa.paths[0]: "bar", gfa.links[0]: "baz", gfa.links[1].from_: "qux"}
    assert d[gfa.segments[0]] == "foo"
    assert d[gfa.paths[0]] == "bar"
    assert d[gfa.links[0]] == "baz"
    assert d[gfa.links[1].from_] == "qux"


def test_slice(gfa):
    assert len(gfa.segments[1:3]) == 2
    assert len(gfa.segments[2:]) == len(gfa.segments) - 2
    assert gfa.segments[1:3][0].name == gfa.segments[1].name
    assert len(gfa.paths[1:]) == 1
    assert len(gfa.links[2:100]) == 2
    assert len(list(gfa.paths[:1])) == 1
    path = gfa.paths[0]
    assert len(path[2:]) == len(path) - 2
    assert path[2:][0] == path[2]
    assert len(list(path[2:])) == len(path) - 2


def test_depth_example_loop_equals_seg_depth_text(gfa):
    # flatgfa-py/examples/depth.py: the reference computes depth with a Python loop over the views
    from collections import Counter
    depths = Counter()
    for path in gfa.paths:
        for step in path:
            depths[step.segment.id] += 1
    assert [depths[s.id] for s in gfa.segments] == [2, 2, 1, 2]
