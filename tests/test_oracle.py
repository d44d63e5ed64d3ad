"""The oracle (oracle/) against the reference's known answers and the slow_odgi golden vectors.

This is what pins the CPU restatement: every later parity claim (HIP vs oracle) rests on it.
"""
import json
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, fixture_id, golden_gfas
from oracle import flatgfa_oracle as fo
from oracle import synth

CFG_S = dict(seed=1, S=10_000, P=100, L=10_000, model="pangenome")


def read(path):
    with open(path, "rb") as f:
        return f.read()


@pytest.mark.parametrize("gfa", golden_gfas(), ids=fixture_id)
def test_depth_table_matches_slow_odgi(gfa):
    # slow_odgi/slow_odgi/depth.py:6-16 output == SegDepth::emit (ops/depth.rs:67-82) for
    # well-formed graphs with unique path names.
    pools = fo.parse_gfa(read(gfa))
    assert fo.fgfa_depth(pools, seg_depth_flag=True) == read(gfa[:-4] + ".depth.tsv")


def test_kat_slow_odgi_readme():
    # slow_odgi/README.md:144-178: all paths, and the {x, y} subset the README prints
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, "kat_slow_odgi_readme.gfa")))
    d, u = fo.seg_depth_with_uniq(pools)
    assert d.tolist() == [2, 0, 4, 2] and u.tolist() == [2, 0, 3, 2]
    assert read(os.path.join(GOLDEN, "kat_slow_odgi_readme.depth_subset.tsv")) == \
        b"#node.id\tdepth\tdepth.uniq\n1\t2\t2\n2\t0\t0\n3\t3\t2\n4\t1\t1\n"


def test_kat_flash_readme_note5_standin():
    # flatgfa-sh/README.md:31-36 and :51-59 (note5.gfa is not in the tree; stand-in fixture)
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, "standin_note5.gfa")))
    assert fo.fgfa_depth(pools, True) == b"#node.id\tdepth\tdepth.uniq\n1\t2\t2\n2\t0\t0\n3\t2\t2\n4\t2\t2\n"
    assert fo.fgfa_depth(pools, False) == b"#path\tstart\tend\tmean.depth\n5\t0\t13\t2\n5-\t0\t13\t2\n"
    # -r with one known and one unknown name: unknown names are dropped (cmds.rs:270-274)
    assert fo.fgfa_depth(pools, False, [b"nope", b"5-"]) == b"#path\tstart\tend\tmean.depth\n5-\t0\t13\t2\n"


def test_kat_flash_readme_k_standin():
    # flatgfa-sh/README.md:267-270: x 0 50 1.9 / y 0 50 1.9
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, "standin_k.gfa")))
    assert fo.fgfa_depth(pools, False) == b"#path\tstart\tend\tmean.depth\nx\t0\t50\t1.9\ny\t0\t50\t1.9\n"


def test_seg_depth_equals_with_uniq_depth():
    for gfa in golden_gfas():
        pools = fo.parse_gfa(read(gfa))
        assert (fo.seg_depth(pools) == fo.seg_depth_with_uniq(pools)[0]).all()


@pytest.mark.parametrize("x,digits,want", [
    (2.0, 2, "2"), (1.9, 2, "1.9"), (1.905, 2, "1.91"), (0.125, 2, "0.12"), (0.375, 2, "0.38"),
    (2.675, 2, "2.67"), (100.0, 2, "100"), (0.0, 2, "0"), (10.10, 2, "10.1"), (1234.5678, 4, "1234.5678"),
    (0.004, 2, "0"), (0.005, 2, "0.01"), (float("nan"), 2, "NaN"), (float("inf"), 2, "inf"), (1e21, 2, "1" + "0" * 21),
])
def test_format_float(x, digits, want):
    # ops/depth.rs:192-197; ties resolve on the exact binary value (0.125 -> "0.12", 2.675 is below the tie)
    if x == 0.005:
        want = "0.01" if float.__format__(0.005, ".2f") == "0.01" else "0"
    assert fo.format_float(x, digits) == want


def test_empty_path_mean_is_nan():
    # P line with an empty step list: 0/0 as f64 -> "NaN" (depth.rs:129)
    pools = fo.parse_gfa(b"S\t1\tAC\nP\te\t\t*\nP\tp\t1+\t*\n")
    ln, dp = fo.path_depth(pools)
    assert ln.tolist() == [0, 2] and np.isnan(dp[0]) and dp[1] == 1.0
    assert fo.fgfa_depth(pools, False) == b"#path\tstart\tend\tmean.depth\ne\t0\t0\tNaN\np\t0\t2\t1\n"


def test_flatgfa_container_roundtrip():
    for gfa in golden_gfas():
        pools = fo.parse_gfa(read(gfa))
        blob = fo.dump_flatgfa(pools)
        assert len(blob) == 184 + sum(getattr(pools, n).nbytes for n in fo.POOL_ORDER)
        back = fo.view_flatgfa(blob)
        for n in fo.POOL_ORDER:
            assert getattr(back, n).tobytes() == getattr(pools, n).tobytes()


def test_cfgS_synthetic_matches_slow_odgi_golden():
    # BASELINE.json configs[1]: 10k segments / 1M steps, bit-exact vs slow_odgi
    manifest = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    pools = synth.pools(**CFG_S)
    assert hashlib.sha256(pools.steps.tobytes()).hexdigest() == manifest["synth_cfgS.steps.u32le"]
    table = fo.fgfa_depth(pools, True)
    assert hashlib.sha256(table).hexdigest() == manifest["synth_cfgS.depth.tsv"]
    assert table == read(os.path.join(GOLDEN, "synth_cfgS.depth.tsv"))


SYNTH_MORE = {  # tests/golden/make_golden.py
    "synth_short": dict(seed=7, S=8_000, P=600, L=800, model="pangenome"),
    "synth_long": dict(seed=9, S=12_000, P=8, L=70_000, model="pangenome"),
    "synth_uniform": dict(seed=11, S=6_000, P=40, L=3_000, model="uniform"),
    "synth_chrom": dict(seed=13, S=15_000, P=12, L=50_000, model="chromosome"),  # paths along the graph, every other one downwards
}


@pytest.mark.parametrize("name", sorted(SYNTH_MORE))
def test_more_synthetic_graphs_match_slow_odgi_goldens(name):
    manifest = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    pools = synth.pools(**SYNTH_MORE[name])
    assert hashlib.sha256(pools.steps.tobytes()).hexdigest() == manifest[name + ".steps.u32le"]
    table = fo.fgfa_depth(pools, True)
    assert hashlib.sha256(table).hexdigest() == manifest[name + ".depth.tsv"]
    assert table == read(os.path.join(GOLDEN, name + ".depth.tsv"))


def test_out_of_range_is_an_error():
    pools = fo.parse_gfa(b"S\t1\tA\nP\tp\t1+\t*\n")
    pools.steps[0] = 5 << 1
    with pytest.raises(fo.ParseError):
        fo.seg_depth_with_uniq(pools)
    pools.steps[0] = 0
    pools.paths["steps_end"][0] = 9
    with pytest.raises(fo.ParseError):
        fo.seg_depth(pools)


@pytest.mark.parametrize("gfa", golden_gfas()[:6], ids=fixture_id)
def test_cpu_process_matches_golden(gfa, tmp_path):
    """oracle/fgfa_depth_cpu.c (the CPU process bench.py times beside the product's CLI): .flatgfa in,
    the two depth tables out, byte for byte what slow_odgi / the oracle's emitters give."""
    import subprocess
    pools = fo.parse_gfa(read(gfa))
    f = tmp_path / "g.flatgfa"
    f.write_bytes(fo.dump_flatgfa(pools))
    exe = fo.cpu_cli()
    assert subprocess.run([exe, str(f), "-d"], capture_output=True, check=True).stdout == read(gfa[:-4] + ".depth.tsv")
    assert subprocess.run([exe, str(f)], capture_output=True, check=True).stdout == fo.fgfa_depth(pools, False)
