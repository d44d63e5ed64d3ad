"""Product host code (C++ behind the C ABI) against the oracle: parser, pools, .flatgfa
container, GFA printer, accessors and their in-band error sentinels.  No GPU needed."""
import os
import subprocess

import numpy as np
import pytest

import pollen_amd as pa
from conftest import GOLDEN, ROOT, fixture_id, golden_gfas
from oracle import flatgfa_oracle as fo
from oracle import synth

FGFA = os.path.join(ROOT, "pollen_amd", "bin", "fgfa")


def read(path):
    with open(path, "rb") as f:
        return f.read()


def assert_same_pools(g: pa.FlatGFA, pools: fo.Pools):
    for name in fo.POOL_ORDER:
        assert g.pool(name).tobytes() == getattr(pools, name).tobytes(), name


@pytest.mark.parametrize("gfa", golden_gfas(), ids=fixture_id)
def test_parser_matches_oracle(gfa):
    assert_same_pools(pa.parse(gfa), fo.parse_gfa(read(gfa)))


QUIRKS = [
    b"",                                             # nothing
    b"S\t1\tA",                                      # no newline at all: the only line is dropped
    b"S\t1\tA\nP\tp\t1+\t*",                         # unterminated last line is dropped (memfile.rs:54-62)
    b"S\t1\tA\n\nP\tp\t1+\t*\n",                     # blank line panics (parse.rs:83)
    b"S\t1\tA\nP\tp\t1+,2+\t*\n",                    # unknown segment
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2+x\t*\n",          # one trailing garbage byte is swallowed
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2+xy\t*\n",         # two are not
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2\t*\n",            # dangling name without orientation is dropped
    b"S\t1\tA\nP\tp\t+\t*\n",                        # name 0 lookup
    b"S\t0\tA\nS\t1\tC\nP\tp\t1+\t*\n",              # a segment named 0 (wrapping name-1)
    b"S\t5\tA\nS\t5\tC\nP\tp\t5+\t*\n",              # duplicate name: later id wins in the hash map
    b"S\t1\tA\nS\t1\tC\nP\tp\t1+\t*\n",              # duplicate sequential name: first id wins
    b"S\t2\tA\nS\t1\tC\nS\t3\tG\nP\tp\t1+,2-,3+\t*\n",
    b"X\t1\n",                                       # unhandled kind
    b"S 1 A\n",                                      # no tab
    b"S\tx\tA\n",                                    # non-numeric name
    b"S\t1\n",                                       # missing tab after the name
    b"H\tVN:Z:1.0\nH\tVN:Z:2.0\n",                   # second header
    b"H\t\nH\tVN:Z:2.0\nS\t1\tA\n",                  # empty first header then another is fine
    b"S\t1\tA\tLN:i:1\tXX:Z:hi\nP\tp\t1+\t*\n",      # optional fields
    b"S\t1\tA\nP\tp\t1+\n",                          # path without an overlaps column
    b"S\t1\tA\nP\tp\t1+\t\t*\n",                     # double tab before '*'
    b"S\t1\tA\nL\t1\t+\t1\t-\t4M2I1D3N\nP\tp\t1+,1-\t4M\n",
    b"S\t1\tA\nL\t1\t+\t1\t-\t300M\n",               # alignment length > 255
    b"S\t1\tA\nL\t1\t+\t1\t-\t4\n",                  # alignment op missing
    b"S\t1\tA\nL\t1\t+\t2\t-\t0M\n",                 # link to unknown segment
    b"S\t1\tA\nL\t1\t+\t1\t-\t0M\tjunk\n",           # trailing junk on a link
    b"S\t1\tA\nP\tp\t1+\t1M,*\n",
    b"P\tp\t\t*\nS\t1\tA\n",                         # empty step list, path before segment
    b"S\t1\tA\r\nP\tp\t1+\t*\r\n",                   # CRLF
]


@pytest.mark.parametrize("text", QUIRKS + [
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2-,1+,2+,\t*\n",      # a trailing comma is accepted
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2-,,1+\t*\n",         # an empty item is not
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2-,1+,3+,1+\t*\n",    # unknown segment in a later chunk
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2-,1+2+\t*\n",        # a missing comma
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2-,+,1+\t*\n",        # a sign without a name
    b"S\t1\tA\nS\t2\tC\nP\tp\t1+,2-,1+,2+\t*\nL\t1\t+\t2\t+\t0M\nP\tq\t2+,2+,1-\t1M,2M\n",
], ids=lambda t: None)
def test_parser_quirks_match_oracle(text, step_parser):
    try:
        want = fo.parse_gfa(text)
    except fo.ParseError:
        want = None
    try:
        got = pa.parse_bytes(text)
    except pa.FlatGFAError:
        got = None
    assert (want is None) == (got is None), (want, got)
    if want is not None:
        assert_same_pools(got, want)


@pytest.fixture(params=["in order", "threads"])
def step_parser(request, monkeypatch):
    """The parser reads step lists in order, or -- large inputs; here forced, with 5-byte chunks --
    with several threads when all of them are plain (anything else must fall back)."""
    for k in ("FLATGFA_PARSE_THREADS", "FLATGFA_PARSE_CHUNK", "FLATGFA_PARSE_MIN_BYTES"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "threads":
        monkeypatch.setenv("FLATGFA_PARSE_THREADS", "3")
        monkeypatch.setenv("FLATGFA_PARSE_CHUNK", "5")
        monkeypatch.setenv("FLATGFA_PARSE_MIN_BYTES", "0")
    return request.param


def test_parser_differential_fuzz(step_parser):
    # Byte-level mutations of real fixtures: the product must fail exactly when the oracle fails
    # and otherwise build identical pools.
    rng = np.random.default_rng(1234)
    alphabet = np.frombuffer(b"\t\n+-,*0123456789SPLHMACGTx", dtype="u1")
    n_ok = n_err = 0
    for gfa in golden_gfas():
        base = np.frombuffer(read(gfa), dtype="u1")
        for _ in range(60):
            buf = base.copy()
            for _k in range(int(rng.integers(1, 4))):
                op = int(rng.integers(0, 3))
                pos = int(rng.integers(0, len(buf)))
                if op == 0:
                    buf[pos] = alphabet[int(rng.integers(0, len(alphabet)))]
                elif op == 1:
                    buf = np.delete(buf, pos)
                else:
                    buf = np.insert(buf, pos, alphabet[int(rng.integers(0, len(alphabet)))])
            text = buf.tobytes()
            try:
                want = fo.parse_gfa(text)
            except fo.ParseError:
                want = None
            try:
                got = pa.parse_bytes(text)
            except pa.FlatGFAError:
                got = None
            assert (want is None) == (got is None), text
            if want is not None:
                assert_same_pools(got, want)
                n_ok += 1
            else:
                n_err += 1
    assert n_ok > 50 and n_err > 50


def test_stream_mode_keeps_last_line_and_orders_links_first():
    text = b"S\t1\tA\nP\tp\t1+\t2M\nL\t1\t+\t1\t-\t3M\nP\tq\t1-\t*"
    mem = pa.parse_bytes(text)
    stream = pa.parse_stream_bytes(text)
    assert mem.path_count == 1 and stream.path_count == 2
    # parse_mem unwinds in file order (path's 2M first); parse_stream adds links first (parse.rs:63-72)
    assert (mem.pool("alignment") >> 8).tolist() == [2, 3]
    assert (stream.pool("alignment") >> 8).tolist() == [3, 2]


@pytest.mark.parametrize("gfa", golden_gfas(), ids=fixture_id)
def test_flatgfa_file_roundtrip(gfa, tmp_path):
    g = pa.parse(gfa)
    out = tmp_path / "g.flatgfa"
    g.write_flatgfa(str(out))
    assert out.read_bytes() == fo.dump_flatgfa(fo.parse_gfa(read(gfa)))   # file::dump, byte for byte
    assert_same_pools(pa.load(str(out)), fo.parse_gfa(read(gfa)))          # file::view


def test_flatgfa_view_honours_capacity_and_rejects_garbage(tmp_path):
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, "ref_ex2.gfa")))
    # a file whose regions have spare capacity (file.rs:163-167): len < capacity
    toc = [np.uint64(fo.MAGIC).tobytes()]
    body = []
    for i, name in enumerate(fo.POOL_ORDER):
        a = getattr(pools, name)
        cap = len(a) + i  # odd, unaligned padding
        toc.append(np.array([len(a), cap], dtype="<u8").tobytes())
        body.append(a.tobytes() + b"\xEE" * (i * a.dtype.itemsize))
    p = tmp_path / "spare.flatgfa"
    p.write_bytes(b"".join(toc + body))
    assert_same_pools(pa.load(str(p)), pools)
    bad = tmp_path / "bad.flatgfa"
    bad.write_bytes(b"\x00" * 200)
    with pytest.raises(pa.FlatGFAError):
        pa.load(str(bad))
    trunc = tmp_path / "trunc.flatgfa"
    trunc.write_bytes(fo.dump_flatgfa(pools)[:-3])
    with pytest.raises(pa.FlatGFAError):
        pa.load(str(trunc))
    with pytest.raises(pa.FlatGFAError):
        pa.load(str(tmp_path / "missing.flatgfa"))
    with pytest.raises(pa.FlatGFAError):
        pa.parse(str(tmp_path / "missing.gfa"))


def test_flatgfa_view_rejects_spans_that_leave_their_pool(tmp_path):
    """A span stored inside a pool that points outside its target pool: the reference panics when
    it indexes (pool.rs:341-347); here the file is refused when it is opened, so no accessor can
    run off a pool.  A step that names no segment is refused by whatever would follow it."""
    pools = fo.parse_gfa(read(os.path.join(GOLDEN, "ref_ex2.gfa")))

    def written(mutate, name):
        import copy
        q = copy.deepcopy(pools)
        mutate(q)
        f = tmp_path / name
        f.write_bytes(fo.dump_flatgfa(q))
        return str(f)

    def set_field(pool, idx, field, value):
        def m(q):
            a = getattr(q, pool).copy()
            a[field][idx] = value
            setattr(q, pool, a)
        return m

    for k, (pool, field) in enumerate([("paths", "steps_end"), ("paths", "name_end"), ("paths", "ov_end"),
                                       ("segs", "seq_end"), ("segs", "opt_end")]):
        with pytest.raises(pa.FlatGFAError):
            pa.load(written(set_field(pool, 0, field, 0x40000000), f"bad{k}.flatgfa"))
    with pytest.raises(pa.FlatGFAError):   # reversed span
        pa.load(written(set_field("paths", 1, "steps_start", 0x7FFFFFF0), "rev.flatgfa"))
    if len(pools.links):
        with pytest.raises(pa.FlatGFAError):
            pa.load(written(set_field("links", 0, "from_", 0x7FFFFFF0), "link.flatgfa"))

    def bad_step(q):
        a = q.steps.copy()
        a[1] = np.uint32((len(q.segs) + 5) << 1)
        q.steps = a
    g = pa.load(written(bad_step, "step.flatgfa"))     # loads: spans are fine
    assert g.get_path_step_count(0) == len(pools.steps[pools.paths["steps_start"][0]:pools.paths["steps_end"][0]])
    with pytest.raises(pa.FlatGFAError):
        g.gfa_text()
    with pytest.raises(pa.FlatGFAError):
        g.window_depth_table(0, 2)


ROUNDTRIP_OK = [g for g in golden_gfas() if "no-test-flip4" not in g and "edge_names_loops" not in g]


@pytest.mark.parametrize("gfa", ROUNDTRIP_OK, ids=fixture_id)
def test_gfa_text_roundtrip(gfa):
    # tests/turnt.toml:162-172: `fgfa < f` reproduces the input byte for byte
    assert pa.parse(gfa).gfa_text() == read(gfa)


def test_gfa_print_known_deviations():
    # D/I letters swap on the way out (gfaline.rs:178-184 vs print.rs:14-23) -- mirrored, not fixed
    g = pa.parse(os.path.join(GOLDEN, "edge_names_loops.gfa"))
    text = g.gfa_text()
    assert b"L\t10\t+\t5\t-\t2M1D\n" in text and b"\t2M,0M,0M,0M,0M,1M1I\n" in text
    # an empty overlap prints as 0M; a zero-step path cannot be printed (print.rs:48 indexes steps[0])
    with pytest.raises(pa.FlatGFAError):
        pa.parse_bytes(b"S\t1\tA\nP\te\t\t*\n").gfa_text()


def test_c_abi_sentinels():
    # flatgfa-c/src/lib.rs:92-165: out-of-range ids answer in band
    g = pa.parse(os.path.join(GOLDEN, "ref_tiny.gfa"))
    assert g.segment_count == 4 and g.path_count == 2
    assert g.get_seq(0) == b"CAAATAAG" and g.get_seq(4) is None
    assert g.get_path_name(0) == b"one" and g.get_path_name(2) is None
    assert g.get_path_step_count(1) == 4 and g.get_path_step_count(2) == 0xFFFFFFFF
    assert g.get_step(0, 2) == (3, False) and g.get_step(0, 3) is None and g.get_step(2, 0) is None
    assert g.find_path(b"two") == 1 and g.find_path(b"three") is None
    # the C example's walk (flatgfa-c/example/example.c) with the inner bound corrected
    walked = [[g.get_step(p, s) for s in range(g.get_path_step_count(p))] for p in range(g.path_count)]
    assert walked == [[(0, True), (1, True), (3, False)], [(0, True), (1, True), (2, True), (3, False)]]


@pytest.mark.parametrize("model", ["pangenome", "uniform", "chromosome", "haplotype", "repeats"])
def test_synth_cxx_matches_numpy_spec(model):
    for (seed, S, P, L) in [(1, 97, 5, 300), (7, 1000, 13, 257), (123456789, 1, 3, 10), (2, 50_000, 4, 5000)]:
        assert_same_pools(pa.synth(seed, S, P, L, model, True), synth.pools(seed, S, P, L, model))


def test_format_float_matches_oracle():
    rng = np.random.default_rng(7)
    xs = list(rng.random(200) * 10) + [k / 8 for k in range(64)] + [k / 200 for k in range(400)] + \
        [float("nan"), float("inf"), 0.0, 1e15 + 0.5, 123456.785]
    for x in xs:
        for digits in (0, 2, 4):
            assert pa.format_float(float(x), digits) == fo.format_float(float(x), digits)


def test_cli_host_commands():
    ex2 = os.path.join(GOLDEN, "ref_ex2.gfa")
    run = lambda *a, **k: subprocess.run([FGFA, *a], capture_output=True, **k)
    assert run("-I", ex2).stdout == read(ex2)
    assert run(input=read(ex2)).stdout == read(ex2)
    assert run("-I", ex2, "paths").stdout == b"path0\npath1\n"
    assert run("-I", ex2, "stats", "-S").stdout == b"#length\tnodes\tedges\tpaths\tsteps\n10\t5\t6\t2\t10\n"
    toc = run("-I", ex2, "toc").stdout.decode().split("\n")
    assert toc[:5] == ["header: 8", "segs: 5", "paths: 2", "links: 6", "steps: 10"]
    assert "segs: 120" in run("-I", ex2, "toc", "-b").stdout.decode()
    r = run("-I", os.path.join(GOLDEN, "missing.gfa"), "paths")
    assert r.returncode != 0 and b"cannot open" in r.stderr


# ---- the preallocated ("in-place") container: `fgfa -m -p N -o OUT [-I GFA]`, tests/turnt.toml:170-172 ----

@pytest.mark.parametrize("gfa", golden_gfas(), ids=fixture_id)
def test_prealloc_flatgfa_matches_oracle_and_round_trips(gfa, tmp_path):
    """Third round-trip variant of the reference (flatgfa_file_inplace): the file written with
    estimated capacities must be, byte for byte, what the restated prealloc_translate leaves
    (estimates of parse.rs:176-216 / file.rs:136-158, `len` of every `capacity` in use, zeros
    behind), and reading it back must print the same GFA as the graph it came from.  Where a pool
    does not fit its estimate the reference panics: an error here, on both sides."""
    text = open(gfa, "rb").read()
    g = pa.parse_bytes(text)
    pools = fo.parse_gfa(text)
    out = str(tmp_path / "inplace.flatgfa")
    try:
        want = fo.dump_flatgfa_prealloc(pools, fo.estimate_toc(text))
    except fo.ParseError:
        with pytest.raises(pa.FlatGFAError) as ei:
            g.write_flatgfa_prealloc(out, text)
        assert ei.value.code == -2
        return
    g.write_flatgfa_prealloc(out, text)
    got = open(out, "rb").read()
    assert got == want
    assert len(got) >= len(fo.dump_flatgfa(pools))  # capacities, not lengths, size the regions
    back = pa.load(out)
    assert str(back) == str(g)
    for name in fo.POOL_ORDER:
        assert back.pool(name).tobytes() == g.pool(name).tobytes(), name


def test_prealloc_flatgfa_guessed_capacities_and_overflow(tmp_path):
    text = open(os.path.join(GOLDEN, "ref_tiny.gfa"), "rb").read()
    g = pa.parse_bytes(text)
    pools = fo.parse_gfa(text)
    out = str(tmp_path / "guess.flatgfa")
    g.write_flatgfa_prealloc(out, None, factor=4)  # no text to measure (stdin in the reference): Toc::guess(4)
    assert open(out, "rb").read() == fo.dump_flatgfa_prealloc(pools, fo.guess_toc(4))
    assert str(pa.load(out)) == text.decode()
    with pytest.raises(fo.ParseError):
        fo.dump_flatgfa_prealloc(pools, fo.guess_toc(1))  # one path of capacity, two paths
    with pytest.raises(pa.FlatGFAError) as ei:
        g.write_flatgfa_prealloc(out, None, factor=1)
    assert ei.value.code == -2 and "paths" in str(ei.value)
    # estimates of known inputs (parse.rs:176-216 by hand): one H line of 10 bytes, 4 S lines, 2 P lines, 4 L lines
    caps = fo.estimate_toc(text)
    lines = text.split(b"\n")[:-1]
    assert caps[1] == 4 and caps[2] == 2 and caps[3] == 4
    assert caps[0] == len(lines[0]) and caps[4] == sum(len(ln) for ln in lines if ln[:1] == b"P") // 3
    assert caps[5] == sum(len(ln) for ln in lines if ln[:1] == b"S") and caps[10] == 4 + 4 + 2 + 8
    with pytest.raises(fo.ParseError):
        fo.estimate_toc(b"S\t1\tA\n\nS\t2\tC\n")  # a blank line is an unknown line type (the reference panics)


def test_cli_inplace_round_trip(tmp_path):
    # tests/turnt.toml:170-172: fgfa -m -p 128 -o X -I f.gfa ; fgfa -m -i X  must print f.gfa
    fgfa = os.path.join(ROOT, "pollen_amd", "bin", "fgfa")
    for name in ("ref_tiny", "ref_ex2", "standin_k"):
        gfa = os.path.join(GOLDEN, name + ".gfa")
        out = str(tmp_path / (name + ".inplace.flatgfa"))
        r = subprocess.run([fgfa, "-m", "-p", "128", "-o", out, "-I", gfa], capture_output=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([fgfa, "-m", "-i", out], capture_output=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == open(gfa, "rb").read()
        # from stdin there is no text to measure: the guess from -p sizes the file
        r = subprocess.run([fgfa, "-m", "-p", "8", "-o", out], input=open(gfa, "rb").read(), capture_output=True)
        assert r.returncode == 0, r.stderr
        assert os.path.getsize(out) > 100_000
        assert subprocess.run([fgfa, "-i", out], capture_output=True).stdout == open(gfa, "rb").read()


# ---- prealloc_translate proper: the text parsed straight into the mapped output (cli/main.rs:216-248, file.rs:255-272) ----

@pytest.mark.parametrize("gfa", golden_gfas(), ids=fixture_id)
def test_translate_prealloc_writes_the_same_file_with_no_graph_in_between(gfa, tmp_path):
    """Parser::for_slice over the file's own regions: byte for byte what the restated prealloc_translate leaves
    (and so what parse + write_flatgfa_prealloc leaves), for the text of a file (estimated capacities,
    parse_mem) and for stdin's (guessed capacities, parse_stream)."""
    text = open(gfa, "rb").read()
    out = str(tmp_path / "direct.flatgfa")
    out2 = str(tmp_path / "by_way_of_a_graph.flatgfa")
    for stream in (False, True):
        try:
            if stream:  # (the oracle restates parse_mem; parse_stream's order of links and paths is the heap route's, checked above)
                pa.parse_stream_bytes(text).write_flatgfa_prealloc(out2, None, factor=16)
                want = open(out2, "rb").read()
            else:
                want = fo.dump_flatgfa_prealloc(fo.parse_gfa(text), fo.estimate_toc(text))
        except (fo.ParseError, pa.FlatGFAError):
            with pytest.raises(pa.FlatGFAError) as ei:
                pa.translate_prealloc(text, out, factor=16, from_stream=stream)
            assert ei.value.code in (-2, -7)
            continue
        pa.translate_prealloc(text, out, factor=16, from_stream=stream)
        assert open(out, "rb").read() == want
        back = pa.load(out)
        assert str(back) == str(pa.parse_stream_bytes(text) if stream else pa.parse_bytes(text))


def test_translate_prealloc_overflow_parse_errors_and_threads(tmp_path, monkeypatch):
    text = open(os.path.join(GOLDEN, "ref_tiny.gfa"), "rb").read()
    out = str(tmp_path / "t.flatgfa")
    # a pool that does not fit its capacity: the reference's fixed store panics on that push, with the file as
    # file::init left it (capacities, every length 0) plus whatever was pushed before
    with pytest.raises(pa.FlatGFAError) as ei:
        pa.translate_prealloc(text, out, factor=1, from_stream=True)  # Toc::guess(1): one path of capacity, two paths
    assert ei.value.code == -2 and "paths" in str(ei.value)
    head = np.frombuffer(open(out, "rb").read()[:184], dtype="<u8")
    assert head[0] == 0xB1011054 and (head[1::2] == 0).all() and list(head[2::2]) == fo.guess_toc(1)
    assert os.path.getsize(out) == 184 + sum(c * s for c, s in zip(fo.guess_toc(1), (1, 24, 24, 16, 4, 1, 8, 4, 1, 1, 1)))
    # text the parser rejects
    for bad in (b"S\t1\tA\nX\tfoo\n", b"S\t1\tA\nP\tp\t2+\t*\n", b"S\t1\tA\nL\t1\t+\t1\t+\t4Q\n"):
        with pytest.raises(pa.FlatGFAError) as ei:
            pa.translate_prealloc(bad, out)
        assert ei.value.code in (-2, -7), bad
    # the step lists parsed by threads into the file (each chunk into its own stretch of the steps region)
    rng = np.random.default_rng(5)
    S, P, L = 3000, 40, 4000
    lines = [b"H\tVN:Z:1.0"] + [b"S\t%d\t%s" % (i + 1, b"ACGT"[: 1 + i % 4]) for i in range(S)]
    for p in range(P):
        ids = rng.integers(1, S + 1, L)
        lines.append(b"P\tpath%d\t" % p + b",".join(b"%d%s" % (i, b"+-"[i & 1: (i & 1) + 1]) for i in ids) + b"\t*")
    lines += [b"L\t%d\t+\t%d\t-\t%dM" % (i + 1, (i * 7) % S + 1, i % 200) for i in range(500)]
    big = b"\n".join(lines) + b"\n"
    want = fo.dump_flatgfa_prealloc(fo.parse_gfa(big), fo.estimate_toc(big))
    for env in ({"FLATGFA_PARSE_THREADS": "0"}, {"FLATGFA_PARSE_MIN_BYTES": "0", "FLATGFA_PARSE_CHUNK": "777", "FLATGFA_PARSE_THREADS": "5"}):
        for k in ("FLATGFA_PARSE_THREADS", "FLATGFA_PARSE_MIN_BYTES", "FLATGFA_PARSE_CHUNK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pa.translate_prealloc(big, out)
        assert open(out, "rb").read() == want, env
    # ... and a step list the threads give up on (a trailing comma, which the reference accepts): what they had
    # written is taken back (zeros again) and the sequential parser does it all
    odd = big.replace(b"\t*\n", b",\t*\n", 1)
    want = fo.dump_flatgfa_prealloc(fo.parse_gfa(odd), fo.estimate_toc(odd))
    pa.translate_prealloc(odd, out)
    assert open(out, "rb").read() == want


def test_keep_host_memory_is_a_host_call_and_leaves_malloc_working():
    """flatgfa_keep_host_memory (include/flatgfa.h) touches glibc's malloc parameters only -- no device -- and the Python package has
    called it once at import; large buffers still come and go after it, on and off."""
    import ctypes
    from pollen_amd import _lib
    lib = _lib.lib()
    for on in (1, 0, 1):
        assert lib.flatgfa_keep_host_memory(on) == 0
        bufs = [bytearray(48 << 20) for _ in range(3)]   # (beyond any mmap threshold)
        bufs[1][-1] = 7
        assert bufs[1][-1] == 7
        del bufs
        g = pa.synth(3, 2000, 20, 300, "pangenome", False)
        assert len(g.segments) == 2000
        g.close()
