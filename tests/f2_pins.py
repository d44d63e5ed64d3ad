"""What ties window / BED interval depth (flatgfa/src/ops/window_depth.rs:84-147,176-211) to rows the
reference itself pins.  The only vector the reference holds for f2 is flatgfa-sh/README.md:282-294 (four windows that
all read 2); everything here derives its expectations from the slow_odgi GOLDEN node depths (tests/golden/*.depth.tsv,
written by importing the reference's Python: make_golden.py) and from path depth (a3), never from the code under test:

  (a) one window over the whole path: every segment lies inside it, so window_depth.rs:135-137 adds
      (depth * len * 1.0) / L per step -- the same sum as measure_path's (depth.rs:116-131) divided term by term.
      Where L is a power of two every term and every partial sum is exact in f64, and the window's value equals
      path_depth's mean BIT FOR BIT; elsewhere the two agree to a few ulps.
  (b) windows of one base: the window inside a segment of length n reads (depth * n * (1 / n)) / 1, i.e. the
      segment's depth up to one rounding, so the emitted column (format_float(.., 4), window_depth.rs:160) must
      read the golden node depth of the segment that covers that base, base after base along the path.
  (c) intervals whose ends fall inside segments: the expected f64 is formed here, in plain Python floats, from the
      golden depths in assign_depths' order (window_depth.rs:116-147) and compared bitwise.

Test infrastructure only (used by tests/test_next_rows_oracle.py on the oracle and by tests/test_gpu_next_rows.py on
the product)."""
import os

import numpy as np

from conftest import GOLDEN

# fixtures that have a slow_odgi golden depth table AND a GFA text in the tree
PINNED_GRAPHS = ["ref_ex1", "ref_ex2", "ref_tiny", "kat_slow_odgi_readme", "kat_window_depth", "standin_note5", "standin_k",
                 "edge_names_loops", "ref_handmade_flip1", "ref_handmade_crush1"]


def read(path):
    with open(path, "rb") as f:
        return f.read()


def golden_node_depth(name):
    """{segment name -> depth} from the slow_odgi golden table (`#node.id\\tdepth\\tdepth.uniq`)."""
    out = {}
    for line in read(os.path.join(GOLDEN, name + ".depth.tsv")).splitlines():
        if line.startswith(b"#") or not line:
            continue
        node, depth, _uniq = line.split(b"\t")
        out[int(node)] = int(depth)
    return out


def path_layout(pools, pid, node_depth):
    """[(segment length, golden depth)] along path `pid`, one entry per step (weighted_depths, window_depth.rs:84-101);
    the depth of a step's segment is looked up by the segment's NAME in the golden table."""
    p = pools.paths[pid]
    lens = pools.seg_lens()
    out = []
    for h in pools.steps[int(p["steps_start"]):int(p["steps_end"])]:
        seg = int(h) >> 1
        out.append((int(lens[seg]), node_depth[int(pools.segs[seg]["name"])]))
    return out


def per_base_depth(layout):
    """The golden depth of the segment that covers each base of the path, base after base."""
    out = []
    for n, d in layout:
        out.extend([d] * n)
    return out


def expected_intervals(layout, starts, ends):
    """assign_depths (window_depth.rs:116-147) in plain Python floats over the golden depths: per window, in the order
    the path's segments come, (depth * len) as f64 * (overlap / len) / window length."""
    depths = [0.0] * len(starts)
    cur, pos = 0, 0
    for n, d in layout:
        lo, hi = pos, pos + n
        pos = hi
        while cur < len(starts):
            a, b = int(starts[cur]), int(ends[cur])
            s, e = max(a, lo), min(b, hi)
            if e > s:
                amt = float(e - s) / float(hi - lo)
                depths[cur] += (float(d * n) * amt) / float(b - a)
            if b > hi:
                break
            cur += 1
    return np.array(depths, dtype=np.float64)


def mean_depth(layout):
    """measure_path (depth.rs:116-131): (sum depth * len) as f64 / (sum len) as f64, over the golden depths."""
    total = sum(n for n, _ in layout)
    return total, (float(sum(n * d for n, d in layout)) / float(total)) if total else float("nan")


def cut_points(total, seed, n_cuts=12):
    """Sorted, disjoint intervals that cover [0, total) with ends inside segments more often than not."""
    rng = np.random.default_rng(seed)
    inner = rng.integers(1, total, size=min(n_cuts, max(total - 1, 0))) if total > 1 else np.array([], dtype=np.int64)
    edges = np.unique(np.concatenate([[0, total], inner]))
    return edges[:-1].astype(np.uint64), edges[1:].astype(np.uint64)


def pow2_gfa(seed, n_segs=40, n_paths=5, log2_len=9):
    """GFA text whose every path is exactly 2^log2_len bases long: random walks over segments of 1..16 bases, the last
    step a segment of its own that fills the path up.  Returns the bytes."""
    rng = np.random.default_rng(seed)
    lens = [int(x) for x in rng.integers(1, 17, size=n_segs)]
    seqs = ["ACGT"[i & 3] * n for i, n in enumerate(lens)]
    target = 1 << log2_len
    lines, paths = [], []
    for p in range(n_paths):
        steps, total = [], 0
        while True:
            s = int(rng.integers(0, n_segs))
            if total + lens[s] > target - 1:
                break
            steps.append("%d%s" % (s + 1, "+-"[int(rng.integers(0, 2))]))
            total += lens[s]
        fill = target - total                      # 1 <= fill: a filler segment of the path's own
        seqs.append("N" * fill)
        steps.append("%d+" % len(seqs))
        paths.append("P\tw%d\t%s\t*" % (p, ",".join(steps)))
    for i, q in enumerate(seqs):
        lines.append("S\t%d\t%s" % (i + 1, q))
    return ("\n".join(lines + paths) + "\n").encode()
