#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the reference's own Python
implementation (mygfa + slow_odgi), imported from /root/reference.

Runs ONLY in the authoring container (the reference does not travel); the
outputs are committed.  For every ``*.gfa`` here it writes

  <name>.depth.tsv          slow_odgi depth <gfa>                (slow_odgi/depth.py:6-16)
  <name>.depth_subset.tsv   slow_odgi depth --paths <name>.subset.paths <gfa>   (if the .paths file exists)
  <name>.overlap.tsv        slow_odgi overlap --paths <name>.overlap.paths <gfa> (all paths if no file)

and for the cfg-S synthetic graph (10k segments / 1M steps; BASELINE.json
configs[1]) -- whose 6 MB GFA text is NOT committed, it is regenerated from
oracle/synth.py -- the file ``synth_cfgS.depth.tsv``.  MANIFEST.json pins the
sha256 of every input and output.

usage: python tests/golden/make_golden.py
"""
import contextlib
import glob
import hashlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "mygfa"))
sys.path.insert(0, os.path.join(REF, "slow_odgi"))
sys.path.insert(0, ROOT)

import mygfa  # noqa: E402
from slow_odgi import depth as so_depth  # noqa: E402
from slow_odgi import overlap as so_overlap  # noqa: E402

from oracle import synth  # noqa: E402

CFG_S = dict(seed=1, S=10_000, P=100, L=10_000, model="pangenome")
# Further synthetic graphs, chosen so that each of the device kernels that walk paths is pinned to
# slow_odgi by at least one golden table: paths of 10 k steps (cfg-S: the medium-path kernel),
# short paths (k_scan_short), long paths (k_scan), uniform-random ids (no runs to speak of), and
# paths that run along the graph, half of them downwards (k_scan with step -1).
SYNTH_MORE = {
    "synth_short": dict(seed=7, S=8_000, P=600, L=800, model="pangenome"),
    "synth_long": dict(seed=9, S=12_000, P=8, L=70_000, model="pangenome"),
    "synth_uniform": dict(seed=11, S=6_000, P=40, L=3_000, model="uniform"),
    "synth_chrom": dict(seed=13, S=15_000, P=12, L=50_000, model="chromosome"),  # paths along the graph, every other one downwards
}


def run(fn, *args) -> bytes:
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*args)
    return buf.getvalue().encode()


def parse(text: bytes) -> "mygfa.Graph":
    return mygfa.Graph.parse(io.StringIO(text.decode()))


def read_paths(path):
    with open(path) as f:
        return [ln.strip() for ln in f if ln.strip()]


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def main():
    manifest = {}
    for gfa in sorted(glob.glob(os.path.join(HERE, "*.gfa"))):
        base = gfa[:-4]
        name = os.path.basename(base)
        text = open(gfa, "rb").read()
        manifest[name + ".gfa"] = sha(text)
        graph = parse(text)
        out = run(so_depth.depth, graph, None)
        open(base + ".depth.tsv", "wb").write(out)
        manifest[name + ".depth.tsv"] = sha(out)
        if os.path.exists(base + ".subset.paths"):
            out = run(so_depth.depth, graph, read_paths(base + ".subset.paths"))
            open(base + ".depth_subset.tsv", "wb").write(out)
            manifest[name + ".depth_subset.tsv"] = sha(out)
        qpaths = read_paths(base + ".overlap.paths") if os.path.exists(base + ".overlap.paths") \
            else list(graph.paths.keys())
        out = run(so_overlap.overlap, graph, qpaths)
        open(base + ".overlap.tsv", "wb").write(out)
        manifest[name + ".overlap.tsv"] = sha(out)
        print(f"{name}: ok")

    # cfg-S synthetic: text regenerated from the spec, never committed.
    pools = synth.pools(**CFG_S)
    text = synth.gfa_text(pools)
    manifest["synth_cfgS.gfa(not committed)"] = sha(text)
    manifest["synth_cfgS.steps.u32le"] = sha(pools.steps.tobytes())
    graph = parse(text)
    out = run(so_depth.depth, graph, None)
    open(os.path.join(HERE, "synth_cfgS.depth.tsv"), "wb").write(out)
    manifest["synth_cfgS.depth.tsv"] = sha(out)
    print("synth_cfgS: ok")
    for name, cfg in SYNTH_MORE.items():
        pools = synth.pools(**cfg)
        text = synth.gfa_text(pools)
        manifest[name + ".steps.u32le"] = sha(pools.steps.tobytes())
        out = run(so_depth.depth, parse(text), None)
        open(os.path.join(HERE, name + ".depth.tsv"), "wb").write(out)
        manifest[name + ".depth.tsv"] = sha(out)
        print(f"{name}: ok")

    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
