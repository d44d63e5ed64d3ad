"""The C ABI driven from C, not ctypes: tests/c_abi/example.c is compiled against include/flatgfa.h
and linked with libflatgfa.so the way flatgfa-c/Makefile:1-9 builds the reference's example."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT
from oracle import flatgfa_oracle as fo

LIBDIR = os.path.join(ROOT, "pollen_amd", "lib")


def build(tmp_path):
    cc = shutil.which("cc") or shutil.which("gcc")
    assert cc, "no C compiler"
    exe = str(tmp_path / "example")
    subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "example.c"), "-L", LIBDIR, "-lflatgfa",
                    f"-Wl,-rpath,{LIBDIR}", "-o", exe], check=True)
    return exe


def expected_walk(pools):
    out = [f"segments {len(pools.segs)} paths {len(pools.paths)}"]
    for i in range(len(pools.paths)):
        b, e = int(pools.paths["steps_start"][i]), int(pools.paths["steps_end"][i])
        out.append(f"path {pools.path_name(i).decode()}: {e - b} steps")
        for h in pools.steps[b:e]:
            sg = pools.segs[int(h) >> 1]
            seq = bytes(pools.seq_data[int(sg["seq_start"]):int(sg["seq_end"])]).decode()
            out.append(f"  {'-' if int(h) & 1 else '+'} {seq}")
    return "\n".join(out) + "\n"


@pytest.mark.parametrize("name", ["ref_ex2", "edge_names_loops", "kat_window_depth"])
def test_c_example_walks_paths_and_steps(name, tmp_path):
    exe = build(tmp_path)
    gfa = os.path.join(GOLDEN, name + ".gfa")
    r = subprocess.run([exe, gfa, "--no-depth"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == expected_walk(fo.parse_gfa(open(gfa, "rb").read()))


@pytest.mark.gpu
def test_c_example_node_depth(tmp_path):
    exe = build(tmp_path)
    gfa = os.path.join(GOLDEN, "ref_ex2.gfa")
    pools = fo.parse_gfa(open(gfa, "rb").read())
    r = subprocess.run([exe, gfa], capture_output=True)
    assert r.returncode == 0, r.stderr
    d, _ = fo.seg_depth_with_uniq(pools)
    want = expected_walk(pools).encode() + open(gfa[:-4] + ".depth.tsv", "rb").read() + f"total depth {int(d.sum())}\nsharded two ways: same vectors\n".encode()
    assert r.stdout == want
