#!/usr/bin/env python3
"""Per-basic-block instruction mix of one depth kernel's gfx950 assembly (k_scan*: depth_scan.hip / depth_scan_paths.hip,
k_accum / k_path_reduce: depth_accum.hip).
Usage: isa_blocks.py [substring of the mangled kernel name, default k_scanILi0ELb1E] [file.s]"""
import re, subprocess, sys, os, tempfile
want = sys.argv[1] if len(sys.argv) > 1 else "k_scanILi0ELb1E"
asm = sys.argv[2] if len(sys.argv) > 2 else None
if asm is None:
    here = os.path.dirname(os.path.abspath(__file__))
    asm = os.path.join(tempfile.gettempdir(), "depth_fast_isa.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           os.path.join(here, "..", "pollen_amd", "csrc", "depth_accum.hip" if ("k_accum" in want or "k_path_reduce" in want) else
                                        "depth_scan_paths.hip" if any(k in want for k in ("k_scan_short", "k_scan_tiny")) else "depth_scan.hip"),
                           "-o", asm], stderr=subprocess.DEVNULL)
lines = open(asm).read().splitlines()
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r"^_Z\S*" + re.escape(want) + r"\S*:", l):
        start = i
    elif start is not None and "s_endpgm" in l:
        end = i
        break
blocks, cur = [], ("entry", [])
for l in lines[start + 1:end]:
    s = l.split(";")[0].strip()
    if not s:
        continue
    if s.endswith(":"):
        blocks.append(cur)
        cur = (s[:-1], [])
    elif not s.startswith("."):
        cur[1].append(s)
blocks.append(cur)
def cls(op):
    for p, c in (("v_", "V"), ("s_", "S"), ("ds_", "L"), ("global_", "M"), ("buffer_", "M"), ("scratch_", "X")):
        if op.startswith(p):
            return c
    return "?"
for name, ins in blocks:
    c = {}
    for x in ins:
        k = cls(x.split()[0])
        c[k] = c.get(k, 0) + 1
    br = [x.split()[0][2:] + ">" + x.split()[1] for x in ins if x.startswith(("s_cbranch", "s_branch"))]
    print(f"{name:10s} n={len(ins):4d} V={c.get('V',0):3d} S={c.get('S',0):3d} L={c.get('L',0):2d} M={c.get('M',0):2d} X={c.get('X',0)} {' '.join(br)}")
