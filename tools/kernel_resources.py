#!/usr/bin/env python3
"""Registers, LDS and scratch of every depth kernel (depth_scan.hip, depth_scan_paths.hip, depth_accum.hip; from the gfx950
assembly's .amdhsa metadata).  Usage: kernel_resources.py [file.hip [flags]]"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [sys.argv[1]] if len(sys.argv) > 1 else [os.path.join(HERE, "..", "pollen_amd", "csrc", n) for n in ("depth_scan.hip", "depth_scan_paths.hip", "depth_accum.hip")]
txt = ""
with tempfile.TemporaryDirectory() as td:
    for src in SRCS:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               src, "-o", asm] + sys.argv[2:], stderr=subprocess.DEVNULL)
        txt += open(asm).read()
names = re.findall(r"\.amdhsa_kernel (\S+)", txt)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for (m, d) in zip(re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S), dem):
    body = m.group(2)

    def g(k):
        mm = re.search(r"\.amdhsa_" + k + r"\s+(\d+)", body)
        return int(mm.group(1)) if mm else None
    d = re.sub(r"fgfa_dev::\(anonymous namespace\)::|\(fgfa_dev.*|void ", "", d)[:64]
    print(f"{d:66s} vgpr={g('next_free_vgpr'):4d} sgpr={g('next_free_sgpr'):4d} lds={g('group_segment_fixed_size'):7d} scratch={g('private_segment_fixed_size')}")
