#!/usr/bin/env python3
"""Build-time check for the pinned landing registers of the depth kernels.

k_scan keeps three 1024-step blocks in flight in the fixed VGPR sets v[80:95], v[96:111] and v[112:127]
(see the comment above load_block_async in pollen_amd/csrc/depth_fast_kernels.hpp).  That is only sound if
nothing else in the kernel touches those registers.  This script compiles the kernels' translation units
(depth_scan.hip, depth_scan_paths.hip, depth_accum.hip) to gfx950 assembly and checks, for every k_scan
instantiation and every function it can call:

  * the only instructions that mention v80..v127 are `global_load_dwordx4 v[Q:Q+3], ..., off`
    (as the destination) and `v_lshrrev_b32 vN, 1, vQ` (as the source);
  * the kernel's VGPR budget stays at or under 128 (a 1024-thread workgroup needs 4 waves/SIMD).

Usage: check_pinned_vgprs.py [depth_scan.hip depth_scan_paths.hip depth_accum.hip]      exit status 0 = ok
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "pollen_amd", "csrc")
SRCS = sys.argv[1:] if len(sys.argv) > 1 else [os.path.join(CSRC, n) for n in ("depth_scan.hip", "depth_scan_paths.hip", "depth_accum.hip")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

PINNED = set(range(80, 128))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
OK_LOAD = re.compile(r"^global_load_dwordx4 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], off( offset:(16|32|48|1024|2048|3072))?( (nt|sc0|sc1))*$")
OK_TAKE = re.compile(r"^v_lshrrev_b32(_e32)? v(\d+), 1, v(\d+)$")
OK_TAKE_F = re.compile(r"^v_alignbit_b32 v(\d+), s\d+, v(\d+), 1$")  # (take_block_flagged: a uniform bit on top of the id)


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def main():
    lines, text = [], ""
    import shlex
    extra = shlex.split(os.environ.get("FGFA_CXXFLAGS", ""))  # (the -D flags of a variant build, tools/variants.sh; quoted as for a shell)
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for k, src in enumerate(SRCS):
            asm = os.path.join(td, f"tu{k}.s")
            procs.append((asm, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                                                 src, "-o", asm] + extra, stderr=subprocess.DEVNULL)))
            text += open(src).read()
        for asm, pr in procs:
            if pr.wait() != 0:
                print("pinned-VGPR check FAILED: hipcc -S failed")
                return 1
            lines += open(asm).read().splitlines()
    # the tagged walk's depth: FGFA_TAG_DEPTH steps are requested ahead, into v(123 - depth) .. v122
    m_depth = re.search(r"#define FGFA_TAG_DEPTH (\d+)", text)  # (depth_accum.hip; absent when only other units are checked)
    depth = int(m_depth.group(1)) if m_depth else 3
    for f in extra:
        if f.startswith("-DFGFA_TAG_DEPTH="):
            depth = int(f.split("=")[1])
    tagged_pins = set(range(123 - depth, 123))
    bad, n_load, n_take, func = [], 0, 0, "?"
    n_acc_load = n_acc_take = 0
    n_tiny_load = n_tiny_take = 0
    n_dense_load = n_dense_take = 0
    # k_scan_dense waits for its landing loads with `s_waitcnt vmcnt(8)`: sound only if the eight record stores of a
    # plain tile are eight store INSTRUCTIONS.  Checked on the ISA: behind the kernel's first barrier some straight-line
    # stretch holds exactly eight global_store_dword and no other vector-memory operation (merged or dropped stores
    # would leave none).
    dense_seen_barrier, dense_blk_stores, dense_blk_other, dense_blocks8, dense_wait8 = False, 0, 0, 0, 0
    budgets = {}
    for ln in lines:
        s = ln.split(";")[0].strip()
        if s.endswith(":") and not s.startswith("."):
            func = s[:-1]
            continue
        if "k_scan_dense" in func and re.match(r"^\.LBB\d+_\d+:", s):  # (a label ends a straight-line stretch)
            if dense_blk_stores == 8 and dense_blk_other == 0:
                dense_blocks8 += 1
            dense_blk_stores = dense_blk_other = 0
        m = re.match(r"\.amdhsa_next_free_vgpr\s+(\d+)", s)
        if m:
            budgets[func] = int(m.group(1))
            continue
        if not s or s.startswith("."):
            continue
        if "k_accum" in func:  # pass 2: pinned record registers -- v120..v122 (rec_request / rec_take), and in the
            # build with two workgroups per CU (k_accum_pair, 64 registers) v61..v63 (rec_request_lo / rec_take_lo)
            tagged = re.search(r"k_accumILb\dELi\d+ELb\dELb\dELb\dELb1ELi\d+EE", func) is not None  # (the template argument before the last: tagged)
            pins = {61, 62, 63} if ("k_accum_pair" in func or "k_accum_small" in func) else tagged_pins if tagged else {120, 121, 122}  # (the build with 64 registers lands its records in v61..v63)
            if regs_of(s) & pins:
                m = re.match(r"^global_load_dword v(\d+), (v\[\d+:\d+\], off|v\d+, s\[\d+:\d+\])$", s)
                t = re.match(r"^v_mov_b32(_e32)? v(\d+), v(\d+)$", s)
                if m and int(m.group(1)) in pins:
                    n_acc_load += 1
                elif t and int(t.group(3)) in pins and int(t.group(2)) not in pins:
                    n_acc_take += 1
                else:
                    bad.append(f"{func}: {s}")
            continue
        if "k_scan_tiny" in func:  # three paths' steps in flight: v118..v123 (tiny_request / tiny_take)
            pins = set(range(118, 124))
            if regs_of(s) & pins:
                m = re.match(r"^global_load_dword v(\d+), v\d+, s\[\d+:\d+\]$", s)
                t = re.match(r"^v_mov_b32(_e32)? v(\d+), v(\d+)$", s)
                if m and int(m.group(1)) in pins:
                    n_tiny_load += 1
                elif t and int(t.group(3)) in pins and int(t.group(2)) not in pins:
                    n_tiny_take += 1
                else:
                    bad.append(f"{func}: {s}")
            continue
        if "k_scan_dense" in func:  # a full tile's steps land in v112..v119 (two dwordx4 a thread), taken out by the shifts that drop the orientation bit
            if s.startswith("s_barrier"):
                dense_seen_barrier = True
            if s.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier")):
                if dense_blk_stores == 8 and dense_blk_other == 0:
                    dense_blocks8 += 1
                dense_blk_stores = dense_blk_other = 0
            elif s.startswith("global_store_dword ") and dense_seen_barrier:
                dense_blk_stores += 1
            elif s.startswith(("global_", "buffer_", "flat_", "scratch_")) and dense_seen_barrier:
                dense_blk_other += 1
            if s == "s_waitcnt vmcnt(8)":
                dense_wait8 += 1
            pins = set(range(112, 120))
            if regs_of(s) & pins:
                m = re.match(r"^global_load_dwordx4 v\[(112:115|116:119)\], v\[(\d+):(\d+)\], off nt$", s)
                t = OK_TAKE.match(s)
                if m and int(m.group(2)) not in pins and int(m.group(3)) not in pins:
                    n_dense_load += 1
                elif t and int(t.group(3)) in pins and int(t.group(2)) not in pins:
                    n_dense_take += 1
                else:
                    bad.append(f"{func}: {s}")
            continue
        if "k_scan" not in func:  # only the kernels that use the landing sets (everything is inlined into them)
            continue
        pinned = set(range(96, 128)) if "k_scan_short" in func else PINNED  # (the wave-per-path kernels land their blocks in v96..v127 only)
        touched = regs_of(s) & pinned
        if not touched:
            continue
        m = OK_LOAD.match(s)
        if m and int(m.group(1)) in pinned and int(m.group(1)) % 4 == 0 and int(m.group(2)) == int(m.group(1)) + 3 \
                and int(m.group(3)) not in pinned and int(m.group(4)) not in pinned:
            n_load += 1
            continue
        m = OK_TAKE.match(s)
        if m and int(m.group(3)) in pinned and int(m.group(2)) not in pinned:
            n_take += 1
            continue
        m = OK_TAKE_F.match(s)
        if m and int(m.group(2)) in pinned and int(m.group(1)) not in pinned:
            n_take += 1
            continue
        bad.append(f"{func}: {s}")
    for name, v in budgets.items():
        if "k_scan" in name and v > 128:
            bad.append(f"{name}: {v} VGPRs > 128")
    if n_acc_load == 0 or n_acc_take == 0:
        bad.append("no pinned record loads/takes found in k_accum -- did the kernel change?")
    if n_tiny_load == 0 or n_tiny_take == 0:
        bad.append("no pinned loads/takes found in k_scan_tiny -- did the kernel change?")
    if n_dense_load == 0 or n_dense_take == 0:
        bad.append("no pinned loads/takes found in k_scan_dense -- did the kernel change?")
    if n_load == 0 or n_take == 0:
        bad.append("no pinned loads/takes found -- did the kernel change?")
    if dense_wait8 == 0 or dense_blocks8 == 0:  # (merged or dropped record stores would leave no stretch of exactly eight)
        bad.append(f"k_scan_dense: vmcnt(8) waits {dense_wait8}, straight-line stretches of exactly eight record stores {dense_blocks8} "
                   "-- the counted wait no longer matches the stores")
    if bad:
        print("pinned-VGPR check FAILED (%d lines; kernels: %s):\n  " % (len(bad), sorted({b.split(":")[0][-40:] for b in bad})) + "\n  ".join(bad[:20]))
        return 1
    print(f"pinned-VGPR check ok: k_accum {n_acc_load} record loads, {n_acc_take} takes, nothing else touches v120..v122 (tagged walk: v{123 - depth}..v122; measurement builds' k_accum_pair / k_accum_small: v61..v63); "
          f"k_scan_tiny {n_tiny_load} loads, {n_tiny_take} takes (v118..v123); k_scan_dense {n_dense_load} loads, {n_dense_take} takes (v112..v119), {dense_blocks8} stretches of eight record stores for {dense_wait8} vmcnt(8); k_scan {n_load} loads, {n_take} takes, nothing else touches v80..v127; "
          f"k_scan budgets {sorted(set(v for k, v in budgets.items() if 'k_scan' in k))}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
