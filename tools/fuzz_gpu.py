#!/usr/bin/env python3
"""Randomised differential check of the HIP depth path against the C oracle (run on a GPU box):
random segment counts, path counts and lengths (mixes of short, medium and long paths in one
graph, arbitrary span alignments), both step models, all device configurations.
Usage: python tools/fuzz_gpu.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pollen_amd as pa  # noqa: E402
from oracle import flatgfa_oracle as fo  # noqa: E402

ENVS = [{}, {"FLATGFA_DEPTH_PATH": "bucketed"}, {"FLATGFA_SHORT_MAX": "0"}, {"FLATGFA_BUCKET_CAP": "8"}, {"FLATGFA_PIECE_STEPS": "512"},
        {"FLATGFA_SHORT_MAX": "300"}, {"FLATGFA_DEPTH_PATH": "atomic"}, {"FLATGFA_ACC_PARTS": "5"},
        {"FLATGFA_RANGE_SEGS": "65536"}, {"FLATGFA_RANGE_SEGS": "40960", "FLATGFA_PIECE_STEPS": "2048"},
        {"FLATGFA_ACC_PARTS": "2", "FLATGFA_PIECE_STEPS": "1024"}, {"FLATGFA_DENSE": "1", "FLATGFA_BIG_GROUPS": "1"}, {"FLATGFA_BIG_GROUPS": "1"}, {"FLATGFA_BIG_GROUPS": "0", "FLATGFA_PIECE_STEPS": "700"},
        {"FLATGFA_DENSE": "1", "FLATGFA_RANGE_SEGS": "65536", "FLATGFA_SHORT_MAX": "0"},
        {"FLATGFA_TAGGED": "0"}, {"FLATGFA_TAGGED": "0", "FLATGFA_PIECE_STEPS": "900"}, {"FLATGFA_NO_CLAIM": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_NO_CLAIM": "0", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"}, {"FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_PIECE_STEPS": "4096", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_SCAN_ALWAYS": "1", "FLATGFA_DEPTH_PATH": "bucketed"}, {"FLATGFA_WB": "12", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_WB": "11", "FLATGFA_DEPTH_PATH": "bucketed"}, {"FLATGFA_PATH_GROUPS": "2", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_PATH_GROUPS": "5", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_TAG_LIMIT": "3", "FLATGFA_TAG_MEAN_ONLY": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},  # k_scan's workgroups run out of tags
        {"FLATGFA_TAG_LIMIT": "6", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_NO_PLAIN": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_NO_TINY": "1", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_PACKED": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_PACKED": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed", "FLATGFA_PIECE_STEPS": "3000"},
        {"FLATGFA_PACKED": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed", "FLATGFA_RANGE_SEGS": "65536"},
        {"FLATGFA_ACC_SLOTS": "8", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_ACC_SLOTS": "4", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_WB": "11", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed", "FLATGFA_PIECE_STEPS": "5000"},
        {"FLATGFA_ACC_OWN": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_ACC_OWN": "1", "FLATGFA_ACC_SLOTS": "8", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_ACC_OWN": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed", "FLATGFA_PIECE_STEPS": "3000"},
        {"FLATGFA_ACC_OWN": "1", "FLATGFA_SCAN_WGS": "16", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_ACC_OWN": "1", "FLATGFA_PACKED": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_NO_CLAIM_BLOCKS_MIN": "0", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_NO_CLAIM_BLOCKS_MIN": "0", "FLATGFA_PACKED": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_NO_CLAIM_BLOCKS_MIN": "0", "FLATGFA_PIECE_STEPS": "3000", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_COUNT_PIECES": "3", "FLATGFA_DEPTH_PATH": "bucketed"}, {"FLATGFA_COUNT_PIECES": "17"},  # the plan-time counting kernel takes a path in pieces
        {"FLATGFA_COUNT_PIECES": "2", "FLATGFA_SHORT_MAX": "300", "FLATGFA_DEPTH_PATH": "bucketed"},
        {"FLATGFA_PACKED_ASK": "1", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed"},  # the counting call first, then the layout it speaks for
        {"FLATGFA_PACKED": "1", "FLATGFA_NO_CLAIM_BLOCKS_MIN": "0", "FLATGFA_SHORT_MAX": "0", "FLATGFA_DEPTH_PATH": "bucketed", "FLATGFA_PIECE_STEPS": "2048"}]


def random_graph(rng):
    S = int(rng.choice([1, 7, 33, 1000, 5000, 70_000, 300_000, 1_100_000, 2_200_000, 9_000_000]))  # (the last with FLATGFA_WB=12: more windows than an untagged plan may have)
    kinds = rng.integers(0, 4)
    lens = []
    n_paths = int(rng.integers(1, 400))
    for _ in range(n_paths):
        k = rng.integers(0, 10)
        if k < 5:
            lens.append(int(rng.integers(1, 2100)))        # short
        elif k < 8:
            lens.append(int(rng.integers(2000, 40_000)))   # medium
        else:
            lens.append(int(rng.integers(40_000, 400_000)))  # long
        if sum(lens) > 3_000_000:
            break
    P = len(lens)
    N = sum(lens)
    model = kinds % 2
    steps = np.empty(N, dtype=np.uint32)
    pos = 0
    for L in lens:
        if L < S and rng.integers(0, 4) == 0:  # a strictly monotone walk, up or down (its records skip pass 2's claim); one in three of them with one segment visited twice after all
            gaps = rng.integers(1, (S - 1) // L, size=L, endpoint=True).astype(np.int64)  # (strictly increasing, the last id below S)
            ids = np.cumsum(gaps) - gaps[0]
            if rng.integers(0, 2):
                ids = ids[::-1].copy()
            if rng.integers(0, 3) == 0 and L > 2:
                k = int(rng.integers(1, L))
                ids[k] = ids[k - 1]
        elif model == 0:  # locally monotone with jumps (wraps around S)
            u = rng.integers(0, 100, size=L)
            j = rng.integers(0, 1 << 30, size=L)
            inc = np.where(u < 90, 1, np.where(u < 95, 2 + (j & 7), np.where(u < 99, -(1 + (j & 3)), 0))).astype(np.int64)
            if rng.integers(0, 3) == 0:
                inc = -inc  # a walk down the segment ids (k_scan's step -1 runs)
            jump = u >= 99
            jump[0] = True
            base = np.where(jump, j % S, 0).astype(np.int64)
            inc[jump] = 0
            c = np.cumsum(inc)
            last = np.maximum.accumulate(np.where(jump, np.arange(L), 0))
            ids = (base[last] + c - c[last]) % S
        else:
            ids = rng.integers(0, S, size=L)
        steps[pos:pos + L] = (ids.astype(np.uint32) << 1) | rng.integers(0, 2, size=L).astype(np.uint32)
        pos += L
    # spans: contiguous, or with gaps of unused steps that hold garbage handles (the type allows
    # arbitrary spans; nothing outside a span may be looked at), plus a few duplicated spans
    begins = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    if kinds >= 2:
        gaps = rng.integers(0, 41, size=P)
        shift = np.cumsum(gaps)
        out = np.full(N + int(shift[-1]) + int(rng.integers(0, 20)), 0xFFFFFFFF, dtype=np.uint32)
        for p in range(P):
            out[begins[p] + shift[p]: begins[p] + shift[p] + lens[p]] = steps[begins[p]: begins[p] + lens[p]]
        steps = out
        begins = begins + shift
    ends = begins + np.array(lens, dtype=np.int64)
    if P > 2 and kinds == 3:  # the same span twice: two paths over the same steps
        k = int(rng.integers(0, P))
        begins = np.append(begins, begins[k])
        ends = np.append(ends, ends[k])
        P += 1
    return S, P, steps, begins.astype(np.uint32), ends.astype(np.uint32)


ENV_KEYS = sorted({k for e in ENVS for k in e} | {"FLATGFA_ACC_PAIR", "FLATGFA_ACC_SMALL"})  # every variable a case may set: cleared before the next one


def run_case(rng, case, verbose=True):
    """One random graph through the device configuration `case` selects; True when node depth, unique depth, depth-only,
    path sums of a subset and path depth of all paths all equal the oracle's, bit for bit.  Sets FLATGFA_* variables of
    the process (callers that care restore them: tests/test_gpu_depth.py::test_fuzz_slice)."""
    from pollen_amd import device as dev
    import torch
    S, P, steps, pb, pe = random_graph(rng)
    seg_len = rng.integers(1, 40, size=S).astype(np.uint32)
    # oracle on raw arrays
    paths = np.zeros(P, dtype=fo.PATH_DT)
    paths["steps_start"], paths["steps_end"] = pb, pe
    segs = np.zeros(S, dtype=fo.SEG_DT)
    segs["seq_end"] = seg_len  # only the length matters to path depth
    pools = fo.Pools(**{n: np.zeros(0, dtype=np.uint8) for n in fo.POOL_ORDER})
    pools.paths, pools.steps, pools.segs = paths, steps, segs
    want_d, want_u = fo.seg_depth_with_uniq(pools)
    env = ENVS[case % len(ENVS)]
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    graph = dev.DeviceGraph(steps, pb, pe, S, seg_len, device="cuda:0")
    plan = dev.DepthPlan(graph)
    d = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    u = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    d2 = torch.zeros(S, dtype=torch.int32, device="cuda:0")
    for _ in range(2):  # twice: the scratch must be clean again
        plan.seg_depth(d, u)
        plan.status()   # (a forced 8-record capacity: the call is only complete after this)
        plan.seg_depth(d2, None)
        plan.status()
    # path depth of a strided subset of the paths: integer sums on the device, one f64 division here
    ids = np.arange(P - 1, -1, -3, dtype=np.uint32)
    t_ids = torch.from_numpy(ids.view(np.int32)).to("cuda:0")
    ln = torch.zeros(len(ids), dtype=torch.int64, device="cuda:0")
    ws = torch.zeros(len(ids), dtype=torch.int64, device="cuda:0")
    plan.path_sums(t_ids, d2, ln, ws)
    plan.status()
    want_ln, want_mean = fo.path_depth(pools, ids)
    got_ln = ln.cpu().numpy().view(np.uint64)
    with np.errstate(divide="ignore", invalid="ignore"):
        got_mean = ws.cpu().numpy().view(np.uint64).astype(np.float64) / got_ln.astype(np.float64)
    # ... and of all paths in the call that also counts node depth (`fgfa depth`); the outputs start as garbage
    d3 = torch.full((S,), -7, dtype=torch.int32, device="cuda:0")
    ln_all = torch.full((P,), 12345, dtype=torch.int64, device="cuda:0")
    ws_all = torch.full((P,), -1, dtype=torch.int64, device="cuda:0")
    for _ in range(2):
        plan.path_depth_all(d3, ln_all, ws_all)
        plan.status()
    want_ln_all, want_mean_all = fo.path_depth(pools, np.arange(P, dtype=np.uint32))
    got_ln_all = ln_all.cpu().numpy().view(np.uint64)
    with np.errstate(divide="ignore", invalid="ignore"):
        got_mean_all = ws_all.cpu().numpy().view(np.uint64).astype(np.float64) / got_ln_all.astype(np.float64)
    gd, gu, gd2, gd3 = (t.cpu().numpy().view(np.uint32) for t in (d, u, d2, d3))
    ok = bool((gd == want_d).all() and (gu == want_u).all() and (gd2 == want_d).all() and (gd3 == want_d).all()
              and (got_ln == want_ln).all() and got_mean.tobytes() == want_mean.tobytes()
              and (got_ln_all == want_ln_all).all() and got_mean_all.tobytes() == want_mean_all.tobytes())
    what = f"case {case}: S={S} P={P} N={len(steps)} env={env} [{plan.describe()[:60]}]"
    if verbose:
        print(f"{what} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            print("   depth bad:", int((gd != want_d).sum()), "uniq bad:", int((gu != want_u).sum()), "depth-only bad:", int((gd2 != want_d).sum()))
    plan.close()
    return ok, what


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        ok, _ = run_case(rng, case)
        bad += 0 if ok else 1
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
