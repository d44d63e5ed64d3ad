"""synth -- TEST INFRASTRUCTURE ONLY: the numpy statement of the synthetic-graph spec.

The product library carries the same generator in C++ (``flatgfa_synth``,
pollen_amd/csrc/synth.cpp) so that bench.py can build the 1M-segment /
100M-step graph in seconds; tests/test_synth.py requires the two to agree
bit-for-bit.  This file is the readable definition:

  GOLDEN = 0x9E3779B97F4A7C15
  mix64(z): z=(z^(z>>30))*0xBF58476D1CE4E5B9; z=(z^(z>>27))*0x94D049BB133111EB; z^(z>>31)
  splitmix64: state += GOLDEN; return mix64(state)
  segment i (0-based): name i+1, len 1 + (mix64((seed^0xA5A5) + (i+1)*GOLDEN) & 31),
      bases "ACGT"[(i+k)&3] for k in 0..len
  path p: name "p{p}", rng state = seed*GOLDEN + p, L steps
      r = next(); cur = r % S
      per step: r = next(); emit handle (cur<<1) | ((r & 0xFF) < 13)
                u = (r>>8) % 100; k = (r>>16) & 0xFF; j = r>>32
                pangenome: u<90: cur+1 | u<95: cur+2+(k&7) | u<99: max(cur-1-(k&3), 0) | else: j%S
                uniform:   cur = j % S
                chromosome (paths walk along the graph; odd paths downwards, handles flipped):
                           d = u<70: 1 | u<95: 2+(k&3) | u<99: 8+(k&63) | else: jump to j%S
                           even p: cur+d, odd p: cur-d (mod S); handle orientation bit ^= p&1
                haplotype: as chromosome, but u>=99 does not jump anywhere: d = 1, or, when k & 0xF0 == 0
                           (one step in 1600), d = 64 + (j & 1023) -- a structural variant, not another chromosome
                cur %= S
  steps of path p occupy steps[p*L:(p+1)*L]  (contiguous, in order: parse.rs:149-159)
"""
from __future__ import annotations

import numpy as np

from .flatgfa_oracle import PATH_DT, SEG_DT, Pools

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)


def mix64(z: np.ndarray) -> np.ndarray:
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        return z ^ (z >> np.uint64(31))


def seg_lens(seed: int, S: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        i = np.arange(1, S + 1, dtype=np.uint64)
        key = np.uint64((seed ^ 0xA5A5) & 0xFFFFFFFFFFFFFFFF)
        return (np.uint64(1) + (mix64(key + i * GOLDEN) & np.uint64(31))).astype(np.uint32)


def steps(seed: int, S: int, P: int, L: int, model: str = "pangenome") -> np.ndarray:
    """Returns the flat u32 handle array, shape (P*L,)."""
    assert model in ("pangenome", "uniform", "chromosome", "haplotype", "repeats")
    out = np.zeros((P, L), dtype=np.uint32)
    with np.errstate(over="ignore"):
        state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * GOLDEN + np.arange(P, dtype=np.uint64)
        state = state + GOLDEN
        cur = mix64(state) % np.uint64(S)
        for t in range(L):
            state = state + GOLDEN
            r = mix64(state)
            back = ((r & np.uint64(0xFF)) < np.uint64(13)).astype(np.uint64)
            out[:, t] = ((cur << np.uint64(1)) | back).astype(np.uint32)
            j = r >> np.uint64(32)
            if model == "uniform":
                cur = j % np.uint64(S)
            elif model in ("chromosome", "haplotype", "repeats"):
                odd = np.arange(P, dtype=np.uint64) & np.uint64(1)
                out[:, t] ^= odd.astype(np.uint32)
                u = (r >> np.uint64(8)) % np.uint64(100)
                k = (r >> np.uint64(16)) & np.uint64(0xFF)
                d = np.where(u < 70, np.uint64(1), np.where(u < 95, np.uint64(2) + (k & np.uint64(3)), np.uint64(8) + (k & np.uint64(63))))
                if model in ("haplotype", "repeats"):
                    d = np.where(u >= 99, np.where((k & np.uint64(0xF0)) != 0, np.uint64(1), np.uint64(64) + (j & np.uint64(1023))), d)
                d = d % np.uint64(S)
                ahead = odd == 0
                if model == "repeats":  # one step in 6400 goes 16 .. 271 segments back along the walk and walks them again (a tandem duplication)
                    rep = (u == 98) & ((k & np.uint64(0xFC)) == 0)
                    d = np.where(rep, (np.uint64(16) + (j & np.uint64(255))) % np.uint64(S), d)
                    ahead = np.where(rep, ~ahead, ahead)
                moved = np.where(ahead, cur + d, cur + np.uint64(S) - d) % np.uint64(S)
                cur = moved if model in ("haplotype", "repeats") else np.where(u < 99, moved, j % np.uint64(S))
            else:
                u = (r >> np.uint64(8)) % np.uint64(100)
                k = (r >> np.uint64(16)) & np.uint64(0xFF)
                fwd1 = cur + np.uint64(1)
                fwdk = cur + np.uint64(2) + (k & np.uint64(7))
                d = np.uint64(1) + (k & np.uint64(3))
                backk = np.where(cur >= d, cur - d, np.uint64(0))
                jump = j % np.uint64(S)
                cur = np.where(u < 90, fwd1, np.where(u < 95, fwdk, np.where(u < 99, backk, jump)))
                cur = cur % np.uint64(S)
    return out.reshape(-1)


def pools(seed: int, S: int, P: int, L: int, model: str = "pangenome") -> Pools:
    """The full FlatGFA the spec describes (no header, no links, line order S..., P...)."""
    lens = seg_lens(seed, S)
    ends = np.cumsum(lens.astype(np.uint64))
    starts = ends - lens
    segs = np.zeros(S, dtype=SEG_DT)
    segs["name"] = np.arange(1, S + 1, dtype=np.uint64)
    segs["seq_start"] = starts.astype(np.uint32)
    segs["seq_end"] = ends.astype(np.uint32)
    total = int(ends[-1]) if S else 0
    seg_of = np.repeat(np.arange(S, dtype=np.uint64), lens)
    k = np.arange(total, dtype=np.uint64) - np.repeat(starts, lens)
    seq = np.frombuffer(b"ACGT", dtype="u1")[((seg_of + k) & np.uint64(3)).astype(np.intp)]
    names = b"".join(b"p%d" % p for p in range(P))
    name_ends = np.cumsum([len(b"p%d" % p) for p in range(P)]).astype(np.uint32) if P else np.zeros(0, np.uint32)
    paths = np.zeros(P, dtype=PATH_DT)
    paths["name_end"] = name_ends
    paths["name_start"][1:] = name_ends[:-1]
    paths["steps_start"] = np.arange(P, dtype=np.uint64) * L
    paths["steps_end"] = (np.arange(P, dtype=np.uint64) + 1) * L
    e8 = np.zeros(0, dtype="u1")
    from .flatgfa_oracle import LINK_DT, SPAN_DT
    return Pools(header=e8, segs=segs, paths=paths, links=np.zeros(0, LINK_DT),
                 steps=steps(seed, S, P, L, model), seq_data=seq.copy(),
                 overlaps=np.zeros(0, SPAN_DT), alignment=np.zeros(0, "<u4"),
                 name_data=np.frombuffer(names, dtype="u1").copy(), optional_data=e8,
                 line_order=np.concatenate([np.full(S, 1, "u1"), np.full(P, 2, "u1")]))


def gfa_text(p: Pools) -> bytes:
    """GFA text for a pools object with no header/links/overlaps (S lines then P lines)."""
    out = []
    seq = p.seq_data.tobytes()
    for s in p.segs:
        out.append(b"S\t%d\t%s\n" % (int(s["name"]), seq[int(s["seq_start"]):int(s["seq_end"])]))
    names = p.segs["name"]
    for i in range(len(p.paths)):
        pa = p.paths[i]
        hs = p.steps[int(pa["steps_start"]):int(pa["steps_end"])]
        nm = names[(hs >> 1).astype(np.intp)]
        toks = [b"%d%s" % (int(n), b"-" if (int(h) & 1) else b"+") for n, h in zip(nm, hs)]
        out.append(b"P\t%s\t%s\t*\n" % (p.path_name(i), b",".join(toks)))
    return b"".join(out)
