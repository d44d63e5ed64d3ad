// Synthetic pangenome-shaped FlatGFA generator (readable spec in
// oracle/synth.py, which tests/test_host.py::test_synth_cxx_matches_numpy_spec requires this file to
// match bit-for-bit).  Paths are independent, so they are generated on host threads.
#include <algorithm>
#include <cstdio>
#include <thread>

#include "flatgfa_core.hpp"

namespace fgfa {

namespace {
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void walk(uint64_t seed, uint32_t S, uint32_t p, uint32_t L, int model, Handle *out) {
    uint64_t state = seed * kGolden + p;
    state += kGolden;
    uint64_t cur = mix64(state) % S;
    for (uint32_t t = 0; t < L; ++t) {
        state += kGolden;
        uint64_t r = mix64(state);
        out[t].bits = (uint32_t)(cur << 1) | (uint32_t)((r & 0xFF) < 13);
        uint64_t j = r >> 32;
        if (model == 1) {
            cur = j % S;
        } else if (model == 2 || model == 3 || model == 4) {  // chromosome: along the graph, odd paths downwards with flipped handles
            out[t].bits ^= p & 1u;
            uint64_t u = (r >> 8) % 100, k = (r >> 16) & 0xFF;
            uint64_t d = u < 70 ? 1 : u < 95 ? 2 + (k & 3) : 8 + (k & 63);
            if (model >= 3 && u >= 99) d = (k & 0xF0) ? 1 : 64 + (j & 1023);  // haplotype: no jumps anywhere; one step in 1600 skips up to 1087 segments (a structural variant, not another chromosome)
            d %= S;
            bool ahead = (p & 1u) == 0;
            if (model == 4 && u == 98 && (k & 0xFC) == 0) {  // repeats: a haplotype walk that, one step in 6400, goes 16 .. 271 segments BACK and walks them again (a tandem duplication)
                d = (16 + (j & 255)) % S;
                ahead = !ahead;
            }
            cur = (u < 99 || model >= 3) ? (ahead ? cur + d : cur + S - d) % S : j % S;
        } else {
            uint64_t u = (r >> 8) % 100, k = (r >> 16) & 0xFF;
            if (u < 90) cur += 1;
            else if (u < 95) cur += 2 + (k & 7);
            else if (u < 99) { uint64_t d = 1 + (k & 3); cur = cur >= d ? cur - d : 0; }
            else cur = j % S;
            cur %= S;
        }
    }
}
}  // namespace

void synth_store(uint64_t seed, uint32_t S, uint32_t P, uint32_t L, int model, bool with_seq, Store *st) {
    *st = Store();
    st->segs.resize(S);
    uint64_t key = seed ^ 0xA5A5ull;
    uint64_t off = 0;
    for (uint32_t i = 0; i < S; ++i) {
        uint32_t len = 1 + (uint32_t)(mix64(key + (uint64_t)(i + 1) * kGolden) & 31);
        Segment &s = st->segs[i];
        s.name = (uint64_t)i + 1;
        s.seq.start = (uint32_t)off;
        off += len;
        s.seq.end = (uint32_t)off;
        s.optional = Span{0, 0};
    }
    if (with_seq) {
        st->seq_data.resize(off);
        for (uint32_t i = 0; i < S; ++i) {
            const Segment &s = st->segs[i];
            for (uint32_t k = 0; k < s.seq.len(); ++k) st->seq_data[s.seq.start + k] = (uint8_t)"ACGT"[(i + k) & 3];
        }
    } else {
        // Spans still index a pool of the right size so Segment::len() is meaningful.
        st->seq_data.assign(off, (uint8_t)'N');
    }
    st->paths.resize(P);
    st->steps.resize((size_t)P * L);
    for (uint32_t p = 0; p < P; ++p) {
        char nm[16];
        int n = snprintf(nm, sizeof nm, "p%u", p);
        Path &pa = st->paths[p];
        pa.name.start = (uint32_t)st->name_data.size();
        st->name_data.insert(st->name_data.end(), nm, nm + n);
        pa.name.end = (uint32_t)st->name_data.size();
        pa.steps.start = (uint32_t)((size_t)p * L);
        pa.steps.end = (uint32_t)((size_t)(p + 1) * L);
        pa.overlaps = Span{0, 0};
    }
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    nt = std::min<unsigned>(nt, P ? P : 1);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        th.emplace_back([=] {
            for (uint32_t p = t; p < P; p += nt) walk(seed, S, p, L, model, st->steps.data() + (size_t)p * L);
        });
    }
    for (auto &x : th) x.join();
    st->line_order.assign((size_t)S, kSegment);
    st->line_order.insert(st->line_order.end(), (size_t)P, kPath);
}

}  // namespace fgfa
