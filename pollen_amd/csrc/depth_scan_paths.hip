// Pass 1 for short paths (seg_depth_with_uniq / seg_depth, ops/depth.rs:15-56): every wave walks whole paths on
// its own.  k_scan_short (paths of at most 2048 steps), its medium build (longer paths with few runs, two waves
// per path), k_scan_tiny (paths a wave holds whole).  See depth_fast.hip for the path as a whole.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "depth_fast_kernels.hpp"

namespace fgfa_dev {
namespace {

// ============================================================ pass 1, wave per path ===
//
// k_scan_short: every wave walks whole (short or medium) paths on its own.  Blocks as in k_scan,
// but runs are cut at bitset-word boundaries and claimed, when emitted, in a per-wave hash set of
// bitset words (open addressing, keyed by word index + 1); the records say what they count for
// (bit 24: depth, bit 25: uniq), so pass 2 applies them without any claim.

// Per-wave state: the run queue (LDS, kQCap entries of (start id << 11) | (len - 1)), how many
// entries it holds, and the queue of partially new claims (see emit_chunk).  `fill`, `pfill` and
// `vm` are wave-uniform.
struct Wave {
    uint32_t *q, *pq;
    uint32_t *dummy;  // 64 (key, bits) pairs no path's words hash to: where lanes without a run probe (claim_hashed)
    uint32_t qcap;    // entries the run queue holds (uniform)
    uint32_t fill, pfill;
    uint32_t vm[3];  // memory instructions issued since the loads into landing set 0 / 1 / 2 (see wait_block)
    int lane;
};

#ifndef FGFA_SHORT_ABLATE
#define FGFA_SHORT_ABLATE 0  /* measurements only (results are wrong): 1 loads only, 2 runs queued but not emitted, 4 no claims, 8 no record stores, 16 hash set not wiped, 32 partly new claims dropped */
#endif

__device__ __forceinline__ uint32_t clamp_id(const ScanArgs &A, uint32_t id) {
    if (id >= A.n_segs) {
        atomicOr(A.status, kStBounds);
        return 0u;
    }
    return id;
}

__device__ __forceinline__ void push_partial(Wave &w, bool e, uint32_t word, uint32_t bits) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(e);
    if (e) reinterpret_cast<uint2 *>(w.pq)[w.pfill + lane_rank(m)] = make_uint2(word, bits);
    w.pfill += (uint32_t)__builtin_popcountll(m);
}

// The path's "seen" words live in a small per-wave hash set instead of a bitset over all
// segments.  The plan only sends a path here when its runs fit the set.
template <int BITS>
__device__ __forceinline__ uint32_t claim_hashed(const ScanArgs &A, uint32_t *tab, uint32_t *dummy, int lane, bool valid, uint32_t word, uint32_t mask) {
    // The first probe -- nearly always the last -- is taken by all lanes with nothing predicated: a lane without a
    // run probes an entry of its own in `dummy` (64 pairs behind the hash sets), a lane whose probe finds another
    // word's entry ORs nothing.  (As a loop with the probes under `if (todo)` hipcc spent eighty scalar
    // instructions per chunk on exec masks.)
    uint32_t h = (word * 0x9E3779B1u) >> (32 - BITS);
    const uint32_t key = word + 1u;
    uint32_t *e = valid ? tab + 2u * h : dummy + 2u * (uint32_t)lane;
    uint32_t k = atomicCAS(e, 0u, key);
    bool ok = k == 0u || k == key;
    uint32_t old = atomicOr(e + 1, ok ? mask : 0u);
    bool todo = valid && !ok;
    uint32_t probes = 1;
    while (__builtin_amdgcn_ballot_w64(todo)) {
        if (++probes > (1u << BITS)) {  // cannot happen while the plan matches the steps: the set would be full
            atomicOr(A.status, kStBounds);
            break;
        }
        h = (h + 1u) & ((1u << BITS) - 1u);
        if (todo) {
            e = tab + 2u * h;
            k = atomicCAS(e, 0u, key);
            if (k == 0u || k == key) {
                old = atomicOr(e + 1, mask);
                todo = false;
            }
        }
    }
    return old;
}

// Emit up to 64 queued runs, one per lane.  Each run becomes one depth record.  For unique
// depth the lane claims the run's segments with ONE returning OR (runs are cut at multiples of
// 32, so a run lies inside one word): the bits that were still clear are exactly the (path,
// segment) pairs this run is the first to touch.  If all of them were clear the depth record
// doubles as the uniq record (kind 2), if none was there is nothing to add.  The rare claim that
// is partly new is parked, as (word index, new bits), on a second queue; that queue is turned
// into uniq records 32..64 entries at a time, so its bit-stretch loop runs with most lanes busy.
template <bool UNIQ, int HASH>
__device__ __forceinline__ void emit_chunk(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine,
                                           bool valid, uint32_t ent, uint32_t next, bool mono) {
    // a run lasts until the next entry's position (positions start over with every block: modulo 1024)
    const uint32_t id = ent >> kPosBits, lenm1 = (next - ent - 1u) & ((1u << kPosBits) - 1u), win = id >> kShortWinBits;
    valid = valid && id < kDummyBase;  // runs of placeholder ids and the entries that only close a run are dropped here
    uint32_t kind = 0, pos;
    if (UNIQ) {
        const uint32_t mask = valid ? (0xFFFFFFFFu >> (31u - lenm1)) << (id & 31u) : 0u;
        // (mono, uniform: the path never meets a segment twice -- every run is all first visits, and the set stays empty)
        const uint32_t old = ((FGFA_SHORT_ABLATE & 4) || mono) ? 0u : claim_hashed<HASH>(A, seen, w.dummy, w.lane, valid, id >> 5, mask);
        pos = take_slots(bcur, w.lane, valid, win);
        const uint32_t nb = mask & ~old;
        kind = (nb == mask) ? 2u : 0u;
        push_partial(w, (nb != mask) & (nb != 0u) & !(FGFA_SHORT_ABLATE & 32), id >> 5, nb);
    } else {
        pos = take_slots(bcur, w.lane, valid, win);
    }
    const uint32_t word = (id & ((1u << kShortWinBits) - 1u)) | (lenm1 << kShortWinBits) | ((kind + 1u) << 24);  // bit 24: counts for depth, bit 25: for uniq
    if (!(FGFA_SHORT_ABLATE & 8)) flag_if_any(A, put<false>(A, w, mine, valid, pos, win, word), kStOverflow);
    else if (word == 0xDEADBEEFu && pos == 77u) atomicOr(A.status, kStBounds);
}

// Turn parked claims into uniq records, one per stretch of new bits: the newest 64 while at
// least 32 are parked (all of them when `all`).
__device__ __forceinline__ void drain_partial(const ScanArgs &A, Wave &w, uint32_t *bcur, uint32_t *mine, bool all) {
    while (w.pfill >= 32u || (all && w.pfill)) {
        const uint32_t n = min(w.pfill, 64u);
        w.pfill -= n;
        const bool valid = (uint32_t)w.lane < n;
        const uint2 ent = valid ? reinterpret_cast<const uint2 *>(w.pq)[w.pfill + w.lane] : make_uint2(0u, 0u);
        const uint32_t base = ent.x << 5;
        const uint32_t win = base >> kShortWinBits;
        uint32_t m = ent.y;  // never all ones: that claim would have been entirely new
        while (__builtin_amdgcn_ballot_w64(m != 0u)) {
            const bool e = m != 0u;
            const uint32_t tz = e ? (uint32_t)__builtin_ctz(m) : 0u;
            const uint32_t run = (uint32_t)__builtin_ctz(~(m >> tz));
            m &= ~(((1u << run) - 1u) << tz);
            const uint32_t p = e ? atomicAdd(&bcur[win], 1u) : 0u;
            const uint32_t word = ((base + tz) & ((1u << kShortWinBits) - 1u)) | ((run - 1u) << kShortWinBits) | (2u << 24);
            flag_if_any(A, put<false>(A, w, mine, e, p, win, word), kStOverflow);
        }
    }
}

// Emit the oldest entries, 64 at a time, while at least 65 are queued (an entry needs the one behind it: that is
// where its run ends), then move what is left to the front of the queue.  With `all` the newest entry closes a
// block, and everything is emitted.
template <bool UNIQ, int HASH>
__device__ __forceinline__ void drain(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine, bool all, bool mono) {
    uint32_t base = 0;
    while (w.fill - base >= 65u || (all && w.fill - base >= 2u)) {
        const uint32_t n = min(64u, w.fill - 1u - base);
        const bool valid = (uint32_t)w.lane < n;
        const uint32_t at = base + (valid ? (uint32_t)w.lane : 0u);
        const uint32_t ent = w.q[at], next = w.q[at + 1u];
        emit_chunk<UNIQ, HASH>(A, w, seen, bcur, mine, valid, ent, next, mono);
        base += n;
        if (UNIQ) drain_partial(A, w, bcur, mine, false);
    }
    if (all) {
        w.fill = 0;
        if (UNIQ) drain_partial(A, w, bcur, mine, true);
    } else if (base) {
        const uint32_t rem = w.fill - base;  // 1 .. 64
        const bool mv = (uint32_t)w.lane < rem;
        const uint32_t v = mv ? w.q[base + w.lane] : 0u;
        if (mv) w.q[w.lane] = v;
        w.fill = rem;
    }
}

// Pass A of block16 for eight consecutive steps of every lane, hand-scheduled: Mj (a lane mask in
// an SGPR pair) = "step j starts a run" = its id is not the id before it plus one, or it sits on
// a cut boundary; CNT += Mj per lane.  Five vector and one scalar instruction per step, where the
// compiler's rendering of the same C++ costs eight and three (it rebuilds every mask from a 0/1
// register).  PM is the id before step 0.
#define FGFA_PASSA_STEP(PMJ, XJ, MJ)                         \
    "v_add_u32 %[t], 1, %[" PMJ "]\n\t"                      \
    "v_cmp_ne_u32 %[" MJ "], %[" XJ "], %[t]\n\t"            \
    "v_and_b32 %[t], %[cut], %[" XJ "]\n\t"                  \
    "v_cmp_eq_u32 vcc, 0, %[t]\n\t"                          \
    "s_or_b64 %[" MJ "], %[" MJ "], vcc\n\t"                 \
    "v_addc_co_u32_e64 %[cnt], vcc, 0, %[cnt], %[" MJ "]\n\t"
#define FGFA_PASSA8(CUT, CNT, PM, X0, X1, X2, X3, X4, X5, X6, X7, M0, M1, M2, M3, M4, M5, M6, M7)                    \
    do {                                                                                                             \
        uint32_t t_;                                                                                                 \
        asm volatile(FGFA_PASSA_STEP("pm", "x0", "m0") FGFA_PASSA_STEP("x0", "x1", "m1")                             \
                         FGFA_PASSA_STEP("x1", "x2", "m2") FGFA_PASSA_STEP("x2", "x3", "m3")                         \
                             FGFA_PASSA_STEP("x3", "x4", "m4") FGFA_PASSA_STEP("x4", "x5", "m5")                     \
                                 FGFA_PASSA_STEP("x5", "x6", "m6") FGFA_PASSA_STEP("x6", "x7", "m7")                 \
                     : [cnt] "+v"(CNT), [t] "=&v"(t_), [m0] "=&s"(M0), [m1] "=&s"(M1), [m2] "=&s"(M2),               \
                       [m3] "=&s"(M3), [m4] "=&s"(M4), [m5] "=&s"(M5), [m6] "=&s"(M6), [m7] "=&s"(M7)                \
                     : [pm] "v"(PM), [x0] "v"(X0), [x1] "v"(X1), [x2] "v"(X2), [x3] "v"(X3), [x4] "v"(X4),           \
                       [x5] "v"(X5), [x6] "v"(X6), [x7] "v"(X7), [cut] "i"(CUT)                                      \
                     : "vcc", "scc");                                                                                \
    } while (0)

// Pass B of block16 for eight consecutive steps of every lane, hand-scheduled: for step j, the
// lanes where a run starts (mask Mj) append (step j's id << 10) | step j's position at their queue
// cursor `p`.  One scalar (exec = lanes that start a run), one LDS and three vector instructions per
// step (two for a lane's first), no branches; exec is restored before the statement ends.  B16 is 16 x lane.
#define FGFA_PASSB_STEP(T, XJ, J, MJ)                       \
    "s_and_b64 exec, %[sv], %[" MJ "]\n\t"                  \
    "v_lshl_or_b32 %[" T "], %[" XJ "], 10, %[b16]\n\t"     \
    "v_or_b32 %[" T "], " J ", %[" T "]\n\t"                \
    "ds_write_b32 %[p], %[" T "]\n\t"                       \
    "v_add_u32 %[p], 4, %[p]\n\t"
#define FGFA_PASSB8(P, B16, J0, J1, J2, J3, J4, J5, J6, J7, X0, X1, X2, X3, X4, X5, X6, X7, M0, M1, M2, M3, M4, M5, M6, M7)   \
    do {                                                                                                             \
        unsigned long long sv_;                                                                                      \
        uint32_t t0_, t1_;                                                                                           \
        asm volatile("s_mov_b64 %[sv], exec\n\t" FGFA_PASSB_STEP("t0", "x0", J0, "m0")                               \
                         FGFA_PASSB_STEP("t1", "x1", J1, "m1") FGFA_PASSB_STEP("t0", "x2", J2, "m2")                 \
                             FGFA_PASSB_STEP("t1", "x3", J3, "m3") FGFA_PASSB_STEP("t0", "x4", J4, "m4")             \
                                 FGFA_PASSB_STEP("t1", "x5", J5, "m5") FGFA_PASSB_STEP("t0", "x6", J6, "m6")         \
                                     FGFA_PASSB_STEP("t1", "x7", J7, "m7") "s_mov_b64 exec, %[sv]"                   \
                     : [p] "+v"(P), [sv] "=&s"(sv_), [t0] "=&v"(t0_), [t1] "=&v"(t1_)                                \
                     : [b16] "v"(B16), [x0] "v"(X0), [x1] "v"(X1), [x2] "v"(X2), [x3] "v"(X3), [x4] "v"(X4),         \
                       [x5] "v"(X5), [x6] "v"(X6), [x7] "v"(X7), [m0] "s"(M0), [m1] "s"(M1), [m2] "s"(M2),           \
                       [m3] "s"(M3), [m4] "s"(M4), [m5] "s"(M5), [m6] "s"(M6), [m7] "s"(M7)                          \
                     : "memory", "scc");                                                                             \
    } while (0)

// One block of a wave-per-path kernel: 1024 consecutive steps, sixteen per lane (lane l holds
// steps 16l..16l+15, i.e. its own 64 bytes), so that fifteen of every sixteen run boundaries are
// found with in-lane compares.  Only the first `nl` lanes hold steps.
//
// A run is queued where it STARTS, as (id, position in the block); it ends where the next entry
// starts, which is all its length takes when it is emitted (drain).  Pass A marks the starts and
// counts them per lane; a wave prefix sum gives every lane its own stretch of the run queue.  Pass
// B then walks the sixteen steps again and each lane appends its starts to its stretch: no ballot
// or lane ranking per step, nothing carried from step to step or from lane to lane.  A block is
// walked on its own: its first step starts a run, and behind its last step the last lane queues
// the entry that closes the last run.
// When the block has more starts than the queue has room for (dense: few steps continue a run),
// the steps are queued four at a time with the queue emitted in between.
// A block may reach beyond its path at either end (it starts and ends on 64-byte boundaries):
// steps at block-relative positions outside [rel_lo, rel_hi) get consecutive placeholder ids,
// whose runs are dropped when emitted.  With QONLY the block's runs are only queued, never
// emitted; the return value says whether they fitted the queue.
template <bool UNIQ, int HASH, bool QONLY>
__device__ __forceinline__ bool block16(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine,
                                        uint32_t (&a)[16], uint32_t nl, uint32_t rel_lo, uint32_t rel_hi, uint32_t blk_pos, bool mono) {
    const bool active = (uint32_t)w.lane < nl;
    const bool last_lane = (uint32_t)w.lane + 1u == nl;
    const bool partial = rel_lo > 0u || rel_hi < 16u * nl;
    // An id beyond the graph is looked for in everything the block holds, the steps of the neighbouring paths
    // (or a reversed copy's padding) included: only if there is one are the path's own steps checked one by one.
    uint32_t mx = a[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = max(mx, a[k]);
    if (mx >= A.n_segs) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t rel = 16u * (uint32_t)w.lane + (uint32_t)k;
            if (rel >= rel_lo && rel < rel_hi) a[k] = clamp_id(A, a[k]);
        }
    }
    if (partial) {  // (every block of a path of a thousand steps: four instructions a step and no branch -- as `inside ? clamp_id(..) : dummy` it was nine and six scalar ones)
        const int lo_l = (int)rel_lo - 16 * w.lane, hi_l = (int)rel_hi - 16 * w.lane;
        const uint32_t lo_c = (uint32_t)min(max(lo_l, 0), 16), hi_c = (uint32_t)min(max(hi_l, 0), 16);
        const uint32_t m16 = hi_c > lo_c ? ((1u << hi_c) - 1u) & ~((1u << lo_c) - 1u) : 0u;  // this lane's steps inside the path
        const uint32_t d0 = kDummyBase + ((blk_pos + 16u * (uint32_t)w.lane) & 0xFFFFu);       // (a multiple of 16: the placeholder of step k is d0 | k)
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = (m16 >> k) & 1u ? a[k] : d0 | (uint32_t)k;
    }
    const uint32_t prev = __builtin_amdgcn_update_dpp(0u, a[15], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    // pass A (lanes beyond `nl` compute garbage flags; they are kept out of `cnt` and of pass B)
    unsigned long long m[16];
    uint32_t cnt = 0;
    FGFA_PASSA8(kCutMask<UNIQ>, cnt, prev, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
    FGFA_PASSA8(kCutMask<UNIQ>, cnt, a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
    cnt += (w.lane == 0) ? 1u - (uint32_t)(m[0] & 1ull) : 0u;  // the block's first step starts a run whatever lies before it
    m[0] |= 1ull;
    cnt = active ? cnt : 0u;
    const uint32_t slots = cnt + (last_lane ? 1u : 0u);  // the last lane also queues the entry that closes the block's last run
    const uint32_t incl = wave_scan_incl(slots);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    const uint32_t b16 = 16u * (uint32_t)w.lane;
    const uint32_t term = kTermEntry | ((16u * nl) & ((1u << kPosBits) - 1u));
    if (w.fill + total <= w.qcap) {
        uint32_t p = lds_addr(w.q + w.fill + (incl - slots));
        if (active) {
            FGFA_PASSB8(p, b16, "0", "1", "2", "3", "4", "5", "6", "7", a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
            FGFA_PASSB8(p, b16, "8", "9", "10", "11", "12", "13", "14", "15", a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
        }
        if (last_lane) *reinterpret_cast<lds_u32 *>((uintptr_t)p) = term;
        w.fill += total;
        if (QONLY) return true;
        drain<UNIQ, HASH>(A, w, seen, bcur, mine, false, mono);
    } else {
        if (QONLY) return false;
        // Entries must lie in the order of their positions, so the block is queued sixteen lanes at a time (at most 256
        // starts and the closing entry), the queue emitted down to at most 64 entries before each.
#pragma unroll 1
        for (uint32_t grp = 0; grp < 4u; ++grp) {
            drain<UNIQ, HASH>(A, w, seen, bcur, mine, false, mono);
            const bool in_g = active && ((uint32_t)w.lane >> 4) == grp;
            const uint32_t sl = in_g ? slots : 0u;
            const uint32_t inc = wave_scan_incl(sl);
            uint32_t p = lds_addr(w.q + w.fill + (inc - sl));
            if (in_g) {
                FGFA_PASSB8(p, b16, "0", "1", "2", "3", "4", "5", "6", "7", a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
                FGFA_PASSB8(p, b16, "8", "9", "10", "11", "12", "13", "14", "15", a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
                if (last_lane) *reinterpret_cast<lds_u32 *>((uintptr_t)p) = term;
            }
            w.fill += __builtin_amdgcn_readlane(inc, 63);
        }
        drain<UNIQ, HASH>(A, w, seen, bcur, mine, false, mono);
    }
    return true;
}

struct ShortBlk {
    uint32_t b, e;     // the path's steps
    uint32_t pos;      // first step of the block (a multiple of 16)
    uint32_t nl;       // lanes holding steps
    uint32_t item;     // the path's position in the list of short paths
    bool last, valid;  // last block of its path; there is a block at all
    bool skip;         // (paired waves) no block: the path's end as seen by the wave whose partner walks its last block
};

// The blocks of this wave's paths, in order.  The descriptors of the wave's next 64 paths are read
// with ONE load, a path per lane, and handed out by v_readlane: read one at a time -- even a path
// ahead of its use -- hipcc waits for the load where it is issued, with vmcnt(0), which also waits
// for the blocks in flight: a memory round trip and a drained pipeline per path, i.e. per block
// where paths have a thousand steps.
struct ShortStream {
    uint32_t gi, stride, b, e, pos, end, nb, ne;
    uint32_t bx, by;  // (per lane) first and last step of the path lane * stride behind the batch's first
    uint32_t bk;      // descriptors of the batch handed out so far
};
__device__ __forceinline__ void stream_fetch(const ScanArgs &A, ShortStream &g, int lane) {  // descriptor of path gi + stride
    const uint32_t nx = g.gi + g.stride;
    if (g.bk >= 64u) {
        const uint64_t idx = (uint64_t)nx + (uint64_t)lane * g.stride;
        uint2 d = make_uint2(0u, 0u);
        if (nx >= g.gi && idx < A.n_short) d = *reinterpret_cast<const uint2 *>(A.short_items + idx);
        asm volatile("" : "+v"(d.x), "+v"(d.y));  // (the wait for the load belongs here, once per batch: left pending, hipcc waits where the paths change, every time)
        g.bx = d.x;
        g.by = d.y;
        g.bk = 0u;
    }
    const bool have = nx < A.n_short && nx >= g.gi;
    g.nb = have ? (uint32_t)__builtin_amdgcn_readlane((int)g.bx, (int)g.bk) : 0u;
    g.ne = have ? (uint32_t)__builtin_amdgcn_readlane((int)g.by, (int)g.bk) : 0u;
    g.bk += 1u;
}
__device__ __forceinline__ ShortBlk stream_next(const ScanArgs &A, ShortStream &g, int lane) {
    ShortBlk k;
    k.skip = false;
    k.valid = g.gi < A.n_short;
    k.b = g.b;
    k.e = g.e;
    k.pos = g.pos;
    k.item = g.gi;
    const uint32_t left = k.valid ? (g.end - g.pos) / 16u : 0u;
    k.nl = min(left, 64u);
    k.last = left <= 64u;
    g.pos += 1024u;
    if (k.valid && k.last) {
        g.gi = (g.gi + g.stride >= g.gi) ? g.gi + g.stride : 0xFFFFFFFFu;
        g.b = g.nb;
        g.e = g.ne;
        g.pos = g.b & ~15u;
        g.end = (g.e + 15u) & ~15u;
        stream_fetch(A, g, lane);
    }
    return k;
}

// WAVES waves per workgroup, each with a hash set of 2^HASH entries.  QONLY: a path's runs are
// queued first and emitted when it is complete (short paths; those that do not fit are handed
// back to k_scan); otherwise they are emitted as they come (medium paths, whose run count the
// plan knows).
// PAIRED (medium paths): TWO waves per path and hash set -- the even wave of a pair walks the path's even
// blocks, the odd one its odd blocks (a block is walked on its own, the claims are LDS atomics, the records
// go through the workgroup's cursors: nothing else is shared) -- so that a 16 KB set costs a CU's LDS eight
// bytes per lane instead of sixteen and fourteen waves fit where eight did.  When a path ends both waves
// meet (a counter in LDS each adds to and then polls; both are resident, neither waits for anything else),
// wipe half of the set each and meet again.
template <bool PAIRED>
__device__ __forceinline__ void pair_meet(uint32_t *ctr, int lane, uint32_t &target) {
    if (!PAIRED) return;
    target += 2u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's claims and wipes are in the LDS before its count is)
    if (lane == 0) atomicAdd(ctr, 1u);
    while ((int)(*reinterpret_cast<volatile uint32_t *>(ctr) - target) < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// MONO: the list has paths that need no claims (ScanArgs::mono_lo / mono_n) -- a build of its own: the test costs the
// lists without any 1-3 %.
template <bool UNIQ, int WAVES, int HASH, bool QONLY, bool PAIRED = false, bool MONO = false>
__global__ __launch_bounds__(WAVES * 64) void k_scan_short(const ScanArgs A) {
    static_assert(!PAIRED || (!QONLY && WAVES % 2 == 0), "pairs walk medium paths");
    constexpr uint32_t kTab = 1u << HASH;
    constexpr int kThr = WAVES * 64;
    constexpr int kSets = PAIRED ? WAVES / 2 : WAVES;
    constexpr uint32_t kQ = PAIRED ? kQPaired : kQCap;
    extern __shared__ uint32_t lds[];
    // layout: [bcur: kShortMaxWin][run queues: WAVES * kQ][parked-claim queues: WAVES * 2 * kPCap][hash sets: kSets * 2 * kTab][dummy: 128][pair counters: kSets]
    uint32_t *bcur = lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int set = PAIRED ? wave >> 1 : wave;
    const uint32_t me = PAIRED ? (uint32_t)wave & 1u : 0u;
    uint32_t *tab = lds + kShortMaxWin + WAVES * (kQ + 2 * kPCap) + set * (2 * kTab);
    uint32_t *mine = A.buckets + (size_t)blockIdx.x * A.cap;
    Wave w;
    w.q = lds + kShortMaxWin + wave * kQ;
    w.pq = lds + kShortMaxWin + WAVES * kQ + wave * (2 * kPCap);
    w.dummy = lds + kShortMaxWin + WAVES * (kQ + 2 * kPCap) + kSets * (2 * kTab);  // (shared by the waves: what lands there is never read)
    uint32_t *meet = w.dummy + 128 + set;
    uint32_t met = 0;  // what the pair's counter reads when both have arrived
    w.qcap = kQ;
    w.fill = w.pfill = 0;
    w.vm[0] = w.vm[1] = w.vm[2] = 0;
    w.lane = lane;
    for (uint32_t i = threadIdx.x; i < kShortMaxWin; i += kThr) bcur[i] = i < A.n_win ? A.counts[(size_t)i * A.n_slots + blockIdx.x] : 0u;
    // (a wave wipes its own set, or its half of the pair's)
    const auto wipe = [&]() {
        for (uint32_t i = (uint32_t)lane + 64u * me; i < kTab / 2; i += PAIRED ? 128u : 64u) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0u, 0u, 0u, 0u);
    };
    if (UNIQ) wipe();
    if (PAIRED && threadIdx.x < (uint32_t)kSets) w.dummy[128 + threadIdx.x] = 0u;
    __syncthreads();

    ShortStream g;
    g.stride = gridDim.x * kSets;
    g.gi = blockIdx.x * kSets + set;
    {
        const uint4 d = g.gi < A.n_short ? A.short_items[g.gi] : make_uint4(0u, 0u, 0u, 0u);
        g.b = d.x;
        g.e = d.y;
        g.pos = g.b & ~15u;
        g.end = (g.e + 15u) & ~15u;
        g.bx = g.by = 0u;
        g.bk = 64u;
        stream_fetch(A, g, lane);
    }
    // this wave's next block: of a pair, the blocks of its parity, and -- where the path's last block is the
    // partner's -- the path's end without a block
    const auto next_own = [&]() {
        ShortBlk k = stream_next(A, g, lane);
        if (PAIRED && k.valid && ((((k.pos - (k.b & ~15u)) >> 10) & 1u) != me)) {
            if (k.last) {
                k.skip = true;
                k.nl = 0u;
            } else {
                k = stream_next(A, g, lane);  // (the same path's next block: this wave's)
            }
        }
        return k;
    };
    ShortBlk slot[2];
    bool handed_back = false;  // the current path did not fit the run queue
    const uint4 *steps4 = reinterpret_cast<const uint4 *>(A.steps), *rev4 = reinterpret_cast<const uint4 *>(A.rev_steps);
    // lanes beyond the last one holding steps re-read lane 0's chunk
#define FGFA_SPTR(K) (((K).item >= A.n_fwd ? rev4 : steps4) + (size_t)(K).pos / 4 + ((uint32_t)lane < (K).nl ? lane * 4 : 0))
    slot[0] = next_own();
    if (slot[0].valid && !slot[0].skip) load_block_async<0>(w, FGFA_SPTR(slot[0]));
    slot[1] = next_own();
    if (slot[1].valid && !slot[1].skip) load_block_async<1>(w, FGFA_SPTR(slot[1]));
    // -DFGFA_SHORT_PROF (tools/short_prof.py): cycles per phase of two workgroups' waves, printed when the kernel ends
#ifdef FGFA_SHORT_PROF
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
#define SP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_readcyclecounter(); tp[i] += n_ - tl; tl = n_; } while (0)
#else
#define SP(i)
#endif
#define FGFA_SBLOCK(SET)                                                                                \
    if (slot[SET].valid) {                                                                              \
        const ShortBlk cur = slot[SET];                                                                 \
        const bool mono = MONO && UNIQ && cur.item - A.mono_lo < A.mono_n;  /* (uniform) this path needs no claims */ \
        uint32_t a[16];                                                                                 \
        SP(0);                                                                                          \
        if (!cur.skip) {                                                                                \
            wait_block<SET>(w);                                                                         \
            SP(1);                                                                                      \
            take_block<SET>(a);                                                                         \
        }                                                                                               \
        slot[SET] = next_own();                                                                         \
        if (slot[SET].valid && !slot[SET].skip) load_block_async<SET>(w, FGFA_SPTR(slot[SET]));         \
        SP(2);                                                                                          \
        if (cur.skip) {                                                                                 \
        } else if (FGFA_SHORT_ABLATE & 1) {                                                             \
            uint32_t x_ = a[0];                                                                         \
            for (int k_ = 1; k_ < 16; ++k_) x_ ^= a[k_];                                                \
            if (x_ == 0xDEADBEEFu) atomicOr(A.status, kStBounds);                                       \
        } else if (!handed_back) {                                                                      \
            const uint32_t lo = cur.b > cur.pos ? cur.b - cur.pos : 0u;                                 \
            const uint32_t hi = cur.e - cur.pos < 1024u ? cur.e - cur.pos : 1024u;                      \
            if (!block16<UNIQ, HASH, QONLY>(A, w, tab, bcur, mine, a, cur.nl, lo, hi, cur.pos, mono)) { \
                handed_back = true;                                                                     \
                w.fill = 0;                                                                             \
            }                                                                                           \
        }                                                                                               \
        SP(3);                                                                                          \
        if (cur.last) {                                                                                 \
            if (handed_back) {                                                                          \
                if (lane == 0) {                                                                        \
                    const uint32_t k = atomicAdd(A.work_counter, 1u);                                   \
                    const uint32_t hp = A.short_items[cur.item].w;                                      \
                    if (k < A.max_back) A.items[A.n_items + k] = cur.item >= A.n_fwd ? make_uint4(A.path_begin[hp], A.path_end[hp], 0u, hp) : make_uint4(cur.b, cur.e, 0u, hp); \
                    else atomicOr(A.status, kStBackOverflow);                                           \
                }                                                                                       \
                handed_back = false;                                                                    \
            } else {                                                                                    \
                if (FGFA_SHORT_ABLATE & 2) w.fill = 0;                                                  \
                drain<UNIQ, HASH>(A, w, tab, bcur, mine, true, mono);                                   \
                if (UNIQ && !mono && !(FGFA_SHORT_ABLATE & 16)) {  /* (both waves of a pair see the same path) */ \
                    pair_meet<PAIRED>(meet, lane, met);  /* both are through with the path's claims */  \
                    wipe();                                                                             \
                    pair_meet<PAIRED>(meet, lane, met);  /* ... and the set is empty for the next */    \
                }                                                                                       \
            }                                                                                           \
        }                                                                                               \
        SP(4);                                                                                          \
    }
#pragma unroll 1
    while (slot[0].valid || slot[1].valid) {
        FGFA_SBLOCK(0)
        FGFA_SBLOCK(1)
    }
#undef FGFA_SBLOCK
#undef FGFA_SPTR
#ifdef FGFA_SHORT_PROF
    if ((blockIdx.x == 0 || blockIdx.x == 100) && (threadIdx.x == 0 || threadIdx.x == 64 * 9)) printf("short wg %u wave %u: between %llu wait %llu take+next+issue %llu block16 %llu path-end %llu\n", blockIdx.x, threadIdx.x >> 6, tp[0], tp[1], tp[2], tp[3], tp[4]);
#endif
    __syncthreads();
    for (uint32_t wdw = threadIdx.x; wdw < A.n_win; wdw += kThr)
        A.counts[(size_t)wdw * A.n_slots + blockIdx.x] = bcur[wdw];
}

template <bool UNIQ, bool MONO = false>
constexpr auto k_walk_short = k_scan_short<UNIQ, kShortWaves, kShortHash, true, false, MONO>;
template <bool UNIQ, bool MONO = false>
constexpr auto k_walk_medium = k_scan_short<UNIQ, kMediumWaves, kMediumHash, false, kMediumPaired, MONO>;

// ------------------------------------------------------------ pass 1, tiny paths ---
//
// k_scan_tiny: paths of at most 128 steps (a million contigs of a hundred steps took k_scan_short a
// millisecond: a block of 1024 lanes-times-steps for a hundred steps, a 4 KB hash set wiped and a
// queue drained with two lanes in sixty-four busy, per path).  Here a wave holds a whole path in two
// registers per lane (steps l and 64 + l), three paths' loads in flight, and keeps no queue:
//   * first visits: each step's segment id goes into a per-wave hash set of 256 ids (one
//     compare-and-swap per probe; whichever step of a (path, segment) pair gets there first is the
//     first visit -- unique depth counts segments, not positions);
//   * a record starts where the id is not the previous id plus one, where the first-visit flag
//     changes, and at window boundaries, so a record lies in one window and counts for depth only
//     or for depth and unique depth as a whole (bits 24 / 25, as k_scan_short's: pass 2 applies
//     them without claims);
//   * its length is the distance to the next start, read off the two ballot masks;
//   * records are queued per wave and leave 64 at a time.
#ifndef FGFA_TINY_ABLATE
#define FGFA_TINY_ABLATE 0  /* measurements only (results are then wrong): 1 = no first-visit test, 2 = no records */
#endif

// (landing registers: path slot d's steps l and 64 + l in v(118 + 2 d), v(119 + 2 d))
template <int D>
__device__ __forceinline__ void tiny_request(const uint32_t *p, uint32_t off0, uint32_t off1) {
#define FGFA_TREQ(R0, R1) asm volatile("global_load_dword " R0 ", %0, %2\n\tglobal_load_dword " R1 ", %1, %2" ::"v"(off0), "v"(off1), "s"(p) : "memory", R0, R1)
    if (D == 0) FGFA_TREQ("v118", "v119");
    else if (D == 1) FGFA_TREQ("v120", "v121");
    else FGFA_TREQ("v122", "v123");
#undef FGFA_TREQ
}
template <int D>
__device__ __forceinline__ void tiny_take(uint32_t n_since, uint32_t &a0, uint32_t &a1) {
    // (loads and stores return in issue order: slot D's pair is there once at most the n_since operations issued behind it are outstanding)
    if (n_since >= 8u) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n_since >= 6u) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n_since >= 4u) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define FGFA_TTAKE(R0, R1) asm volatile("v_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 : "=v"(a0), "=v"(a1)::"memory")
    if (D == 0) FGFA_TTAKE("v118", "v119");
    else if (D == 1) FGFA_TTAKE("v120", "v121");
    else FGFA_TTAKE("v122", "v123");
#undef FGFA_TTAKE
}

template <bool UNIQ, bool MONO = false>
__global__ __launch_bounds__(kThreads) void k_scan_tiny(const ScanArgs A) {
    extern __shared__ uint32_t lds[];
    // layout: [bcur: kShortMaxWin][id sets: kWaves * kTinyTab][record queues: kWaves * kTinyQueue entries of 8 bytes][dummy: 128]
    uint32_t *bcur = lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *tab = lds + kShortMaxWin + wave * kTinyTab;
    uint2 *q = reinterpret_cast<uint2 *>(lds + kShortMaxWin + kWaves * kTinyTab) + wave * kTinyQueue;  // {record, window}
    uint32_t *dummy = lds + kShortMaxWin + kWaves * kTinyTab + kWaves * kTinyQueue * 2u;  // 128 words nobody reads (shared by the waves)
    uint32_t fill = 0;  // (uniform) entries in the queue: fewer than 64 between paths
    uint32_t *mine = A.buckets + (size_t)blockIdx.x * A.cap;
    Wave w;
    w.q = w.pq = nullptr;
    w.fill = w.pfill = 0;
    w.vm[0] = w.vm[1] = w.vm[2] = 0;
    w.lane = lane;
    for (uint32_t i = threadIdx.x; i < kShortMaxWin; i += kThreads) bcur[i] = i < A.n_win ? A.counts[(size_t)i * A.n_slots + blockIdx.x] : 0u;
    if (UNIQ)
        for (uint32_t i = (uint32_t)lane; i < kTinyTab / 4u; i += 64u) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // This wave's paths: gi, gi + stride, ...  Their descriptors come 64 at a time (one load, a path
    // per lane) and are handed out by v_readlane; a path's steps are requested three paths ahead.
    const uint32_t stride = gridDim.x * kWaves;
    uint32_t gi = blockIdx.x * kWaves + (uint32_t)wave;  // the path whose steps are requested next
    uint32_t bx = 0, by = 0, bk = 64u;                   // (per lane) the batch; how many of it are handed out
    uint32_t pb[3] = {0u, 0u, 0u}, pn[3] = {0u, 0u, 0u};  // the paths in flight: first step, number of steps (0: none)
    uint32_t pm[3] = {0u, 0u, 0u};                        // ... and whether the plan found them to walk the ids strictly one way (no first-visit test)
    uint32_t since[3] = {0u, 0u, 0u};                     // memory operations issued behind each slot's loads
    const auto next_path = [&](uint32_t &b, uint32_t &n, uint32_t &m) {
        n = 0u;
        b = 0u;
        m = 0u;
        if (gi >= A.n_short) return;
        m = MONO && gi - A.mono_lo < A.mono_n ? 1u : 0u;
        if (bk >= 64u) {
            const uint64_t idx = (uint64_t)gi + (uint64_t)lane * stride;
            uint2 d = make_uint2(0u, 0u);
            if (idx < A.n_short) d = *reinterpret_cast<const uint2 *>(A.short_items + idx);
            bx = d.x;
            by = d.y;
            bk = 0u;
        }
        b = (uint32_t)__builtin_amdgcn_readlane((int)bx, (int)bk);
        n = (uint32_t)__builtin_amdgcn_readlane((int)by, (int)bk) - b;
        bk += 1u;
        gi = gi + stride >= gi ? gi + stride : 0xFFFFFFFFu;
    };
#define FGFA_TINY_REQ(D)                                                                            \
    do {                                                                                            \
        next_path(pb[D], pn[D], pm[D]);                                                             \
        if (pn[D]) {                                                                                \
            /* lanes beyond the path re-read its first step */                                      \
            const uint32_t o0 = (uint32_t)lane < pn[D] ? 4u * (uint32_t)lane : 0u;                  \
            const uint32_t o1 = 64u + (uint32_t)lane < pn[D] ? 256u + 4u * (uint32_t)lane : 0u;    \
            tiny_request<D>(A.steps + pb[D], o0, o1);                                               \
            since[0] += 2u, since[1] += 2u, since[2] += 2u;                                         \
            since[D] = 0u;                                                                          \
        }                                                                                           \
    } while (0)
    FGFA_TINY_REQ(0);
    FGFA_TINY_REQ(1);
    FGFA_TINY_REQ(2);
    bool bad = false, ovf = false;
#define FGFA_TINY_PATH(D)                                                                                              \
    if (pn[D]) {                                                                                                       \
        uint32_t a0, a1;                                                                                               \
        tiny_take<D>(since[D], a0, a1);                                                                                \
        const uint32_t n = pn[D];                                                                                      \
        const bool mono_##D = pm[D] != 0u;                                                                             \
        FGFA_TINY_REQ(D);                                                                                              \
        const bool v0 = (uint32_t)lane < n, v1 = 64u + (uint32_t)lane < n;                                             \
        uint32_t x0 = a0 >> 1, x1 = a1 >> 1;                                                                           \
        bad |= (v0 && x0 >= A.n_segs) || (v1 && x1 >= A.n_segs);                                                       \
        x0 = x0 < A.n_segs ? x0 : 0u;                                                                                  \
        x1 = x1 < A.n_segs ? x1 : 0u;                                                                                  \
        bool f0 = true, f1 = true;  /* first visits */                                                                 \
        if (UNIQ && !(FGFA_TINY_ABLATE & 1) && !mono_##D) {  /* (a path that walks the ids strictly one way: all first visits) */ \
            uint32_t h0 = (x0 * 0x9E3779B1u) >> (32 - kTinyBits), h1 = (x1 * 0x9E3779B1u) >> (32 - kTinyBits);                                     \
            /* the first probe -- nearly always the last -- by all lanes, nothing predicated (a lane without a step */   \
            /* probes a word of its own in `dummy`): as `if (t0) CAS` the loop was mostly scalar exec-mask traffic  */   \
            uint32_t *e0 = v0 ? &tab[h0] : &dummy[lane], *e1 = v1 ? &tab[h1] : &dummy[64 + lane];                      \
            const uint32_t c0 = atomicCAS(e0, 0u, x0 + 1u), c1 = atomicCAS(e1, 0u, x1 + 1u);                           \
            f0 = c0 == 0u;                                                                                             \
            f1 = c1 == 0u;                                                                                             \
            bool t0 = v0 && c0 != 0u && c0 != x0 + 1u, t1 = v1 && c1 != 0u && c1 != x1 + 1u;                           \
            h0 = (h0 + 1u) & (kTinyTab - 1u);                                                                          \
            h1 = (h1 + 1u) & (kTinyTab - 1u);                                                                          \
            uint32_t probes = 0;                                                                                       \
            while (__builtin_amdgcn_ballot_w64(t0 || t1)) {  /* (the few that met another id's entry; unpredicated like the first probe this loop measured 3 % slower) */ \
                if (++probes > 2u * kTinyTab) {  /* cannot happen: the set holds at most 128 ids */                    \
                    atomicOr(A.status, kStInternal);                                                                   \
                    break;                                                                                             \
                }                                                                                                      \
                if (t0) {                                                                                              \
                    const uint32_t k = atomicCAS(&tab[h0], 0u, x0 + 1u);                                               \
                    if (k == 0u || k == x0 + 1u) f0 = k == 0u, t0 = false;                                             \
                    else h0 = (h0 + 1u) & (kTinyTab - 1u);                                                             \
                }                                                                                                      \
                if (t1) {                                                                                              \
                    const uint32_t k = atomicCAS(&tab[h1], 0u, x1 + 1u);                                               \
                    if (k == 0u || k == x1 + 1u) f1 = k == 0u, t1 = false;                                             \
                    else h1 = (h1 + 1u) & (kTinyTab - 1u);                                                             \
                }                                                                                                      \
            }                                                                                                          \
            for (uint32_t i_ = (uint32_t)lane; i_ < kTinyTab / 4u; i_ += 64u) reinterpret_cast<uint4 *>(tab)[i_] = make_uint4(0u, 0u, 0u, 0u);  /* clean for the next path */ \
        }                                                                                                              \
        /* where records start */                                                                                      \
        const uint32_t k0 = x0 | (f0 ? 0x80000000u : 0u), k1 = x1 | (f1 ? 0x80000000u : 0u);  /* id and flag in one word */ \
        const uint32_t q0 = __builtin_amdgcn_update_dpp(0u, k0, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);              \
        uint32_t q1 = __builtin_amdgcn_update_dpp(0u, k1, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);                    \
        const uint32_t k0_last = __builtin_amdgcn_readlane(k0, 63);                                                    \
        q1 = lane == 0 ? k0_last : q1;                                                                                 \
        const bool s0 = v0 && (lane == 0 || k0 != q0 + 1u || (x0 & 4095u) == 0u);                                      \
        const bool s1 = v1 && (k1 != q1 + 1u || (x1 & 4095u) == 0u);                                                   \
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(s0), m1 = __builtin_amdgcn_ballot_w64(s1);           \
        /* a record's length: to the next start, or to the end of the path */                                          \
        const unsigned long long r0 = (m0 >> 1) >> lane, r1 = (m1 >> 1) >> lane;                                       \
        const uint32_t len0 = r0 ? (uint32_t)__builtin_ctzll(r0) + 1u                                                  \
                                 : m1 ? 64u - (uint32_t)lane + (uint32_t)__builtin_ctzll(m1) : n - (uint32_t)lane;     \
        const uint32_t len1 = r1 ? (uint32_t)__builtin_ctzll(r1) + 1u : n - 64u - (uint32_t)lane;                      \
        /* The records are queued (a path has a dozen: a store instruction per path would run with ten lanes in */     \
        /* sixty-four busy) and leave 64 at a time; which ones go together does not matter to pass 2.             */    \
        if (!(FGFA_TINY_ABLATE & 2)) {                                                                                 \
            if (s0) q[fill + lane_rank(m0)] = make_uint2((x0 & 4095u) | ((len0 - 1u) << kShortWinBits) | ((UNIQ && f0 ? 3u : 1u) << 24), x0 >> kShortWinBits); \
            fill += (uint32_t)__builtin_popcountll(m0);                                                                \
            if (s1) q[fill + lane_rank(m1)] = make_uint2((x1 & 4095u) | ((len1 - 1u) << kShortWinBits) | ((UNIQ && f1 ? 3u : 1u) << 24), x1 >> kShortWinBits); \
            fill += (uint32_t)__builtin_popcountll(m1);                                                                \
            while (fill >= 64u) {                                                                                      \
                fill -= 64u;                                                                                           \
                const uint2 e = q[fill + (uint32_t)lane];                                                              \
                const uint32_t pos = take_slots(bcur, lane, true, e.y);                                                \
                ovf |= put<false>(A, w, mine, true, pos, e.y, e.x);                                                    \
                since[0] += 1u, since[1] += 1u, since[2] += 1u;                                                        \
            }                                                                                                          \
        }                                                                                                              \
    }
#pragma unroll 1
    while (pn[0] || pn[1] || pn[2]) {
        FGFA_TINY_PATH(0)
        FGFA_TINY_PATH(1)
        FGFA_TINY_PATH(2)
    }
#undef FGFA_TINY_PATH
#undef FGFA_TINY_REQ
    if (fill) {  // what is left in the queue
        const bool v = (uint32_t)lane < fill;
        const uint2 e = v ? q[lane] : make_uint2(0u, 0u);
        const uint32_t pos = take_slots(bcur, lane, v, e.y);
        ovf |= put<false>(A, w, mine, v, pos, e.y, e.x);
    }
    flag_if_any(A, bad, kStBounds);
    flag_if_any(A, ovf, kStOverflow);
    __syncthreads();
    for (uint32_t wdw = threadIdx.x; wdw < A.n_win; wdw += kThreads) A.counts[(size_t)wdw * A.n_slots + blockIdx.x] = bcur[wdw];
}

}  // namespace

bool path_kernels_setup() {
    static OncePerDevice once;
    const bool ok = once([] {
        bool good = true;
        for (const void *k : {(const void *)k_walk_short<true>, (const void *)k_walk_short<false>, (const void *)k_walk_medium<true>,
                              (const void *)k_walk_medium<false>, (const void *)k_walk_short<true, true>, (const void *)k_walk_medium<true, true>})
            good = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit) == hipSuccess && good;
        return good;
    });
    if (!ok) set_error("hipFuncSetAttribute(k_scan_short): dynamic shared memory");
    return ok;
}

// (dynamic LDS: the cursor table, per wave its run queue, parked claims and hash set -- a medium pair shares one set --, the dummies)
void launch_scan_tiny(const FastPlan &fp, const ScanArgs &sk, bool uniq, uint32_t grid, hipStream_t stream) {
    const uint32_t lds = (kShortMaxWin + kWaves * kTinyTab + kWaves * kTinyQueue * 2u + 128u) * 4u;
    if (uniq && sk.mono_n) hipLaunchKernelGGL((k_scan_tiny<true, true>), dim3(grid), dim3(kThreads), lds, stream, sk);
    else if (uniq) hipLaunchKernelGGL(k_scan_tiny<true>, dim3(grid), dim3(kThreads), lds, stream, sk);
    else hipLaunchKernelGGL(k_scan_tiny<false>, dim3(grid), dim3(kThreads), lds, stream, sk);
}

void launch_scan_short(const FastPlan &fp, const ScanArgs &sk, bool medium, bool uniq, uint32_t grid, hipStream_t stream) {
    const uint32_t lds_short = (kShortMaxWin + kShortWaves * (kQCap + 2 * kPCap + (2u << kShortHash)) + 128u + 16u) * 4u;
    const uint32_t lds_medium = kMediumPaired ? (kShortMaxWin + kMediumWaves * (kQPaired + 2 * kPCap) + (kMediumWaves / 2) * (2u << kMediumHash) + 128u + 16u) * 4u
                                              : (kShortMaxWin + kMediumWaves * (kQCap + 2 * kPCap + (2u << kMediumHash)) + 128u + 16u) * 4u;
    if (medium) {
        if (uniq && sk.mono_n) hipLaunchKernelGGL((k_walk_medium<true, true>), dim3(grid), dim3(kMediumWaves * 64), lds_medium, stream, sk);
        else if (uniq) hipLaunchKernelGGL(k_walk_medium<true>, dim3(grid), dim3(kMediumWaves * 64), lds_medium, stream, sk);
        else hipLaunchKernelGGL(k_walk_medium<false>, dim3(grid), dim3(kMediumWaves * 64), lds_medium, stream, sk);
    } else {
        if (uniq && sk.mono_n) hipLaunchKernelGGL((k_walk_short<true, true>), dim3(grid), dim3(kShortWaves * 64), lds_short, stream, sk);
        else if (uniq) hipLaunchKernelGGL(k_walk_short<true>, dim3(grid), dim3(kShortWaves * 64), lds_short, stream, sk);
        else hipLaunchKernelGGL(k_walk_short<false>, dim3(grid), dim3(kShortWaves * 64), lds_short, stream, sk);
    }
}

}  // namespace fgfa_dev
