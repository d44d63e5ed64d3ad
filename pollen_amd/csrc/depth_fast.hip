// The bucketed node-depth path for gfx950 (seg_depth_with_uniq / seg_depth, ops/depth.rs:15-56):
// three kernels, no global atomics on the data path.  DESIGN.md section 3 is the long version.
//
//   k_scan        (pass 1)  one persistent workgroup per CU walks one path (or one piece of a
//                 long path) at a time.  The path's steps are cut into blocks of 1024; a wave takes
//                 a block, every lane owns sixteen consecutive steps (its own 64 bytes), finds
//                 the maximal +1 runs of segment ids among them, and queues each run as ONE range
//                 record (start id, length) instead of `length` histogram updates.  For unique
//                 depth the path's "seen" bitset of depth.rs:23-34 lives in LDS (1 bit per
//                 segment): a queued run claims its bits with one returning LDS OR (runs are cut
//                 at multiples of 32), and the bits that were clear are what the path touches for
//                 the first time.  A record goes to the bucket of its 4096-segment window; buckets
//                 are split into one private sub-bucket per workgroup, so the append cursor is an
//                 LDS counter and a workgroup's partial lines stay in its own XCD's L2.
//   k_scan_short  (pass 1 for paths of at most 2048 steps)  every wave walks whole paths on its
//                 own -- no barrier, no 125 KB bitset: the same blocks, but a path's runs are
//                 queued first and then claimed in a per-wave hash set of bitset words.
//   k_accum       (pass 2)  one workgroup per window: applies the window's records as +1/-1 pairs
//                 to an LDS difference array (depth and uniq packed in 64-bit cells), prefix-sums
//                 it and writes depth/uniq with 16-byte stores.  It also zeroes the counts it
//                 consumed, so the scratch is clean for the next call without any memset.
//
// Exactness: every step lies in exactly one run, so it contributes +1 to exactly one depth
// record; every (path, segment) pair that occurs sets exactly one bit, claimed by exactly one
// lane, covered by exactly one uniq record.  Sums of +1s are order-independent, hence the results
// equal depth.rs bit for bit under any scheduling.  A record never crosses a window.  Sub-buckets
// have a fixed capacity; a record that does not fit is applied to a global difference array with
// atomics instead (slow, still exact) and k_accum folds that array in.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <string>
#include <vector>

#include "depth_fast.hpp"
#include "device_common.hpp"
#include "prof.hpp"

namespace fgfa_dev {
namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr uint32_t kRunBits = 11;  // a queued run is (start id << 11) | (len - 1)
constexpr uint32_t kRunSpan = 1u << kRunBits;
// Where runs are cut: depth-only runs at ids that are multiples of 2048 (a record never crosses a
// window); with unique depth at multiples of 32, so that a run lies inside ONE word of the "seen"
// bitset and is claimed with a single returning LDS OR.
template <bool UNIQ>
constexpr uint32_t kCutMask = UNIQ ? 31u : kRunSpan - 1u;
constexpr uint32_t kWinBits = 12;  // accumulation window: 4096 segment ids = 128 bitset words
constexpr uint32_t kWin = 1u << kWinBits;
constexpr uint32_t kWinWords = kWin / 32;
constexpr uint32_t kMaxWin = 256;  // LDS cursor table entries (the bitset limit keeps n_win below this)
constexpr uint32_t kLdsLimit = 160 * 1024;
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
// The short-path kernel (k_scan_short): every wave walks whole short paths on its own.
constexpr uint32_t kShortMax = 2048;      // steps; longer paths go through k_scan
constexpr uint32_t kDummyBase = 1u << 20;  // ids from here up stand in for steps outside the path (never emitted)
constexpr int kShortHash = 9;               // per-wave hash set of 512 (bitset word index + 1, bits) pairs
// The medium-path kernel: eight waves per workgroup, each with a hash set of 2048 entries, for paths
// whose run count (known to the plan) fits it.
constexpr int kMediumHash = 11, kMediumWaves = 8;
constexpr uint32_t kMediumRuns = 1500;
// Segment-range passes for graphs whose bitset does not fit LDS (see fast_plan_create).
constexpr uint32_t kRangeSegs = 253u * 4096u;  // < kDummyBase; its bitset (129.5 KB) fits beside the queues
constexpr uint32_t kMaxPasses = 6;
constexpr int kAccThreads = 1024;
constexpr uint32_t kMaxSlots = 1024;  // sub-buckets per window (= workgroups of k_scan) k_accum can stage

// diagnostic ablations (FLATGFA_DEBUG_SKIP, results are then wrong by construction)
constexpr uint32_t kDbgNoStore = 1, kDbgNoBitset = 4, kDbgNoTiles = 8, kDbgHotLoads = 16, kDbgTime = 32;
// the ablation checks exist only in the DBG instantiation of the kernel
#define FGFA_SKIP(bit) (DBG && (A.dbg & (bit)))

struct ScanArgs {
    const uint32_t *steps;
    uint4 *items;        // work items, longest first: {begin, end, piece slot or kNoSlot, path}; room behind the
                         // first n_items for the short paths k_scan_short hands back (counted in *work_counter)
    const uint4 *short_items;  // paths of at most kShortMax steps, longest first
    uint32_t n_short;
    uint32_t *piece_bits;  // [n_piece_slots][n_words]: "seen" bitsets of the pieces of split paths
    uint32_t n_items, n_segs, n_win, n_words, n_slots;
    // Segment-range passes (graphs whose bitset does not fit LDS): this launch covers segments
    // [seg_lo, seg_lo + seg_n) only, renumbered from 0; n_segs stays the graph's segment count.
    uint32_t ranged, seg_lo, seg_n;
    uint32_t *work_counter;
    uint32_t *counts;   // [n_win][n_slots]
    uint32_t *buckets;  // [n_win + 1][n_slots][cap]; window n_win is a write sink
    uint32_t cap;
    uint32_t stride;    // n_slots * cap: elements between consecutive windows (< 2^30 in total)
    uint32_t sink;      // n_win * stride
    int *ovf_d;
    int *ovf_u;
    uint32_t *ovf_flag;
    uint32_t *status;
    uint32_t dbg;
};

// A record whose sub-bucket is full: apply it to the global difference array instead.
// `n` = segments of this launch: the -1 is dropped where k_accum would not read (and clear) it --
// past the window, and past the last segment (where a later segment-range pass would find it).
__device__ __noinline__ void overflow_apply(int *arr, uint32_t *flag, uint32_t id, uint32_t len, uint32_t n) {
    flag[id >> kWinBits] = 1u;
    atomicAdd(&arr[id], 1);
    if ((id & (kWin - 1)) + len < kWin && id + len < n) atomicAdd(&arr[id + len], -1);
}
// kind: 0 = depth, 1 = uniq, 2 = both
__device__ __forceinline__ void overflow_record(const ScanArgs &A, uint32_t id, uint32_t len, uint32_t kind) {
    if (kind != 1u) overflow_apply(A.ovf_d, A.ovf_flag, id, len, A.seg_n);  // by value: A stays in SGPRs
    if (kind != 0u) overflow_apply(A.ovf_u, A.ovf_flag, id, len, A.seg_n);
}

// Per-wave state: the run queue (LDS, kQCap entries of (start id << 11) | (len - 1)), how many
// entries it holds, the id of the step before the next one, and the start id of the run that
// step belongs to; and the queue of partially new claims (see emit_chunk).  `fill`, `pfill`,
// `prev`, `rs` and `vm` are wave-uniform.
struct Wave {
    uint32_t *q, *pq;
    uint32_t fill, pfill, prev, rs;
    uint32_t vm[2];  // memory instructions issued since the loads into landing set 0 / 1 (see wait_block)
    unsigned long long tacc[8], tlast;  // kDbgTime (diagnostic): cycles per phase of this wave
    int lane;
};

// Store a record at slot `pos` of this workgroup's sub-bucket of window (id >> 12).  Branch
// free: lanes with nothing to store (or no room) write to the sink window.  Returns whether
// the record still has to take the overflow route.
template <bool DBG>
__device__ __forceinline__ bool put(const ScanArgs &A, Wave &w, uint32_t *mine, bool e, uint32_t pos, uint32_t id,
                                    uint32_t lenm1, uint32_t kind) {
    const bool ok = e && pos < A.cap;
    // The bucket array holds fewer than 2^30 records, so a 32-bit byte offset from a uniform base
    // suffices.  window * stride + pos as one full-rate 24-bit multiply-add (the plan keeps the
    // stride below 2^24; hipcc would otherwise pick the quarter-rate 64-bit mad).
    uint32_t slot;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(slot) : "v"(id >> kWinBits), "s"(A.stride), "v"(pos));
    const uint32_t boff = (ok ? slot : A.sink) << 2;
    if (!FGFA_SKIP(kDbgNoStore)) {
        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(mine) + boff) =
            (id & (kWin - 1)) | (lenm1 << kWinBits) | ((kind + 1u) << 24);  // bit 24: counts for depth, bit 25: for uniq
        w.vm[0] += 1;  // exactly one store instruction, executed by the whole wave
        w.vm[1] += 1;
    }
    return e && !ok;
}

__device__ __forceinline__ uint32_t clamp_id(const ScanArgs &A, uint32_t id) {
    if (id >= A.n_segs) {
        *A.status = 1u;
        return 0u;
    }
    return id;
}

// kDbgTime: charge the cycles since the last mark to phase `ph`
template <bool DBG>
__device__ __forceinline__ void tmark(const ScanArgs &A, Wave &w, int ph) {
    if (DBG && (A.dbg & kDbgTime)) {
        const unsigned long long t = __builtin_readcyclecounter();
        w.tacc[ph] += t - w.tlast;
        w.tlast = t;
    }
}

// Ranged launches: a segment id becomes its offset in the launch's range, or -- outside the range --
// a placeholder that continues from the placeholder of the step before (so that a stretch of
// outside steps is one run), which is dropped when emitted.  `pos` is the step's index.
__device__ __forceinline__ uint32_t map_id(const ScanArgs &A, uint32_t id, uint32_t pos) {
    if (id >= A.n_segs) {
        *A.status = 1u;
        id = A.seg_lo;
    }
    const uint32_t rel = id - A.seg_lo;
    return rel < A.seg_n ? rel : kDummyBase + (pos & 0xFFFFu);
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ void enqueue(Wave &w, bool e, uint32_t rec) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(e);
    if (e) w.q[w.fill + lane_rank(m)] = rec;
    w.fill += (uint32_t)__builtin_popcountll(m);
}

constexpr uint32_t kQCap = 320;  // 63 left over + up to 256 from four steps of every lane
constexpr uint32_t kPCap = 96;   // parked claims (two words each): 31 left over + up to 64 from one chunk

__device__ __forceinline__ void push_partial(Wave &w, bool e, uint32_t word, uint32_t bits) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(e);
    if (e) reinterpret_cast<uint2 *>(w.pq)[w.pfill + lane_rank(m)] = make_uint2(word, bits);
    w.pfill += (uint32_t)__builtin_popcountll(m);
}

// Emit up to 64 queued runs, one per lane.  Each run becomes one depth record.  For unique
// depth the lane claims the run's segments in the path's "seen" bitset with ONE returning LDS OR
// (runs are cut at multiples of 32, so a run lies inside one word): the bits that were still
// clear are exactly the (path, segment) pairs this run is the first to touch.  If all of them
// were clear the depth record doubles as the uniq record (kind 2), if none was there is nothing
// to add.  The rare claim that is partly new is parked, as (word index, new bits), on a second
// queue; that queue is turned into uniq records 32..64 entries at a time, so its bit-stretch
// loop runs with most lanes busy instead of once per chunk for a lane or two.
// The short-path kernel's claim: the path's "seen" words live in a small per-wave hash set (open
// addressing, keyed by word index + 1) instead of a bitset over all segments.  A path is only
// walked this way when it has at most kQCap runs, so the set never holds more than kQCap of
// its 512 entries.
template <int BITS>
__device__ __forceinline__ uint32_t claim_hashed(const ScanArgs &A, uint32_t *tab, bool valid, uint32_t word, uint32_t mask) {
    uint32_t h = (word * 0x9E3779B1u) >> (32 - BITS);
    bool todo = valid;
    uint32_t old = 0, probes = 0;
    while (__builtin_amdgcn_ballot_w64(todo)) {
        if (++probes > (1u << BITS)) {  // cannot happen while the plan matches the steps: the set would be full
            *A.status = 1u;
            break;
        }
        if (todo) {
            uint32_t *e = tab + 2u * h;
            const uint32_t k = atomicCAS(e, 0u, word + 1u);
            if (k == 0u || k == word + 1u) {
                old = atomicOr(e + 1, mask);
                todo = false;
            } else {
                h = (h + 1u) & ((1u << BITS) - 1u);
            }
        }
    }
    return old;
}

template <bool UNIQ, bool DBG, int HASH = 0>
__device__ __forceinline__ void emit_chunk(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine,
                                           bool valid, uint32_t rec) {
    const uint32_t id = rec >> kRunBits, lenm1 = rec & (kRunSpan - 1), win = id >> kWinBits;
    if (HASH || A.ranged) valid = valid && id < kDummyBase;  // runs of placeholder ids are dropped here
    uint32_t kind = 0, pos;
    if (UNIQ && !FGFA_SKIP(kDbgNoBitset)) {
        const uint32_t mask = valid ? (0xFFFFFFFFu >> (31u - lenm1)) << (id & 31u) : 0u;
        const uint32_t old = HASH ? claim_hashed<HASH ? HASH : 1>(A, seen, valid, id >> 5, mask) : (mask ? atomicOr(&seen[id >> 5], mask) : 0u);
        pos = valid ? atomicAdd(&bcur[win], 1u) : 0u;  // both LDS round trips in flight together
        const uint32_t nb = mask & ~old;
        kind = (nb == mask) ? 2u : 0u;
        push_partial(w, (nb != mask) & (nb != 0u), id >> 5, nb);
    } else {
        pos = valid ? atomicAdd(&bcur[win], 1u) : 0u;
    }
    const bool o0 = put<DBG>(A, w, mine, valid, pos, id, lenm1, kind);
    if (__builtin_amdgcn_ballot_w64(o0) && o0) overflow_record(A, id, lenm1 + 1, kind);  // rare: the sub-bucket is full
}

// Turn parked claims into uniq records, one per stretch of new bits: the newest 64 while at
// least 32 are parked (all of them when `all`).
template <bool DBG>
__device__ __forceinline__ void drain_partial(const ScanArgs &A, Wave &w, uint32_t *bcur, uint32_t *mine, bool all) {
    while (w.pfill >= 32u || (all && w.pfill)) {
        const uint32_t n = min(w.pfill, 64u);
        w.pfill -= n;
        const bool valid = (uint32_t)w.lane < n;
        const uint2 ent = valid ? reinterpret_cast<const uint2 *>(w.pq)[w.pfill + w.lane] : make_uint2(0u, 0u);
        const uint32_t base = ent.x << 5;
        const uint32_t win = base >> kWinBits;
        uint32_t m = ent.y;  // never all ones: that claim would have been entirely new
        while (__builtin_amdgcn_ballot_w64(m != 0u)) {
            const bool e = m != 0u;
            const uint32_t tz = e ? (uint32_t)__builtin_ctz(m) : 0u;
            const uint32_t run = (uint32_t)__builtin_ctz(~(m >> tz));
            m &= ~(((1u << run) - 1u) << tz);
            const uint32_t p = e ? atomicAdd(&bcur[win], 1u) : 0u;
            const bool o1 = put<DBG>(A, w, mine, e, p, base + tz, run - 1u, 1u);
            if (__builtin_amdgcn_ballot_w64(o1) && o1) overflow_record(A, base + tz, run, 1u);
        }
    }
}

// Two full chunks at once, two runs per lane: the same as emit_chunk twice, but with all four LDS
// round trips (two claims, two cursors) in flight together, so that a wave waits once instead of
// twice for every 128 runs.
template <bool UNIQ, bool DBG>
__device__ __forceinline__ void emit_pair(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine,
                                          uint32_t rec0, uint32_t rec1) {
    const uint32_t id0 = rec0 >> kRunBits, l0 = rec0 & (kRunSpan - 1), win0 = id0 >> kWinBits;
    const uint32_t id1 = rec1 >> kRunBits, l1 = rec1 & (kRunSpan - 1), win1 = id1 >> kWinBits;
    uint32_t kind0 = 0, kind1 = 0, pos0, pos1;
    if (UNIQ && !FGFA_SKIP(kDbgNoBitset)) {
        const uint32_t mask0 = (0xFFFFFFFFu >> (31u - l0)) << (id0 & 31u);
        const uint32_t mask1 = (0xFFFFFFFFu >> (31u - l1)) << (id1 & 31u);
        const uint32_t old0 = atomicOr(&seen[id0 >> 5], mask0);  // LDS operations of one wave execute in order,
        const uint32_t old1 = atomicOr(&seen[id1 >> 5], mask1);  // so a lane's second claim sees its first
        pos0 = atomicAdd(&bcur[win0], 1u);
        pos1 = atomicAdd(&bcur[win1], 1u);
        const uint32_t nb0 = mask0 & ~old0, nb1 = mask1 & ~old1;
        kind0 = (nb0 == mask0) ? 2u : 0u;
        kind1 = (nb1 == mask1) ? 2u : 0u;
        push_partial(w, (nb0 != mask0) & (nb0 != 0u), id0 >> 5, nb0);
        drain_partial<DBG>(A, w, bcur, mine, false);  // keeps the parked-claim queue within its 96 entries
        push_partial(w, (nb1 != mask1) & (nb1 != 0u), id1 >> 5, nb1);
    } else {
        pos0 = atomicAdd(&bcur[win0], 1u);
        pos1 = atomicAdd(&bcur[win1], 1u);
    }
    const bool o0 = put<DBG>(A, w, mine, true, pos0, id0, l0, kind0);
    const bool o1 = put<DBG>(A, w, mine, true, pos1, id1, l1, kind1);
    if (__builtin_amdgcn_ballot_w64(o0 | o1)) {  // rare: a sub-bucket is full
        if (o0) overflow_record(A, id0, l0 + 1, kind0);
        if (o1) overflow_record(A, id1, l1 + 1, kind1);
    }
}

// Emit the newest 64 queued runs while at least 64 are queued (all of them when `all`).
template <bool UNIQ, bool DBG, int HASH = 0>
__device__ __forceinline__ void drain(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine, bool all) {
    while (!HASH && !A.ranged && w.fill >= 128u) {
        w.fill -= 128u;
        const uint32_t rec0 = w.q[w.fill + w.lane], rec1 = w.q[w.fill + 64u + w.lane];
        emit_pair<UNIQ, DBG>(A, w, seen, bcur, mine, rec0, rec1);
        if (UNIQ) drain_partial<DBG>(A, w, bcur, mine, false);
    }
    while (w.fill >= 64u || (all && w.fill)) {
        const uint32_t n = min(w.fill, 64u);
        w.fill -= n;
        const bool valid = (uint32_t)w.lane < n;
        const uint32_t rec = valid ? w.q[w.fill + w.lane] : 0u;
        emit_chunk<UNIQ, DBG, HASH>(A, w, seen, bcur, mine, valid, rec);
        if (UNIQ) drain_partial<DBG>(A, w, bcur, mine, false);
    }
    if (UNIQ && all) drain_partial<DBG>(A, w, bcur, mine, true);
}

// LDS byte address of a pointer into the workgroup's shared memory
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(uint32_t *p) { return (uint32_t)(uintptr_t)(lds_u32 *)p; }

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
// lanes where a run starts (mask Mj) append (cur << 11) | (id before step j - cur) at their queue
// cursor `p` and make step j's id their `cur`.  Written as asm so that each step is one scalar
// instruction (exec = lanes that start a run) and five vector ones, with no branches; exec is
// restored before the statement ends.  PM is the id before step 0.
#define FGFA_PASSB_STEP(T, PMJ, XJ, MJ)                \
    "s_and_b64 exec, %[sv], %[" MJ "]\n\t"             \
    "v_sub_u32 %[" T "], %[" PMJ "], %[cur]\n\t"       \
    "v_lshl_or_b32 %[" T "], %[cur], 11, %[" T "]\n\t" \
    "ds_write_b32 %[p], %[" T "]\n\t"                  \
    "v_add_u32 %[p], 4, %[p]\n\t"                      \
    "v_mov_b32 %[cur], %[" XJ "]\n\t"
#define FGFA_PASSB8(CUR, P, PM, X0, X1, X2, X3, X4, X5, X6, X7, M0, M1, M2, M3, M4, M5, M6, M7)                      \
    do {                                                                                                             \
        unsigned long long sv_;                                                                                      \
        uint32_t t0_, t1_;                                                                                           \
        asm volatile("s_mov_b64 %[sv], exec\n\t" FGFA_PASSB_STEP("t0", "pm", "x0", "m0")                             \
                         FGFA_PASSB_STEP("t1", "x0", "x1", "m1") FGFA_PASSB_STEP("t0", "x1", "x2", "m2")             \
                             FGFA_PASSB_STEP("t1", "x2", "x3", "m3") FGFA_PASSB_STEP("t0", "x3", "x4", "m4")         \
                                 FGFA_PASSB_STEP("t1", "x4", "x5", "m5") FGFA_PASSB_STEP("t0", "x5", "x6", "m6")     \
                                     FGFA_PASSB_STEP("t1", "x6", "x7", "m7") "s_mov_b64 exec, %[sv]"                 \
                     : [cur] "+v"(CUR), [p] "+v"(P), [sv] "=&s"(sv_), [t0] "=&v"(t0_), [t1] "=&v"(t1_)               \
                     : [pm] "v"(PM), [x0] "v"(X0), [x1] "v"(X1), [x2] "v"(X2), [x3] "v"(X3), [x4] "v"(X4),           \
                       [x5] "v"(X5), [x6] "v"(X6), [x7] "v"(X7), [m0] "s"(M0), [m1] "s"(M1), [m2] "s"(M2),           \
                       [m3] "s"(M3), [m4] "s"(M4), [m5] "s"(M5), [m6] "s"(M6), [m7] "s"(M7)                          \
                     : "memory", "scc");                                                                             \
    } while (0)

// inclusive prefix sum across the wave
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, true);
    return x;
}

// One block: 1024 consecutive steps of a wave's span, sixteen per lane (lane l holds steps
// 16l..16l+15, i.e. its own 64 bytes), so that fifteen of every sixteen run boundaries are found
// with in-lane compares.  Only the first `nl` lanes hold steps (nl < 64 for the last, partial
// block of a span).
//
// A run ends wherever the next one starts, and that is where its (start, length) is queued.
// Pass A marks the starts and counts them per lane; a wave prefix sum gives every lane its own
// stretch of the run queue.  Pass B then walks the sixteen steps again and each lane appends
// its runs to its stretch: no ballot or lane ranking per step.  The run that is in progress when
// a lane's first step arrives started in a lane below; its start id is fetched afterwards (one
// ballot + ds_bpermute per block) and patched into the lane's first queue entry.
// When the block has more starts than the queue has room for (dense: few steps continue a run),
// the steps are queued four at a time with the queue emitted in between.
// In the wave-per-path kernels (HASH = log2 of the hash set's entries) a block may reach beyond its
// path at either end (it starts and ends on 64-byte boundaries): steps at block-relative positions
// outside [rel_lo, rel_hi) get consecutive placeholder ids, whose runs are dropped when emitted.
// With QONLY the block's runs are only queued, never emitted; the return value says whether they
// fitted the queue.
template <bool UNIQ, bool DBG, int HASH = 0, bool QONLY = (HASH != 0)>
__device__ __forceinline__ bool block16(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine,
                                        uint32_t (&a)[16], uint32_t nl, uint32_t rel_lo = 0, uint32_t rel_hi = 1024,
                                        uint32_t blk_pos = 0) {
    const bool active = (uint32_t)w.lane < nl;
    const bool last_lane = (uint32_t)w.lane + 1u == nl;
    if (HASH && (rel_lo > 0u || rel_hi < 16u * nl)) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t rel = 16u * (uint32_t)w.lane + (uint32_t)k;
            const bool inside = rel >= rel_lo && rel < rel_hi;
            a[k] = !inside ? kDummyBase + ((blk_pos + rel) & 0xFFFFu) : A.ranged ? map_id(A, a[k], blk_pos + rel) : clamp_id(A, a[k]);
        }
    } else if (A.ranged) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = map_id(A, a[k], blk_pos + 16u * (uint32_t)w.lane + (uint32_t)k);
    } else {
        uint32_t mx = a[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) mx = max(mx, a[k]);
        if (mx >= A.n_segs) {
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = clamp_id(A, a[k]);
        }
    }
    // a block is walked on its own: its first step opens a run, its last step closes one
    const uint32_t first = __builtin_amdgcn_readfirstlane(a[0]);
    const uint32_t prev = __builtin_amdgcn_update_dpp(0u, a[15], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    // pass A (lanes beyond `nl` compute garbage flags; they are kept out of `cnt` and of pass B)
    unsigned long long m[16];
    uint32_t cnt = 0;
    FGFA_PASSA8(kCutMask<UNIQ>, cnt, prev, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
    FGFA_PASSA8(kCutMask<UNIQ>, cnt, a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
    cnt -= (w.lane == 0) ? (uint32_t)(m[0] & 1ull) : 0u;  // nothing ends at the block's first step
    m[0] &= ~1ull;
    cnt = active ? cnt : 0u;
    const uint32_t slots = cnt + (last_lane ? 1u : 0u);  // the last lane also queues the run that is open at the end
    const uint32_t incl = wave_scan_incl(slots);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    const unsigned long long below = __builtin_amdgcn_ballot_w64(cnt != 0u) & ((1ull << w.lane) - 1ull);
    const int src = below ? 63 - __builtin_clzll(below) : w.lane;
    if (w.fill + total <= kQCap) {
        // pass B, lane-local: `cur` is the start of the run in progress, 0 standing in for the
        // one that entered the lane (then the entry holds just the run's last id until patched)
        uint32_t *const p0 = w.q + w.fill + (incl - slots);
        uint32_t cur = 0u;
        uint32_t p = lds_addr(p0);
        if (active) {
            FGFA_PASSB8(cur, p, prev, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
            FGFA_PASSB8(cur, p, a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
        }
        // start id of the run in progress when this lane's first step arrived: the last start below
        const uint32_t from_below = __shfl(cur, src, 64);
        const uint32_t rs = below ? from_below : first;
        if (cnt) {
            const uint32_t last = *p0;
            *p0 = (rs << kRunBits) | (last - rs);
        } else {
            cur = rs;
        }
        if (last_lane) *reinterpret_cast<lds_u32 *>((uintptr_t)p) = (cur << kRunBits) | (a[15] - cur);
        w.fill += total;
        if (QONLY) return true;
        tmark<DBG>(A, w, 2);
        drain<UNIQ, DBG, HASH>(A, w, seen, bcur, mine, false);
        tmark<DBG>(A, w, 3);
    } else {
        if (QONLY) return false;
        bool st[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("" : "+s"(m[k]));  // keeps this path's work from being hoisted above the branch
            st[k] = __builtin_amdgcn_inverse_ballot_w64(m[k]) & active;
        }
        uint32_t last_start = a[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) last_start = st[k] ? a[k] : last_start;
        const uint32_t from_below = __shfl(last_start, src, 64);
        const uint32_t rs = below ? from_below : first;
        uint32_t cur = rs;
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            uint32_t pm, x0, x1, x2, x3;
            bool s0, s1, s2, s3;
            switch (g) {  // wave-uniform: one copy of the queueing and emitting code for all four groups
                case 0: pm = prev, x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3], s0 = st[0], s1 = st[1], s2 = st[2], s3 = st[3]; break;
                case 1: pm = a[3], x0 = a[4], x1 = a[5], x2 = a[6], x3 = a[7], s0 = st[4], s1 = st[5], s2 = st[6], s3 = st[7]; break;
                case 2: pm = a[7], x0 = a[8], x1 = a[9], x2 = a[10], x3 = a[11], s0 = st[8], s1 = st[9], s2 = st[10], s3 = st[11]; break;
                default: pm = a[11], x0 = a[12], x1 = a[13], x2 = a[14], x3 = a[15], s0 = st[12], s1 = st[13], s2 = st[14], s3 = st[15]; break;
            }
            enqueue(w, s0, (cur << kRunBits) | (pm - cur));
            cur = s0 ? x0 : cur;
            enqueue(w, s1, (cur << kRunBits) | (x0 - cur));
            cur = s1 ? x1 : cur;
            enqueue(w, s2, (cur << kRunBits) | (x1 - cur));
            cur = s2 ? x2 : cur;
            enqueue(w, s3, (cur << kRunBits) | (x2 - cur));
            cur = s3 ? x3 : cur;
            drain<UNIQ, DBG, HASH>(A, w, seen, bcur, mine, false);
        }
        enqueue(w, last_lane, (cur << kRunBits) | (a[15] - cur));
    }
    return true;
}

// Up to 64 consecutive steps, one per lane (heads, tails and short spans).
template <bool UNIQ>
__device__ __forceinline__ void tile_narrow(const ScanArgs &A, Wave &w, uint64_t t, uint32_t count, bool fresh) {
    const bool valid = (uint32_t)w.lane < count;
    const uint32_t raw = valid ? A.steps[t + w.lane] >> 1 : 0u;
    const uint32_t id = !valid ? 0u : A.ranged ? map_id(A, raw, (uint32_t)t + (uint32_t)w.lane) : clamp_id(A, raw);
    if (fresh) {  // as in block16
        w.rs = __builtin_amdgcn_readfirstlane(id);
        w.prev = w.rs - 1u;
    }
    uint32_t prev = __builtin_amdgcn_update_dpp(0u, id, 0x138, 0xf, 0xf, false);
    if (w.lane == 0) prev = w.prev;
    const bool s = valid && ((id != prev + 1) | ((id & kCutMask<UNIQ>) == 0));
    const unsigned long long m = __builtin_amdgcn_ballot_w64(s);
    const unsigned long long below = m & ((1ull << w.lane) - 1ull);
    const int src = below ? 63 - __builtin_clzll(below) : w.lane;
    const uint32_t from_below = __shfl(id, src, 64);
    const uint32_t rs = below ? from_below : w.rs;
    const uint32_t len = prev - rs + 1;
    enqueue(w, s && len != 0, (rs << kRunBits) | (len - 1));
    w.prev = __shfl(id, (int)count - 1, 64);
    if (m) w.rs = __shfl(id, 63 - __builtin_clzll(m), 64);
}

// How one work item (a path, or a piece of a long one) is cut up: the few steps [b, t0) before
// the first 64-byte boundary, then `nblk` blocks of 1024 steps starting at t0, the last of which
// may hold only `nl_last` lanes' worth of 16-step chunks, then fewer than 16 steps [tail, e).
// Blocks are not assigned to waves in advance: a wave takes the next free one from an LDS counter
// whenever one of its landing sets is free (its first three are its own index plus 0, 16 and 32).
// The SIMDs favour their older waves, so with equal fixed shares the youngest four waves finished
// a path up to 25% after the oldest four, which then idled at the barrier.
struct Item {
    uint64_t b, e, t0, tail;
    uint32_t nblk, nl_last;
    const uint4 *src;  // this lane's 64 bytes of block 0
    uint32_t slot;     // where to leave the bitset when this is a piece of a split path
};


// Which item a workgroup takes in its r-th turn.  Items are sorted longest first and dealt out
// in snake order (0..G-1, then G-1..0, ...), which balances a sorted list well and needs no
// queue: a returning global atomic per item sat on the critical path of every path.
__device__ __forceinline__ uint32_t item_of(uint32_t round, uint32_t wg, uint32_t n_wg) {
    return round * n_wg + ((round & 1u) ? n_wg - 1u - wg : wg);
}

__device__ __forceinline__ Item make_item(const ScanArgs &A, bool have, uint4 d, int lane) {
    Item it;
    it.b = it.e = it.t0 = it.tail = 0;
    it.nblk = 0;
    it.nl_last = 64;
    it.src = nullptr;
    it.slot = kNoSlot;
    if (have) {
        it.b = d.x;
        it.e = d.y;
        it.slot = d.z;
        const uint64_t up = (it.b + 15) & ~(uint64_t)15;
        it.t0 = up < it.e ? up : it.e;
        const uint64_t chunks = (it.e - it.t0) / 16;
        it.tail = it.t0 + chunks * 16;
        it.nblk = (uint32_t)((chunks + 63) / 64);
        it.nl_last = (chunks % 64) ? (uint32_t)(chunks % 64) : 64u;
        // kDbgHotLoads (diagnostic): every item reads the same cache-resident megabyte
        it.src = reinterpret_cast<const uint4 *>(A.steps + ((A.dbg & kDbgHotLoads) ? (it.t0 & 0x3FFF0u) : it.t0)) + lane * 4;
    }
    return it;
}

// Streaming loads of steps.  Two blocks per wave (8 KiB; 128 KiB per CU) are kept in flight
// across loop iterations.  hipcc cannot express that: it drains vmcnt to 0 at the top of the
// loop, and an inline-asm load into a compiler-allocated register is unsafe because the compiler
// may copy the register (to rotate it through the loop) while the load is still in flight.  So
// the landing registers are two fixed sets of sixteen, v[96:111] and v[112:127], which the
// compiler is told are clobbered and never otherwise allocates (the kernel needs < 96 VGPRs; 128
// is the budget of a 1024-thread workgroup).  tools/check_pinned_vgprs.py checks the generated
// ISA for exactly that (`make check`, and the CPU test suite).
// A block is taken out of its set, already shifted down to segment ids, by v_lshrrevs issued
// after a counted s_waitcnt (wait_block).  On gfx950 vmcnt counts loads and stores alike and they
// return in issue order (hipcc itself relies on that: it waits vmcnt(2) for a load followed by
// two stores), so the wait counts the record stores issued since, too -- otherwise every block
// would wait for the stores of the block before it to be acknowledged.
// The four loads of a lane cover its own 64 bytes; the wave's four instructions together cover
// 4 KiB, every 64-byte sector exactly once per instruction (measured at the same 5.9 TB/s as
// fully coalesced loads, tools/loadpat.hip).  No nontemporal hint here: the sectors must survive
// in cache from the first of the four instructions to the last.
#define FGFA_CLOB_A "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"
#define FGFA_CLOB_B "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
template <int SET>
__device__ __forceinline__ void load_block_async(Wave &w, const uint4 *p) {
    w.vm[SET] = 0;
    w.vm[1 - SET] += 4;
    if (SET == 0)
        asm volatile("global_load_dwordx4 v[96:99], %0, off\n\t"
                     "global_load_dwordx4 v[100:103], %0, off offset:16\n\t"
                     "global_load_dwordx4 v[104:107], %0, off offset:32\n\t"
                     "global_load_dwordx4 v[108:111], %0, off offset:48" ::"v"(p) : "memory", FGFA_CLOB_A);
    else
        asm volatile("global_load_dwordx4 v[112:115], %0, off\n\t"
                     "global_load_dwordx4 v[116:119], %0, off offset:16\n\t"
                     "global_load_dwordx4 v[120:123], %0, off offset:32\n\t"
                     "global_load_dwordx4 v[124:127], %0, off offset:48" ::"v"(p) : "memory", FGFA_CLOB_B);
}
// Waits until the loads into landing set SET have returned.  `w.vm[SET]` counts the memory
// instructions this wave is known to have issued since (the other set's loads and the record
// stores); they return in issue order, so the loads are back once at most that many operations
// are outstanding.  Rounded down to one of a few immediates; anything issued but not counted
// (rare paths) only makes the wait stricter.
template <int SET>
__device__ __forceinline__ void wait_block(const Wave &w) {
    const uint32_t n = w.vm[SET];
    if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#define FGFA_TAKE16(R0, R1, R2, R3, R4, R5, R6, R7, R8, R9, R10, R11, R12, R13, R14, R15)                              \
    asm volatile("v_lshrrev_b32 %0, 1, " R0 "\n\tv_lshrrev_b32 %1, 1, " R1 "\n\tv_lshrrev_b32 %2, 1, " R2                \
                 "\n\tv_lshrrev_b32 %3, 1, " R3 "\n\tv_lshrrev_b32 %4, 1, " R4 "\n\tv_lshrrev_b32 %5, 1, " R5            \
                 "\n\tv_lshrrev_b32 %6, 1, " R6 "\n\tv_lshrrev_b32 %7, 1, " R7 "\n\tv_lshrrev_b32 %8, 1, " R8            \
                 "\n\tv_lshrrev_b32 %9, 1, " R9 "\n\tv_lshrrev_b32 %10, 1, " R10 "\n\tv_lshrrev_b32 %11, 1, " R11        \
                 "\n\tv_lshrrev_b32 %12, 1, " R12 "\n\tv_lshrrev_b32 %13, 1, " R13 "\n\tv_lshrrev_b32 %14, 1, " R14      \
                 "\n\tv_lshrrev_b32 %15, 1, " R15                                                                       \
                 : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]),       \
                   "=v"(a[8]), "=v"(a[9]), "=v"(a[10]), "=v"(a[11]), "=v"(a[12]), "=v"(a[13]), "=v"(a[14]), "=v"(a[15])  \
                 :                                                                                                      \
                 : "memory")
template <int SET>
__device__ __forceinline__ void take_block(uint32_t (&a)[16]) {
    if (SET == 0) FGFA_TAKE16("v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
    else FGFA_TAKE16("v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

template <bool UNIQ, bool DBG>
__global__ __launch_bounds__(kThreads) void k_scan(const ScanArgs A) {
    extern __shared__ uint32_t lds[];
    // layout: [bcur: kMaxWin][run queues: kWaves * kQCap][parked-claim queues: kWaves * 2 * kPCap][seen: n_words]
    uint32_t *bcur = lds;
    uint32_t *seen = lds + kMaxWin + kWaves * (kQCap + 2 * kPCap);
    __shared__ uint32_t next_blk_cell;
    uint32_t *next_blk = &next_blk_cell;  // the next block of the current item nobody has taken yet
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the span math on the scalar unit
    uint32_t *mine = A.buckets + (size_t)blockIdx.x * A.cap;  // this workgroup's sub-bucket of window 0
    Wave w;
    w.q = lds + kMaxWin + wave * kQCap;
    w.pq = lds + kMaxWin + kWaves * kQCap + wave * (2 * kPCap);
    w.fill = w.pfill = 0;
    w.vm[0] = w.vm[1] = 0;
    for (int k = 0; k < 8; ++k) w.tacc[k] = 0;
    w.tlast = (DBG && (A.dbg & kDbgTime)) ? __builtin_readcyclecounter() : 0ull;
    w.lane = lane;
    // the cursors continue where k_scan_short (if it ran) left this workgroup's sub-buckets
    for (uint32_t i = threadIdx.x; i < kMaxWin; i += kThreads) bcur[i] = i < A.n_win ? A.counts[(size_t)i * A.n_slots + blockIdx.x] : 0u;
    if (UNIQ)
        for (uint32_t i = threadIdx.x; i < A.n_words; i += kThreads) seen[i] = 0u;
    if (threadIdx.x == 0) *next_blk = 3u * kWaves;
    __syncthreads();
    const uint32_t n_items = A.n_items + (A.n_short ? *A.work_counter : 0u);  // plus what k_scan_short handed back

    // The first blocks of an item are requested while the previous item is being wrapped up, and
    // its descriptor while the previous item is being walked.
    uint32_t round = 0;
    uint32_t job = item_of(0, blockIdx.x, gridDim.x);
    Item it = make_item(A, job < n_items, job < n_items ? A.items[job] : make_uint4(0u, 0u, 0u, 0u), lane);
    uint32_t blk[2];  // the block each landing set holds (or will hold next)
    uint32_t resv;    // the block this wave takes after those
    // lanes beyond a partial block's last one re-read lane 0's chunk: same instruction stream for all
#define FGFA_BLOCK_PTR(j) \
    (it.src + (size_t)(j) * 256 - (((j) + 1 == it.nblk && (uint32_t)lane >= it.nl_last) ? lane * 4 : 0))
#define FGFA_PRELOAD()                                                      \
    do {                                                                    \
        blk[0] = (uint32_t)wave;                                            \
        blk[1] = (uint32_t)wave + kWaves;                                   \
        resv = (uint32_t)wave + 2u * kWaves;                                \
        if (blk[0] < it.nblk) load_block_async<0>(w, FGFA_BLOCK_PTR(blk[0])); \
        if (blk[1] < it.nblk) load_block_async<1>(w, FGFA_BLOCK_PTR(blk[1])); \
    } while (0)
    // one block: wait for its data, take the next free block for its register set, process it
#define FGFA_BLOCK(SET)                                                                       \
    if (blk[SET] < it.nblk) {                                                                 \
        tmark<DBG>(A, w, 4);                                                                  \
        wait_block<SET>(w);                                                                   \
        tmark<DBG>(A, w, 0);                                                                  \
        uint32_t a[16];                                                                       \
        take_block<SET>(a);                                                                   \
        const uint32_t mine_now = blk[SET];                                                   \
        blk[SET] = resv;  /* taken one block ago, so that the LDS round trip is off this path */ \
        if (blk[SET] < it.nblk) load_block_async<SET>(w, FGFA_BLOCK_PTR(blk[SET]));           \
        uint32_t got = 0;                                                                     \
        if (lane == 0) got = atomicAdd(next_blk, 1u);                                         \
        if (!FGFA_SKIP(kDbgNoTiles)) {                                                        \
            block16<UNIQ, DBG>(A, w, seen, bcur, mine, a, mine_now + 1 == it.nblk ? it.nl_last : 64u, 0u, 1024u, \
                               (uint32_t)it.t0 + mine_now * 1024u);                           \
        } else if (a[0] == 0x3FFFFFFFu) {                                                     \
            *A.status = 2u;                                                                   \
        }                                                                                     \
        resv = __builtin_amdgcn_readfirstlane(got);                                           \
    }
    FGFA_PRELOAD();

    while (job < n_items) {
        const uint32_t next_job = item_of(++round, blockIdx.x, gridDim.x);
        const uint4 next_item = next_job < n_items ? A.items[next_job] : make_uint4(0u, 0u, 0u, 0u);  // needed after the barrier below
        // the few steps outside the blocks are walked on their own, by the first and the last wave
        if (wave == 0 && it.t0 > it.b) {
            tile_narrow<UNIQ>(A, w, it.b, (uint32_t)(it.t0 - it.b), true);
            enqueue(w, lane == 0, (w.rs << kRunBits) | (w.prev - w.rs));
        }
        if (wave == kWaves - 1 && it.e > it.tail) {
            tile_narrow<UNIQ>(A, w, it.tail, (uint32_t)(it.e - it.tail), true);
            enqueue(w, lane == 0, (w.rs << kRunBits) | (w.prev - w.rs));
        }
#pragma unroll 1
        while (blk[0] < it.nblk || blk[1] < it.nblk) {
            FGFA_BLOCK(0)
            FGFA_BLOCK(1)
        }
        drain<UNIQ, DBG>(A, w, seen, bcur, mine, true);
        // This wave is done with the item: it requests its first two blocks of the next one right
        // away, so that the waves' preloads do not all queue up behind the barrier.
        const uint32_t done_slot = it.slot;
        job = next_job;
        it = make_item(A, job < n_items, next_item, lane);
        FGFA_PRELOAD();
        tmark<DBG>(A, w, 4);
        __syncthreads();  // every wave is done with this path's bitset
        tmark<DBG>(A, w, 1);
        tmark<DBG>(A, w, 6);
        if (threadIdx.x == 0) *next_blk = 3u * kWaves;  // nobody takes a block before the barrier below
        if (UNIQ) {
            uint4 *sv = reinterpret_cast<uint4 *>(seen);
            if (done_slot != kNoSlot) {
                // a piece of a split path: other pieces may have claimed the same segments, so
                // the bitset is kept for k_accum to find the duplicates
                uint4 *dst = reinterpret_cast<uint4 *>(A.piece_bits + (size_t)done_slot * A.n_words);
                for (uint32_t i = threadIdx.x; i < A.n_words / 4; i += kThreads) dst[i] = sv[i];
            }
            for (uint32_t i = threadIdx.x; i < A.n_words / 4; i += kThreads) sv[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        tmark<DBG>(A, w, 7);
        __syncthreads();  // the bitset is clean, and the block counter set, before the next path starts
        tmark<DBG>(A, w, 5);
    }
    if (DBG && (A.dbg & kDbgTime) && lane == 0) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(A.status + 8);
        for (int k = 0; k < 8; ++k) atomicAdd(&acc[k], w.tacc[k]);
        atomicAdd(&acc[8 + wave], w.tacc[1]);  // barrier wait by wave index
    }
#undef FGFA_PRELOAD
#undef FGFA_BLOCK
#undef FGFA_BLOCK_PTR
    // publish how many records this workgroup left in each window's sub-bucket
    __syncthreads();
    for (uint32_t wdw = threadIdx.x; wdw < A.n_win; wdw += kThreads)
        A.counts[(size_t)wdw * A.n_slots + blockIdx.x] = bcur[wdw];
}

// ------------------------------------------------------- pass 1, short paths ---
//
// k_scan gives a whole workgroup to one path at a time, because the path's "seen" bitset fills the
// CU's LDS; a path of a thousand steps then costs two barriers, a 125 KB wipe and an exposed
// memory latency for one block of work.  Here every wave walks its own short paths: blocks as in
// k_scan (block16), but the runs of a path are only queued; when the path is complete they are
// emitted against a per-wave hash set of bitset words.  A path with more runs than the queue
// holds is handed back to k_scan (appended to its item list), which runs afterwards.

struct ShortBlk {
    uint32_t b, e;     // the path's steps
    uint32_t pos;      // first step of the block (a multiple of 16)
    uint32_t nl;       // lanes holding steps
    bool last, valid;  // last block of its path; there is a block at all
};

// The blocks of this wave's paths, in order.  The next path's descriptor is always requested one
// path ahead of its use.
struct ShortStream {
    uint32_t gi, stride, b, e, pos, end, nb, ne;
};
__device__ __forceinline__ void stream_fetch(const ScanArgs &A, ShortStream &g) {  // descriptor of path gi + stride
    const uint32_t nx = g.gi + g.stride;
    const uint4 d = nx < A.n_short && nx >= g.gi ? A.short_items[nx] : make_uint4(0u, 0u, 0u, 0u);
    g.nb = d.x;
    g.ne = d.y;
}
__device__ __forceinline__ ShortBlk stream_next(const ScanArgs &A, ShortStream &g) {
    ShortBlk k;
    k.valid = g.gi < A.n_short;
    k.b = g.b;
    k.e = g.e;
    k.pos = g.pos;
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
        stream_fetch(A, g);
    }
    return k;
}

// WAVES waves per workgroup, each with a hash set of 2^HASH entries.  QONLY: a path's runs are
// queued first and emitted when it is complete (short paths; those that do not fit are handed
// back); otherwise they are emitted as they come (medium paths, whose run count the plan knows).
template <bool UNIQ, int WAVES, int HASH, bool QONLY>
__global__ __launch_bounds__(WAVES * 64) void k_scan_short(const ScanArgs A) {
    constexpr uint32_t kTab = 1u << HASH;
    constexpr int kThr = WAVES * 64;
    extern __shared__ uint32_t lds[];
    // layout: [bcur: kMaxWin][run queues: WAVES * kQCap][parked-claim queues: WAVES * 2 * kPCap][hash sets: WAVES * 2 * kTab]
    uint32_t *bcur = lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *tab = lds + kMaxWin + WAVES * (kQCap + 2 * kPCap) + wave * (2 * kTab);
    uint32_t *mine = A.buckets + (size_t)blockIdx.x * A.cap;
    Wave w;
    w.q = lds + kMaxWin + wave * kQCap;
    w.pq = lds + kMaxWin + WAVES * kQCap + wave * (2 * kPCap);
    w.fill = w.pfill = 0;
    w.vm[0] = w.vm[1] = 0;
    w.tlast = 0;
    w.lane = lane;
    for (uint32_t i = threadIdx.x; i < kMaxWin; i += kThr) bcur[i] = i < A.n_win ? A.counts[(size_t)i * A.n_slots + blockIdx.x] : 0u;
    if (UNIQ)
        for (uint32_t i = lane; i < kTab / 2; i += 64) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    ShortStream g;
    g.stride = gridDim.x * WAVES;
    g.gi = blockIdx.x * WAVES + wave;
    {
        const uint4 d = g.gi < A.n_short ? A.short_items[g.gi] : make_uint4(0u, 0u, 0u, 0u);
        g.b = d.x;
        g.e = d.y;
        g.pos = g.b & ~15u;
        g.end = (g.e + 15u) & ~15u;
        stream_fetch(A, g);
    }
    ShortBlk slot[2];
    bool handed_back = false;  // the current path did not fit the run queue
    const uint4 *steps4 = reinterpret_cast<const uint4 *>(A.steps);
    // lanes beyond the last one holding steps re-read lane 0's chunk
#define FGFA_SPTR(K) (steps4 + (size_t)(K).pos / 4 + ((uint32_t)lane < (K).nl ? lane * 4 : 0))
    slot[0] = stream_next(A, g);
    if (slot[0].valid) load_block_async<0>(w, FGFA_SPTR(slot[0]));
    slot[1] = stream_next(A, g);
    if (slot[1].valid) load_block_async<1>(w, FGFA_SPTR(slot[1]));
#define FGFA_SBLOCK(SET)                                                                                \
    if (slot[SET].valid) {                                                                              \
        wait_block<SET>(w);                                                                             \
        uint32_t a[16];                                                                                 \
        take_block<SET>(a);                                                                             \
        const ShortBlk cur = slot[SET];                                                                 \
        slot[SET] = stream_next(A, g);                                                                  \
        if (slot[SET].valid) load_block_async<SET>(w, FGFA_SPTR(slot[SET]));                            \
        if (!handed_back) {                                                                             \
            const uint32_t lo = cur.b > cur.pos ? cur.b - cur.pos : 0u;                                 \
            const uint32_t hi = cur.e - cur.pos < 1024u ? cur.e - cur.pos : 1024u;                      \
            if (!block16<UNIQ, false, HASH, QONLY>(A, w, tab, bcur, mine, a, cur.nl, lo, hi, cur.pos)) { \
                handed_back = true;                                                                     \
                w.fill = 0;                                                                             \
            }                                                                                           \
        }                                                                                               \
        if (cur.last) {                                                                                 \
            if (handed_back) {                                                                          \
                if (lane == 0) A.items[A.n_items + atomicAdd(A.work_counter, 1u)] = make_uint4(cur.b, cur.e, kNoSlot, 0u); \
                handed_back = false;                                                                    \
            } else {                                                                                    \
                drain<UNIQ, false, HASH>(A, w, tab, bcur, mine, true);                                  \
                if (UNIQ)                                                                               \
                    for (uint32_t i = lane; i < kTab / 2; i += 64)                                      \
                        reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0u, 0u, 0u, 0u);                 \
            }                                                                                           \
        }                                                                                               \
    }
#pragma unroll 1
    while (slot[0].valid || slot[1].valid) {
        FGFA_SBLOCK(0)
        FGFA_SBLOCK(1)
    }
#undef FGFA_SBLOCK
#undef FGFA_SPTR
    __syncthreads();
    for (uint32_t wdw = threadIdx.x; wdw < A.n_win; wdw += kThr)
        A.counts[(size_t)wdw * A.n_slots + blockIdx.x] = bcur[wdw];
}

template <bool UNIQ>
constexpr auto k_walk_short = k_scan_short<UNIQ, kWaves, kShortHash, true>;
template <bool UNIQ>
constexpr auto k_walk_medium = k_scan_short<UNIQ, kMediumWaves, kMediumHash, false>;

// ------------------------------------------------------------------ pass 2 ---

struct AccArgs {
    uint32_t n_segs, n_win, n_slots, cap;
    // split paths (see fast_plan_create): the bitsets their pieces left behind
    const uint32_t *piece_bits;
    const uint2 *split;  // per split path: {first piece slot, number of pieces}
    uint32_t n_split, n_words;
    uint32_t *counts;
    const uint32_t *buckets;
    int *ovf_d;
    int *ovf_u;
    uint32_t *ovf_flag;
    uint32_t *work_counter;
    uint32_t *depth_out;
    uint32_t *uniq_out;
};

// One record = +1 at its first segment and -1 just past its last one, in a difference array over
// the window.  With unique depth the two difference arrays share one array of 64-bit cells, depth
// in the low word and uniq in the high word, so a record is two LDS atomics whatever it counts
// for: the packed cells add up as 64-bit integers (depth + uniq * 2^32, both signed), prefix-sum
// as such, and are taken apart only at the end.
template <bool UNIQ>
__device__ __forceinline__ void apply_record(long long *acc, uint32_t rec) {
    const uint32_t rel = rec & (kWin - 1), end = rel + ((rec >> kWinBits) & (kWin - 1)) + 1;  // end <= 4096; cell 4096 is a sink
    if (UNIQ) {
        const unsigned long long v = (unsigned long long)((rec >> 24) & 1u) | ((unsigned long long)((rec >> 25) & 1u) << 32);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[rel]), v);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[end]), 0ull - v);
    } else {
        int *dd = reinterpret_cast<int *>(acc);
        atomicAdd(&dd[rel], 1);
        atomicAdd(&dd[end], -1);
    }
}

// inclusive prefix sum of 4096 values held 4 per thread by 1024 threads; returns this thread's
// four prefix values.
template <typename T>
__device__ __forceinline__ void block_scan4(const T *arr, T *wave_tot, T (&v)[4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = arr[4 * tid + k];
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    T incl = v[3];
    for (int off = 1; off < 64; off <<= 1) {
        const T t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    T add = incl - v[3];
    for (int k = 0; k < wave; ++k) add += wave_tot[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += add;
}

__device__ __forceinline__ void store4(uint32_t *out, uint32_t i0, uint32_t nvalid, const uint32_t (&a)[4]) {
    if (i0 + 3 < nvalid) {
        *reinterpret_cast<uint4 *>(out + i0) = make_uint4(a[0], a[1], a[2], a[3]);
    } else {
        for (uint32_t k = 0; k < 4; ++k)
            if (i0 + k < nvalid) out[i0 + k] = a[k];
    }
}

template <bool UNIQ>
__global__ __launch_bounds__(kAccThreads) void k_accum(const AccArgs A) {
    // difference array over the window: packed 64-bit cells with unique depth, plain ints without
    __shared__ __attribute__((aligned(16))) long long cells[UNIQ ? kWin + 64 : (kWin + 64) / 2];
    __shared__ long long wave_tot[kAccThreads / 64];
    __shared__ uint32_t scnt[kMaxSlots];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t win = blockIdx.x, w0 = win * kWin;
    // this window's record counts, one per sub-bucket: staged in LDS, and zeroed in place so that
    // the scratch is clean for the next call
    for (uint32_t sl = tid; sl < A.n_slots; sl += kAccThreads) {
        uint32_t *c = A.counts + (size_t)win * A.n_slots + sl;
        scnt[sl] = min(*c, A.cap);
        *c = 0u;
    }
    const uint32_t nvalid = min(kWin, A.n_segs - w0);
    const bool ovf = A.ovf_flag[win] != 0;
    for (uint32_t i = tid; i < kWin + 64; i += kAccThreads) {
        int d0 = 0, u0 = 0;
        if (ovf && i < nvalid) {
            d0 = A.ovf_d[w0 + i];
            A.ovf_d[w0 + i] = 0;
            if (UNIQ) {
                u0 = A.ovf_u[w0 + i];
                A.ovf_u[w0 + i] = 0;
            }
        }
        if (UNIQ) cells[i] = (long long)d0 + ((long long)u0 << 32);
        else reinterpret_cast<int *>(cells)[i] = d0;
    }
    __syncthreads();
    if (ovf && tid == 0) A.ovf_flag[win] = 0u;
    if (UNIQ && A.n_split) {
        // A path longer than the piece length was scanned as several pieces by different
        // workgroups, each with its own bitset, so a segment touched by two pieces of one path
        // was counted twice in uniq.  Walk the pieces' bitsets for this window in order and take
        // one back for every bit an earlier piece of the same path already had.
        const uint32_t word = tid & (kWinWords - 1), grp = tid / kWinWords;  // 128 words x 8 paths at a time
        const uint32_t gw = win * kWinWords + word;
        if (gw < A.n_words) {
            for (uint32_t s = grp; s < A.n_split; s += kAccThreads / kWinWords) {
                const uint2 sp = A.split[s];
                const uint32_t *base = A.piece_bits + (size_t)sp.x * A.n_words + gw;
                uint32_t acc = 0;
                for (uint32_t k0 = 0; k0 < sp.y; k0 += 8) {
                    uint32_t b[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[j] = (k0 + j < sp.y) ? base[(size_t)(k0 + j) * A.n_words] : 0u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        uint32_t dup = b[j] & acc;
                        acc |= b[j];
                        while (dup) {
                            const uint32_t tz = __builtin_ctz(dup);
                            const uint32_t y = dup >> tz;
                            const uint32_t run = (y == 0xFFFFFFFFu) ? 32u : (uint32_t)__builtin_ctz(~y);
                            dup &= ~(((run == 32u) ? 0xFFFFFFFFu : ((1u << run) - 1u)) << tz);
                            atomicAdd(reinterpret_cast<unsigned long long *>(&cells[(word << 5) + tz]), 0ull - (1ull << 32));
                            atomicAdd(reinterpret_cast<unsigned long long *>(&cells[(word << 5) + tz + run]), 1ull << 32);
                        }
                    }
                }
            }
        }
    }
    // Drain the window's sub-buckets.  Their record counts were staged in LDS above; each wave
    // takes sixteen sub-buckets per round and requests the first 64 x 16 bytes of every one before
    // it applies any, so a round pays the memory latency once.
    constexpr uint32_t kAccWaves = kAccThreads / 64;
    constexpr int kPerRound = 16;
    const uint32_t uw = __builtin_amdgcn_readfirstlane(wave);  // wave-uniform: sub-bucket addressing stays scalar
    const uint32_t *wbase = A.buckets + (size_t)win * A.n_slots * A.cap;
    for (uint32_t s0 = uw; s0 < A.n_slots; s0 += kPerRound * kAccWaves) {
        uint4 r[kPerRound];
        uint32_t cnt[kPerRound];
#pragma unroll
        for (int k = 0; k < kPerRound; ++k) {
            const uint32_t s = s0 + k * kAccWaves;
            const uint32_t sc = s < A.n_slots ? s : 0u;
            cnt[k] = s < A.n_slots ? scnt[sc] : 0u;
            // unconditional (slot 0 always exists): a predicated load would be waited for on the spot
            r[k] = reinterpret_cast<const uint4 *>(wbase + sc * A.cap)[(uint32_t)lane < (cnt[k] >> 2) ? lane : 0];
        }
#pragma unroll
        for (int k = 0; k < kPerRound; ++k) {
            if ((uint32_t)lane < (cnt[k] >> 2)) {
                apply_record<UNIQ>(cells, r[k].x);
                apply_record<UNIQ>(cells, r[k].y);
                apply_record<UNIQ>(cells, r[k].z);
                apply_record<UNIQ>(cells, r[k].w);
            }
        }
        // what does not fit the first pass (skewed sub-buckets), and the last 1..3 records
#pragma unroll 1
        for (int k = 0; k < kPerRound; ++k) {
            const uint32_t s = s0 + k * kAccWaves;
            if (s >= A.n_slots) break;
            const uint32_t c = scnt[s];
            const uint32_t *bk = wbase + s * A.cap;
            for (uint32_t i = 64 + lane; i < (c >> 2); i += 64) {
                const uint4 v = reinterpret_cast<const uint4 *>(bk)[i];
                apply_record<UNIQ>(cells, v.x);
                apply_record<UNIQ>(cells, v.y);
                apply_record<UNIQ>(cells, v.z);
                apply_record<UNIQ>(cells, v.w);
            }
            const uint32_t rest = (c & ~3u) + lane;
            if (rest < c) apply_record<UNIQ>(cells, bk[rest]);
        }
    }
    __syncthreads();
    if (win == 0 && tid == 0) *A.work_counter = 0u;
    const uint32_t i0 = 4 * tid;
    uint32_t d[4], u[4];
    if (UNIQ) {
        long long v[4];
        block_scan4(cells, wave_tot, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[k] = (uint32_t)v[k];
            u[k] = (uint32_t)((v[k] - (long long)(int)d[k]) >> 32);  // depth may carry a sign into the high word
        }
        store4(A.depth_out + w0, i0, nvalid, d);
        store4(A.uniq_out + w0, i0, nvalid, u);
    } else {
        int v[4];
        block_scan4(reinterpret_cast<const int *>(cells), reinterpret_cast<int *>(wave_tot), v);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = (uint32_t)v[k];
        store4(A.depth_out + w0, i0, nvalid, d);
    }
}

// Plan time: how many runs (as k_scan cuts them: +1 continuations, cut at multiples of 32) each
// path has.  One workgroup per path at a time.
__global__ __launch_bounds__(256) void k_count_runs(const uint32_t *__restrict__ steps, const uint32_t *__restrict__ pb,
                                                     const uint32_t *__restrict__ pe, uint32_t n_paths,
                                                     uint32_t *__restrict__ runs) {
    __shared__ uint32_t total;
    for (uint32_t p = blockIdx.x; p < n_paths; p += gridDim.x) {
        if (threadIdx.x == 0) total = 0;
        __syncthreads();
        const uint64_t b = pb[p], e = pe[p];
        uint32_t mine = 0;
        for (uint64_t i = b + threadIdx.x; i < e; i += 256) {
            const uint32_t id = steps[i] >> 1;
            const bool start = i == b || id != (steps[i - 1] >> 1) + 1u || (id & 31u) == 0u;
            mine += start ? 1u : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&total, mine);
        __syncthreads();
        if (threadIdx.x == 0) runs[p] = total;
        __syncthreads();
    }
}

uint32_t scan_lds_bytes(uint32_t n_words) { return (kMaxWin + kWaves * (kQCap + 2 * kPCap) + n_words) * 4u; }

#define FAST_TRY(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            return false;                                                                   \
        }                                                                                   \
    } while (0)

}  // namespace

bool fast_plan_create(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, FastPlan *fp) {
    *fp = FastPlan();
    if (g.n_segs == 0 || g.n_paths == 0 || g.n_steps == 0) return true;
    if ((reinterpret_cast<uintptr_t>(g.steps) & 15u) != 0) return true;  // 16-byte step loads
    // The per-path "seen" bitset must fit one CU's LDS next to the queues.  A graph with more
    // segments is done in up to kMaxPasses passes, each over one range of kRangeSegs segments
    // (every pass reads all steps; beyond that the simple atomic kernels are no slower).
    uint32_t seg_range = g.n_segs, n_pass = 1;
    if (scan_lds_bytes(((g.n_segs + 31) / 32 + 3) & ~3u) + 64 > kLdsLimit || (g.n_segs + kWin - 1) / kWin > kMaxWin) {
        seg_range = kRangeSegs;
        n_pass = (g.n_segs + kRangeSegs - 1) / kRangeSegs;
        if (n_pass > kMaxPasses) return true;
        if (const char *off = getenv("FLATGFA_MAX_PASSES")) {
            if (n_pass > strtoul(off, nullptr, 10)) return true;
        }
    }
    const uint32_t n_win = (seg_range + kWin - 1) / kWin;
    const uint32_t n_words = ((seg_range + 31) / 32 + 3) & ~3u;
    if (n_win > kMaxWin || scan_lds_bytes(n_words) + 64 > kLdsLimit) return true;
    hipDeviceProp_t prop;
    int dev = 0;
    FAST_TRY(hipGetDevice(&dev));
    FAST_TRY(hipGetDeviceProperties(&prop, dev));
    fp->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    fp->n_slots = fp->n_cus;
    if (fp->n_slots > kMaxSlots) return true;
    fp->n_win = n_win;
    fp->n_pass = n_pass;
    fp->seg_range = seg_range;
    fp->n_words = n_words;
    fp->lds_bytes_uniq = scan_lds_bytes(n_words);
    fp->lds_bytes_depth = scan_lds_bytes(0);
    // Worst case is one depth record per step plus one uniq record per step; 25% headroom for
    // skew.  Record slots are addressed with 32-bit element offsets, so the whole bucket array
    // (one extra sink window included) must stay below 2^30 elements; beyond that the capacity
    // is trimmed and the overflow route absorbs the worst case.
    const uint64_t slots = (uint64_t)n_win * fp->n_slots;
    uint64_t cap = (2 * g.n_steps + slots - 1) / slots;
    cap = cap + cap / 4 + 256;
    const uint64_t max_cap = ((1ull << 30) - 1) / ((uint64_t)(n_win + 1) * fp->n_slots);
    cap = std::min(cap, max_cap);
    if (const char *forced = getenv("FLATGFA_BUCKET_CAP")) cap = strtoull(forced, nullptr, 10);  // tests: force overflow
    // put() addresses a slot as window * (n_slots * cap) + pos with a 24-bit multiply
    cap = std::min<uint64_t>(cap, ((1ull << 24) - 1) / fp->n_slots);
    cap = std::min<uint64_t>(std::max<uint64_t>(cap & ~3ull, 4), max_cap & ~3ull);
    if (cap < 4) return true;
    fp->cap = (uint32_t)cap;
    if (const char *d = getenv("FLATGFA_DEBUG_SKIP")) fp->dbg = (uint32_t)strtoul(d, nullptr, 10);
    // Work items: whole paths, except that a path longer than `piece` steps is cut into pieces so
    // that graphs with few long paths still fill the chip.  Pieces carry a slot for their bitset.
    uint64_t piece = std::max<uint64_t>(65536, (g.n_steps + 2ull * fp->n_cus - 1) / (2ull * fp->n_cus));
    if (const char *forced = getenv("FLATGFA_PIECE_STEPS")) piece = std::max<uint64_t>(256, strtoull(forced, nullptr, 10));
    piece = (piece + 255) & ~255ull;
    // Paths of at most `short_max` steps are walked by single waves (k_scan_short), unless their
    // last block would reach beyond the step array.
    uint64_t short_max = fp->dbg ? 0 : kShortMax;  // the ablation switches are k_scan's
    if (const char *forced = getenv("FLATGFA_SHORT_MAX")) short_max = std::min<uint64_t>(kShortMax, strtoull(forced, nullptr, 10));
    // Which kernel walks a path depends on how many runs it has: short paths must fit the run queue,
    // paths with at most kMediumRuns runs are walked by single waves too, eight per CU, each with a
    // bigger hash set (k_scan_short's medium variant).  The counts come from a one-off kernel.
    std::vector<uint32_t> runs;
    if (short_max) {
        uint32_t *d_runs = nullptr;
        FAST_TRY(hipMalloc(&d_runs, (size_t)g.n_paths * 4));
        hipLaunchKernelGGL(k_count_runs, dim3(std::min<uint32_t>(g.n_paths, fp->n_cus * 8u)), dim3(256), 0, nullptr, g.steps,
                           g.path_begin, g.path_end, g.n_paths, d_runs);
        runs.resize(g.n_paths);
        const hipError_t e = hipMemcpy(runs.data(), d_runs, (size_t)g.n_paths * 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_runs);
        FAST_TRY(e);
    }
    const bool short_any = getenv("FLATGFA_SHORT_ANY") != nullptr;  // tests: let k_scan_short find out and hand back
    std::vector<uint4> items, short_items, medium_items;
    std::vector<uint2> split;
    uint32_t n_piece_slots = 0;
    for (uint32_t p = 0; p < g.n_paths; ++p) {
        const uint64_t b = hb[p], e = he[p], n = e - b;
        if (n == 0) continue;
        const bool in_reach = ((e + 15) & ~15ull) <= g.n_steps;  // the last block must not read past the step array
        // a short path's runs (plus a few block-closing and placeholder ones) must fit the run queue
        if (n <= short_max && in_reach && (runs[p] + 16 <= kQCap || short_any)) {
            short_items.push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else if (short_max && in_reach && runs[p] <= kMediumRuns) {
            medium_items.push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else if (n <= piece) {
            items.push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else {
            const uint32_t k = (uint32_t)((n + piece - 1) / piece);
            split.push_back(make_uint2(n_piece_slots, k));
            for (uint32_t j = 0; j < k; ++j) {
                const uint64_t pb = b + n * j / k, pe = b + n * (j + 1) / k;
                items.push_back(make_uint4((uint32_t)pb, (uint32_t)pe, n_piece_slots + j, p));
            }
            n_piece_slots += k;
        }
    }
    const auto longer = [](const uint4 &a, const uint4 &b) { return a.y - a.x > b.y - b.x; };
    std::stable_sort(items.begin(), items.end(), longer);
    std::stable_sort(short_items.begin(), short_items.end(), longer);
    std::stable_sort(medium_items.begin(), medium_items.end(), longer);
    fp->n_items = (uint32_t)items.size();
    fp->n_short = (uint32_t)short_items.size();
    fp->n_medium = (uint32_t)medium_items.size();
    fp->n_split = (uint32_t)split.size();
    if (items.empty() && short_items.empty() && medium_items.empty()) return true;
    FAST_TRY(hipMalloc(&fp->counts, slots * 4));
    FAST_TRY(hipMemset(fp->counts, 0, slots * 4));
    FAST_TRY(hipMalloc(&fp->buckets, (slots + fp->n_slots) * cap * 4));
    FAST_TRY(hipMalloc(&fp->ovf_d, ((size_t)g.n_segs + 1) * 4));
    FAST_TRY(hipMalloc(&fp->ovf_u, ((size_t)g.n_segs + 1) * 4));
    FAST_TRY(hipMemset(fp->ovf_d, 0, ((size_t)g.n_segs + 1) * 4));
    FAST_TRY(hipMemset(fp->ovf_u, 0, ((size_t)g.n_segs + 1) * 4));
    FAST_TRY(hipMalloc(&fp->ovf_flag, (size_t)n_win * 4));
    FAST_TRY(hipMemset(fp->ovf_flag, 0, (size_t)n_win * 4));
    FAST_TRY(hipMalloc(&fp->items, (items.size() + short_items.size() + 1) * sizeof(uint4)));
    if (!items.empty()) FAST_TRY(hipMemcpy(fp->items, items.data(), items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (!short_items.empty()) {
        FAST_TRY(hipMalloc(&fp->short_items, short_items.size() * sizeof(uint4)));
        FAST_TRY(hipMemcpy(fp->short_items, short_items.data(), short_items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    if (!medium_items.empty()) {
        FAST_TRY(hipMalloc(&fp->medium_items, medium_items.size() * sizeof(uint4)));
        FAST_TRY(hipMemcpy(fp->medium_items, medium_items.data(), medium_items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    fp->lds_bytes_short = (kMaxWin + kWaves * (kQCap + 2 * kPCap + (2u << kShortHash))) * 4u;
    fp->lds_bytes_medium = (kMaxWin + kMediumWaves * (kQCap + 2 * kPCap + (2u << kMediumHash))) * 4u;
    FAST_TRY(hipFuncSetAttribute((const void *)k_walk_short<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp->lds_bytes_short));
    FAST_TRY(hipFuncSetAttribute((const void *)k_walk_short<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp->lds_bytes_short));
    FAST_TRY(hipFuncSetAttribute((const void *)k_walk_medium<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp->lds_bytes_medium));
    FAST_TRY(hipFuncSetAttribute((const void *)k_walk_medium<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp->lds_bytes_medium));
    if (n_piece_slots) {
        FAST_TRY(hipMalloc(&fp->piece_bits, (size_t)n_piece_slots * n_words * 4));
        FAST_TRY(hipMalloc(&fp->split, split.size() * sizeof(uint2)));
        FAST_TRY(hipMemcpy(fp->split, split.data(), split.size() * sizeof(uint2), hipMemcpyHostToDevice));
    }
    FAST_TRY(hipMalloc(&fp->work_counter, 256));
    FAST_TRY(hipMemset(fp->work_counter, 0, 256));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp->lds_bytes_uniq));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp->lds_bytes_uniq));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp->lds_bytes_depth));
    fp->eligible = true;
    return true;
}

void fast_plan_destroy(FastPlan *fp) {
    for (void *p : {(void *)fp->counts, (void *)fp->buckets, (void *)fp->ovf_d, (void *)fp->ovf_u, (void *)fp->ovf_flag,
                    (void *)fp->items, (void *)fp->short_items, (void *)fp->medium_items, (void *)fp->piece_bits, (void *)fp->split, (void *)fp->work_counter})
        if (p) (void)hipFree(p);
    *fp = FastPlan();
}

int fast_seg_depth(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                   uint32_t *status, hipStream_t stream) {
    const uint32_t stride = fp.n_slots * fp.cap;
    // one persistent workgroup per CU; k_scan may be handed short paths back, so it gets a full grid when there are any
    const uint32_t grid = fp.n_short ? fp.n_slots : std::min<uint32_t>(fp.n_items, fp.n_slots);
    for (uint32_t pass = 0; pass < fp.n_pass; ++pass) {
        const uint32_t seg_lo = pass * fp.seg_range, seg_n = std::min(fp.seg_range, g.n_segs - seg_lo);
        const uint32_t n_win = (seg_n + kWin - 1) / kWin, n_words = ((seg_n + 31) / 32 + 3) & ~3u;
        ScanArgs sa;
        sa.steps = g.steps;
        sa.items = reinterpret_cast<uint4 *>(fp.items);
        sa.short_items = reinterpret_cast<const uint4 *>(fp.short_items);
        sa.n_short = fp.n_short;
        sa.piece_bits = fp.piece_bits;
        sa.n_items = fp.n_items;
        sa.n_segs = g.n_segs;
        sa.n_win = n_win;
        sa.n_words = uniq_out ? n_words : 0u;
        sa.n_slots = fp.n_slots;
        sa.ranged = fp.n_pass > 1 ? 1u : 0u;
        sa.seg_lo = seg_lo;
        sa.seg_n = seg_n;
        sa.work_counter = fp.work_counter;
        sa.counts = fp.counts;
        sa.buckets = fp.buckets;
        sa.cap = fp.cap;
        sa.stride = stride;
        sa.sink = fp.n_win * stride;
        sa.ovf_d = fp.ovf_d;
        sa.ovf_u = fp.ovf_u;
        sa.ovf_flag = fp.ovf_flag;
        sa.status = status;
        sa.dbg = fp.dbg;
        AccArgs aa{seg_n, n_win, fp.n_slots, fp.cap, fp.piece_bits, reinterpret_cast<const uint2 *>(fp.split),
                   uniq_out ? fp.n_split : 0u, n_words, fp.counts, fp.buckets, fp.ovf_d, fp.ovf_u, fp.ovf_flag,
                   fp.work_counter, depth_out + seg_lo, uniq_out ? uniq_out + seg_lo : nullptr};
        const uint32_t lds_uniq = scan_lds_bytes(n_words);
        if (fp.n_short) {
            const uint32_t sgrid = std::min<uint32_t>((fp.n_short + kWaves - 1) / kWaves, fp.n_slots);
            ProfScope ps(uniq_out ? "k_scan_short<uniq>" : "k_scan_short<depth>", stream);
            if (uniq_out) hipLaunchKernelGGL(k_walk_short<true>, dim3(sgrid), dim3(kThreads), fp.lds_bytes_short, stream, sa);
            else hipLaunchKernelGGL(k_walk_short<false>, dim3(sgrid), dim3(kThreads), fp.lds_bytes_short, stream, sa);
        }
        if (fp.n_medium) {
            ScanArgs sm = sa;
            sm.short_items = reinterpret_cast<const uint4 *>(fp.medium_items);
            sm.n_short = fp.n_medium;
            const uint32_t mgrid = std::min<uint32_t>((fp.n_medium + kMediumWaves - 1) / kMediumWaves, fp.n_slots);
            ProfScope ps(uniq_out ? "k_scan_medium<uniq>" : "k_scan_medium<depth>", stream);
            if (uniq_out) hipLaunchKernelGGL(k_walk_medium<true>, dim3(mgrid), dim3(kMediumWaves * 64), fp.lds_bytes_medium, stream, sm);
            else hipLaunchKernelGGL(k_walk_medium<false>, dim3(mgrid), dim3(kMediumWaves * 64), fp.lds_bytes_medium, stream, sm);
        }
        if (uniq_out) {
            if (grid) {
                ProfScope ps("k_scan<uniq>", stream);
                if (fp.dbg) hipLaunchKernelGGL((k_scan<true, true>), dim3(grid), dim3(kThreads), lds_uniq, stream, sa);
                else hipLaunchKernelGGL((k_scan<true, false>), dim3(grid), dim3(kThreads), lds_uniq, stream, sa);
            }
            {
                ProfScope ps("k_accum<uniq>", stream);
                hipLaunchKernelGGL(k_accum<true>, dim3(n_win), dim3(kAccThreads), 0, stream, aa);
            }
        } else {
            if (grid) {
                ProfScope ps("k_scan<depth>", stream);
                hipLaunchKernelGGL((k_scan<false, false>), dim3(grid), dim3(kThreads), fp.lds_bytes_depth, stream, sa);
            }
            {
                ProfScope ps("k_accum<depth>", stream);
                hipLaunchKernelGGL(k_accum<false>, dim3(n_win), dim3(kAccThreads), 0, stream, aa);
            }
        }
    }
    if (hipGetLastError() != hipSuccess) {
        set_error("fast_seg_depth: kernel launch failed");
        return FLATGFA_ERR_HIP;
    }
    if (fp.dbg & kDbgTime) {  // diagnostic: where the waves of k_scan spend their cycles
        unsigned long long acc[8 + kWaves] = {};
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(acc, status + 8, sizeof acc, hipMemcpyDeviceToHost);
        (void)hipMemset(status + 8, 0, sizeof acc);
        const double waves = (double)grid * kWaves;
        fprintf(stderr, "k_scan cycles per wave (s_memtime ticks): wait_block %.0f  barrier_wait %.0f  passA+B %.0f  drain %.0f  other %.0f  next_item+wipe %.0f\n",
                acc[0] / waves, acc[1] / waves, acc[2] / waves, acc[3] / waves, acc[4] / waves, acc[5] / waves);
        fprintf(stderr, "  of next_item+wipe: make_item+preload %.0f  wipe %.0f (the rest is the second barrier)\n", acc[6] / waves, acc[7] / waves);
        fprintf(stderr, "  barrier wait by wave index:");
        for (int k = 0; k < kWaves; ++k) fprintf(stderr, " %.0f", (double)acc[8 + k] / grid);
        fprintf(stderr, "\n");
    }
    return FLATGFA_OK;
}

}  // namespace fgfa_dev
