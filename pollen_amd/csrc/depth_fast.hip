// The bucketed node-depth path for gfx950: two kernels, no global atomics on the data path.
//
//   k_scan   (pass 1)  persistent workgroups (one per CU), each pulling whole paths from a
//            queue.  Every wave streams a contiguous span of the path's steps with 16-byte
//            loads (4 handles per lane, 4 tiles in flight), detects maximal +1 runs of segment
//            ids, and turns each run into ONE range record (start id, length) instead of
//            `length` histogram updates.  For unique depth the path's "seen" bitset of
//            ops/depth.rs:23-34 lives in LDS (1 bit per segment); when the path ends, the
//            set-bit runs of the bitset become range records of a second kind and the bitset
//            is left zeroed.  A record goes straight to the bucket of its 4096-segment window:
//            buckets are split into one private sub-bucket per workgroup, so the append cursor
//            is an LDS counter and no global atomic is needed; each workgroup's writes stay on
//            its own XCD's L2 until the line is full.
//   k_accum  (pass 2)  one workgroup per window: applies the window's records as +1/-1 pairs
//            to two LDS difference arrays, prefix-sums them, and writes depth/uniq with
//            coalesced 16-byte stores.  It also zeroes the counts it consumed, so the scratch
//            is clean for the next call without any memset.
//
// Exactness: every step lies in exactly one run, so it contributes +1 to exactly one depth
// record; every (path, segment) pair that occurs sets exactly one bit, which lies in exactly
// one uniq record.  Sums of +1s are order-independent, hence the results equal
// ops/depth.rs:15-39 bit for bit under any scheduling.  Runs are cut at ids that are multiples
// of 2048, so a record never crosses a window.  Sub-buckets have a fixed capacity; a record
// that does not fit is applied to a global difference array with atomics instead (slow, still
// exact) and k_accum folds that array in.
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
constexpr int kAccThreads = 1024;

// diagnostic ablations (FLATGFA_DEBUG_SKIP, results are then wrong by construction)
constexpr uint32_t kDbgNoStore = 1, kDbgNoBitset = 4, kDbgNoTiles = 8;
// the ablation checks exist only in the DBG instantiation of the kernel
#define FGFA_SKIP(bit) (DBG && (A.dbg & (bit)))

struct ScanArgs {
    const uint32_t *steps;
    const uint4 *items;  // work queue, longest first: {begin, end, piece slot or kNoSlot, unused}
    uint32_t *piece_bits;  // [n_piece_slots][n_words]: "seen" bitsets of the pieces of split paths
    uint32_t n_items, n_segs, n_win, n_words, n_slots;
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
__device__ __noinline__ void overflow_apply(int *arr, uint32_t *flag, uint32_t id, uint32_t len) {
    flag[id >> kWinBits] = 1u;
    atomicAdd(&arr[id], 1);
    if ((id & (kWin - 1)) + len < kWin) atomicAdd(&arr[id + len], -1);
}
// kind: 0 = depth, 1 = uniq, 2 = both
__device__ __forceinline__ void overflow_record(const ScanArgs &A, uint32_t id, uint32_t len, uint32_t kind) {
    if (kind != 1u) overflow_apply(A.ovf_d, A.ovf_flag, id, len);  // by value: A stays in SGPRs
    if (kind != 0u) overflow_apply(A.ovf_u, A.ovf_flag, id, len);
}

// Store a record at slot `pos` of this workgroup's sub-bucket of window (id >> 12).  Branch
// free: lanes with nothing to store (or no room) write to the sink window.  Returns whether
// the record still has to take the overflow route.
template <bool DBG>
__device__ __forceinline__ bool put(const ScanArgs &A, uint32_t *mine, bool e, uint32_t pos, uint32_t id,
                                    uint32_t lenm1, uint32_t kind) {
    const bool ok = e && pos < A.cap;
    // the bucket array holds fewer than 2^30 records, so a 32-bit byte offset from a uniform base suffices
    const uint32_t boff = (ok ? (id >> kWinBits) * A.stride + pos : A.sink) << 2;
    if (!FGFA_SKIP(kDbgNoStore))
        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(mine) + boff) =
            (id & (kWin - 1)) | (lenm1 << kWinBits) | (kind << 24);
    return e && !ok;
}

__device__ __forceinline__ uint32_t clamp_id(const ScanArgs &A, uint32_t id) {
    if (id >= A.n_segs) {
        *A.status = 1u;
        return 0u;
    }
    return id;
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Per-wave state: the run queue (LDS, kQCap entries of (start id << 11) | (len - 1)), how many
// entries it holds, the id of the step before the next one, and the start id of the run that
// step belongs to; and the queue of partially new claims (see emit_chunk).  `fill`, `pfill`,
// `prev` and `rs` are wave-uniform.
struct Wave {
    uint32_t *q, *pq;
    uint32_t fill, pfill, prev, rs;
    int lane;
};

__device__ __forceinline__ void enqueue(Wave &w, bool e, uint32_t rec) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(e);
    if (e) w.q[w.fill + lane_rank(m)] = rec;
    w.fill += (uint32_t)__builtin_popcountll(m);
}

constexpr uint32_t kQCap = 320;  // 63 left over + up to 256 from one tile
constexpr uint32_t kPCap = 192;  // 63 left over + up to 2 x 64 from one chunk of claims

__device__ __forceinline__ void push_partial(Wave &w, bool e, uint32_t ent) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(e);
    if (e) w.pq[w.pfill + lane_rank(m)] = ent;
    w.pfill += (uint32_t)__builtin_popcountll(m);
}

// Emit up to 64 queued runs, one per lane.  Each run becomes one depth record.  For unique
// depth the lane claims the run's segments in the path's "seen" bitset with ONE returning LDS OR
// (runs are cut at multiples of 32, so a run lies inside one word): the bits that were still
// clear are exactly the (path, segment) pairs this run is the first to touch.  If all of them
// were clear the depth record doubles as the uniq record (kind 2), if none was there is nothing
// to add.  The rare claim that is partly new is parked, as (word, half, 16 new bits), on a second
// queue; that queue is turned into uniq records 64 entries at a time, so its bit-stretch loop
// runs with all lanes busy instead of once per chunk for a lane or two.
template <bool UNIQ, bool DBG>
__device__ __forceinline__ void emit_chunk(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine,
                                           bool valid, uint32_t rec) {
    const uint32_t id = rec >> kRunBits, lenm1 = rec & (kRunSpan - 1), win = id >> kWinBits;
    uint32_t kind = 0;
    if (UNIQ && !FGFA_SKIP(kDbgNoBitset)) {
        const uint32_t mask = valid ? (0xFFFFFFFFu >> (31u - lenm1)) << (id & 31u) : 0u;
        const uint32_t old = mask ? atomicOr(&seen[id >> 5], mask) : 0u;
        const uint32_t nb = mask & ~old;
        kind = (nb == mask) ? 2u : 0u;
        const uint32_t part = (nb == mask) ? 0u : nb;
        const uint32_t word = (id >> 5) << 17;
        push_partial(w, (part & 0xFFFFu) != 0u, word | (part & 0xFFFFu));
        push_partial(w, (part >> 16) != 0u, word | 0x10000u | (part >> 16));
    }
    const uint32_t pos = valid ? atomicAdd(&bcur[win], 1u) : 0u;
    const bool o0 = put<DBG>(A, mine, valid, pos, id, lenm1, kind);
    if (__builtin_amdgcn_ballot_w64(o0) && o0) overflow_record(A, id, lenm1 + 1, kind);  // rare: the sub-bucket is full
}

// Turn whole chunks of 64 parked claims (all of them when `all`) into uniq records: one per
// stretch of new bits.
template <bool DBG>
__device__ __forceinline__ void drain_partial(const ScanArgs &A, Wave &w, uint32_t *bcur, uint32_t *mine, bool all) {
    while (w.pfill >= 64u || (all && w.pfill)) {
        const uint32_t n = min(w.pfill, 64u);
        const bool valid = (uint32_t)w.lane < n;
        const uint32_t ent = valid ? w.pq[w.lane] : 0u;
        const uint32_t rest = w.pfill - n;
        for (uint32_t i = w.lane; i < rest; i += 64) {  // ds ops of one wave execute in order
            const uint32_t t = w.pq[n + i];
            w.pq[i] = t;
        }
        w.pfill = rest;
        const uint32_t base = ((ent >> 17) << 5) | ((ent >> 12) & 16u);  // first segment of the half word
        const uint32_t win = base >> kWinBits;
        uint32_t m = ent & 0xFFFFu;
        while (__builtin_amdgcn_ballot_w64(m != 0u)) {
            const bool e = m != 0u;
            const uint32_t tz = e ? (uint32_t)__builtin_ctz(m) : 0u;
            const uint32_t run = (uint32_t)__builtin_ctz(~(m >> tz));  // m has 16 bits: ~(m >> tz) is never 0
            m &= ~(((1u << run) - 1u) << tz);
            const uint32_t p = e ? atomicAdd(&bcur[win], 1u) : 0u;
            const bool o1 = put<DBG>(A, mine, e, p, base + tz, run - 1u, 1u);
            if (__builtin_amdgcn_ballot_w64(o1) && o1) overflow_record(A, base + tz, run, 1u);
        }
    }
}

// Emit whole chunks of 64 queued runs (all of them when `all`), keeping the rest at the front.
template <bool UNIQ, bool DBG>
__device__ __forceinline__ void drain(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine, bool all) {
    while (w.fill >= 64u || (all && w.fill)) {
        const uint32_t n = min(w.fill, 64u);
        const bool valid = (uint32_t)w.lane < n;
        const uint32_t rec = valid ? w.q[w.lane] : 0u;
        emit_chunk<UNIQ, DBG>(A, w, seen, bcur, mine, valid, rec);
        const uint32_t rest = w.fill - n;
        for (uint32_t i = w.lane; i < rest; i += 64) {  // ds ops of one wave execute in order
            const uint32_t t = w.q[n + i];
            w.q[i] = t;
        }
        w.fill = rest;
        if (UNIQ) drain_partial<DBG>(A, w, bcur, mine, false);
    }
    if (UNIQ && all) drain_partial<DBG>(A, w, bcur, mine, true);
}

// 256 steps, four consecutive ones per lane (lane i holds steps 4i..4i+3 of the tile).
template <bool UNIQ, bool DBG>
__device__ __forceinline__ void tile_full(const ScanArgs &A, Wave &w, uint4 v) {
    uint32_t a0 = v.x >> 1, a1 = v.y >> 1, a2 = v.z >> 1, a3 = v.w >> 1;
    if (max(max(a0, a1), max(a2, a3)) >= A.n_segs) {
        a0 = clamp_id(A, a0);
        a1 = clamp_id(A, a1);
        a2 = clamp_id(A, a2);
        a3 = clamp_id(A, a3);
    }
    uint32_t prev = __builtin_amdgcn_update_dpp(0u, a3, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    if (w.lane == 0) prev = w.prev;
    const bool s0 = (a0 != prev + 1) | ((a0 & kCutMask<UNIQ>) == 0);
    const bool s1 = (a1 != a0 + 1) | ((a1 & kCutMask<UNIQ>) == 0);
    const bool s2 = (a2 != a1 + 1) | ((a2 & kCutMask<UNIQ>) == 0);
    const bool s3 = (a3 != a2 + 1) | ((a3 & kCutMask<UNIQ>) == 0);
    // start id of the run in progress when this lane's first step arrives
    const bool any = s0 | s1 | s2 | s3;
    const uint32_t last_start = s3 ? a3 : (s2 ? a2 : (s1 ? a1 : a0));
    const unsigned long long below = __builtin_amdgcn_ballot_w64(any) & ((1ull << w.lane) - 1ull);
    const int src = below ? 63 - __builtin_clzll(below) : w.lane;
    const uint32_t from_below = __shfl(last_start, src, 64);
    const uint32_t rs = below ? from_below : w.rs;
    // a run ends wherever the next one starts: queue (start, length) of the run that just ended
    const uint32_t len0 = prev - rs + 1;
    const uint32_t rsA = s0 ? a0 : rs;
    const uint32_t rsB = s1 ? a1 : rsA;
    const uint32_t rsC = s2 ? a2 : rsB;
    const uint32_t rsD = s3 ? a3 : rsC;
    enqueue(w, s0 && len0 != 0, (rs << kRunBits) | (len0 - 1));
    enqueue(w, s1, (rsA << kRunBits) | (a0 - rsA));
    enqueue(w, s2, (rsB << kRunBits) | (a1 - rsB));
    enqueue(w, s3, (rsC << kRunBits) | (a2 - rsC));
    w.prev = __builtin_amdgcn_readlane(a3, 63);
    w.rs = __builtin_amdgcn_readlane(rsD, 63);
}

// Up to 64 consecutive steps, one per lane (heads, tails and short spans).
template <bool UNIQ>
__device__ __forceinline__ void tile_narrow(const ScanArgs &A, Wave &w, uint64_t t, uint32_t count) {
    const bool valid = (uint32_t)w.lane < count;
    const uint32_t id = valid ? clamp_id(A, A.steps[t + w.lane] >> 1) : 0u;
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

// One wave's share of one path: steps [lo, hi), of which [t0, t0 + 256 * nfull) are whole,
// 16-byte-aligned tiles read through `src`.
struct Span {
    uint64_t lo, hi, t0, nfull;
    const uint4 *src;
    uint32_t slot;  // where to leave the bitset when this is a piece of a split path
};

__device__ __forceinline__ Span make_span(const ScanArgs &A, uint32_t job, int wave, int lane) {
    Span s;
    s.lo = s.hi = s.t0 = s.nfull = 0;
    s.src = nullptr;
    s.slot = kNoSlot;
    if (job < A.n_items) {
        const uint4 it = A.items[job];
        const uint64_t b = it.x, e = it.y, n = e - b;
        s.slot = it.z;
        // contiguous span per wave, a whole number of tiles
        const uint64_t per = ((n + kWaves - 1) / kWaves + 255) / 256 * 256;
        s.lo = min(b + per * wave, e);
        s.hi = min(s.lo + per, e);
        s.t0 = min(s.lo + ((4 - (s.lo & 3)) & 3), s.hi);
        s.nfull = (s.hi - s.t0) / 256;
        s.src = reinterpret_cast<const uint4 *>(A.steps + s.t0) + lane;
    }
    return s;
}

// Streaming loads of steps.  Each handle is read exactly once (nontemporal: keep it out of the
// way of the bucket lines the L2 is write-combining).  Four tiles per wave are kept in flight
// across loop iterations.  hipcc cannot express that: it drains vmcnt to 0 at the top of the
// loop, and an inline-asm load into a compiler-allocated register is unsafe because the compiler
// may copy the register (to rotate it through the loop) while the load is still in flight.  So
// the landing registers are four fixed quads, v[112:127], which the compiler is told are
// clobbered and never otherwise allocates (the kernel needs < 100 VGPRs; 128 is the budget of a
// 1024-thread workgroup).  tools/check_pinned_vgprs.py checks the generated ISA for exactly that
// (run by `make check` and by the CPU test suite).
// A tile is taken out of its quad by v_movs issued after a counted s_waitcnt: on gfx950 vmcnt
// counts loads and stores in issue order, so "at most N outstanding" with N <= the number of
// tile loads issued after the one we need is always sufficient; record stores issued in between
// only make the wait stricter.
#define FGFA_LOAD_Q(Q, A, B, C, D, p) \
    asm volatile("global_load_dwordx4 " Q ", %0, off nt" ::"v"(p) : "memory", A, B, C, D)
#define FGFA_TAKE_Q(A, B, C, D, cur)                                                                  \
    asm volatile("v_mov_b32 %0, " A "\n\tv_mov_b32 %1, " B "\n\tv_mov_b32 %2, " C "\n\tv_mov_b32 %3, " D \
                 : "=v"(cur.x), "=v"(cur.y), "=v"(cur.z), "=v"(cur.w)::"memory")
template <int SLOT>
__device__ __forceinline__ void load_tile_async(const uint4 *p) {
    if (SLOT == 0) FGFA_LOAD_Q("v[112:115]", "v112", "v113", "v114", "v115", p);
    if (SLOT == 1) FGFA_LOAD_Q("v[116:119]", "v116", "v117", "v118", "v119", p);
    if (SLOT == 2) FGFA_LOAD_Q("v[120:123]", "v120", "v121", "v122", "v123", p);
    if (SLOT == 3) FGFA_LOAD_Q("v[124:127]", "v124", "v125", "v126", "v127", p);
}
// Waits until at most `younger` (capped at 3) of this wave's memory operations are outstanding.
__device__ __forceinline__ void wait_tiles(uint64_t younger) {
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int SLOT>
__device__ __forceinline__ uint4 take_tile() {
    uint4 cur;
    if (SLOT == 0) FGFA_TAKE_Q("v112", "v113", "v114", "v115", cur);
    if (SLOT == 1) FGFA_TAKE_Q("v116", "v117", "v118", "v119", cur);
    if (SLOT == 2) FGFA_TAKE_Q("v120", "v121", "v122", "v123", cur);
    if (SLOT == 3) FGFA_TAKE_Q("v124", "v125", "v126", "v127", cur);
    return cur;
}

template <bool UNIQ, bool DBG>
__global__ __launch_bounds__(kThreads) void k_scan(const ScanArgs A) {
    extern __shared__ uint32_t lds[];
    // layout: [bcur: kMaxWin][run queues: kWaves * kQCap][partial-claim queues: kWaves * kPCap][seen: n_words]
    uint32_t *bcur = lds;
    uint32_t *seen = lds + kMaxWin + kWaves * (kQCap + kPCap);
    __shared__ uint32_t next_job;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the span math on the scalar unit
    uint32_t *mine = A.buckets + (size_t)blockIdx.x * A.cap;  // this workgroup's sub-bucket of window 0
    Wave w;
    w.q = lds + kMaxWin + wave * kQCap;
    w.pq = lds + kMaxWin + kWaves * kQCap + wave * kPCap;
    w.fill = w.pfill = 0;
    w.lane = lane;
    for (uint32_t i = threadIdx.x; i < kMaxWin; i += kThreads) bcur[i] = 0u;
    if (UNIQ)
        for (uint32_t i = threadIdx.x; i < A.n_words; i += kThreads) seen[i] = 0u;
    if (threadIdx.x == 0) next_job = atomicAdd(A.work_counter, 1u);
    __syncthreads();
    uint32_t job = __builtin_amdgcn_readfirstlane(next_job);
    __syncthreads();

    // The first tiles of a path are requested while the previous path is being wrapped up.
    Span sp = make_span(A, job, wave, lane);
    uint32_t first_raw = 0;
#define FGFA_PRELOAD()                                              \
    do {                                                            \
        if (sp.lo < sp.hi) first_raw = A.steps[sp.lo];              \
        if (sp.nfull > 0) load_tile_async<0>(sp.src);               \
        if (sp.nfull > 1) load_tile_async<1>(sp.src + 64);          \
        if (sp.nfull > 2) load_tile_async<2>(sp.src + 128);         \
        if (sp.nfull > 3) load_tile_async<3>(sp.src + 192);         \
    } while (0)
    // one tile: wait for its data, re-issue its register for the tile four ahead, process it
#define FGFA_TILE(K)                                                                      \
    if (i + (K) < sp.nfull) {                                                             \
        wait_tiles(sp.nfull - 1 - (i + (K)));                                             \
        const uint4 cur = take_tile<K>();                                                 \
        if (i + (K) + 4 < sp.nfull) load_tile_async<K>(sp.src + (i + (K) + 4) * 64);      \
        if (!FGFA_SKIP(kDbgNoTiles)) {                                                    \
            tile_full<UNIQ, DBG>(A, w, cur);                                              \
            drain<UNIQ, DBG>(A, w, seen, bcur, mine, false);                              \
        } else if (cur.x == 0x7FFFFFFFu) {                                                \
            *A.status = 2u;                                                               \
        }                                                                                 \
    }
    FGFA_PRELOAD();

    while (job < A.n_items) {
        if (threadIdx.x == 0) next_job = atomicAdd(A.work_counter, 1u);  // consumed after the barrier below
        if (sp.lo < sp.hi) {
            const uint32_t first = clamp_id(A, first_raw >> 1);
            w.prev = first - 1;  // the first step then continues a (so far empty) run that starts at it
            w.rs = first;
            if (sp.t0 > sp.lo) tile_narrow<UNIQ>(A, w, sp.lo, (uint32_t)(sp.t0 - sp.lo));
            // four tiles (4 KiB per wave, 64 KiB per CU) stay in flight
#pragma unroll 1
            for (uint64_t i = 0; i < sp.nfull; i += 4) {
                FGFA_TILE(0)
                FGFA_TILE(1)
                FGFA_TILE(2)
                FGFA_TILE(3)
            }
            uint64_t t = sp.t0 + sp.nfull * 256;
            while (t < sp.hi) {
                const uint32_t cnt = (uint32_t)min((uint64_t)64, sp.hi - t);
                tile_narrow<UNIQ>(A, w, t, cnt);
                t += cnt;
                drain<UNIQ, DBG>(A, w, seen, bcur, mine, false);
            }
            // close the run still open at the end of the span
            enqueue(w, lane == 0, (w.rs << kRunBits) | (w.prev - w.rs));
            drain<UNIQ, DBG>(A, w, seen, bcur, mine, true);
        }
        __syncthreads();  // every wave is done with this path's bitset; next_job is visible
        const uint32_t done_slot = sp.slot;
        job = __builtin_amdgcn_readfirstlane(next_job);
        sp = make_span(A, job, wave, lane);
        FGFA_PRELOAD();
        if (UNIQ) {
            uint4 *sv = reinterpret_cast<uint4 *>(seen);
            if (done_slot != kNoSlot) {
                // a piece of a split path: other pieces may have claimed the same segments, so
                // the bitset is kept for k_merge to find the duplicates
                uint4 *dst = reinterpret_cast<uint4 *>(A.piece_bits + (size_t)done_slot * A.n_words);
                for (uint32_t i = threadIdx.x; i < A.n_words / 4; i += kThreads) dst[i] = sv[i];
            }
            for (uint32_t i = threadIdx.x; i < A.n_words / 4; i += kThreads) sv[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();  // the bitset is clean before the next path claims bits
        }
    }
#undef FGFA_PRELOAD
#undef FGFA_TILE
#undef FGFA_LOAD_Q
#undef FGFA_TAKE_Q
    // publish how many records this workgroup left in each window's sub-bucket
    __syncthreads();
    for (uint32_t wdw = threadIdx.x; wdw < A.n_win; wdw += kThreads)
        A.counts[(size_t)wdw * A.n_slots + blockIdx.x] = bcur[wdw];
}

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

template <bool UNIQ>
__device__ __forceinline__ void apply_record(int *dd, int *ud, uint32_t rec) {
    const uint32_t rel = rec & (kWin - 1), len = ((rec >> kWinBits) & (kWin - 1)) + 1;
    const uint32_t kind = (rec >> 24) & 3u;  // 0 depth, 1 uniq, 2 both
    if (!UNIQ || kind != 1u) {
        atomicAdd(&dd[rel], 1);
        atomicAdd(&dd[rel + len], -1);  // rel + len <= 4096; slot 4096 is a sink
    }
    if (UNIQ && kind != 0u) {
        atomicAdd(&ud[rel], 1);
        atomicAdd(&ud[rel + len], -1);
    }
}

// inclusive prefix sum of 4096 ints held 4 per thread by 1024 threads; returns this thread's
// four prefix values.
__device__ __forceinline__ int4 block_scan4(const int *arr, int *wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int4 v = *reinterpret_cast<const int4 *>(arr + 4 * tid);
    v.y += v.x;
    v.z += v.y;
    v.w += v.z;
    int incl = v.w;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int add = incl - v.w;
    for (int k = 0; k < wave; ++k) add += wave_tot[k];
    __syncthreads();
    v.x += add;
    v.y += add;
    v.z += add;
    v.w += add;
    return v;
}

__device__ __forceinline__ void store4(uint32_t *out, uint32_t i0, uint32_t nvalid, int4 v) {
    if (i0 + 3 < nvalid) {
        *reinterpret_cast<uint4 *>(out + i0) = make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
    } else {
        const int a[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t k = 0; k < 4; ++k)
            if (i0 + k < nvalid) out[i0 + k] = (uint32_t)a[k];
    }
}

template <bool UNIQ>
__global__ __launch_bounds__(kAccThreads) void k_accum(const AccArgs A) {
    __shared__ __attribute__((aligned(16))) int dd[kWin + 64];
    __shared__ __attribute__((aligned(16))) int ud[UNIQ ? kWin + 64 : 64];
    __shared__ int wave_tot[kAccThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t win = blockIdx.x, w0 = win * kWin;
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
        dd[i] = d0;
        if (UNIQ) ud[i] = u0;
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
                            atomicAdd(&ud[(word << 5) + tz], -1);
                            atomicAdd(&ud[(word << 5) + tz + run], 1);
                        }
                    }
                }
            }
        }
    }
    // Drain the window's sub-buckets: each wave takes four of them per round so that four
    // independent 16-byte loads per lane are in flight.
    constexpr uint32_t kAccWaves = kAccThreads / 64;
    const uint32_t *cnt_base = A.counts + (size_t)win * A.n_slots;
    for (uint32_t s0 = wave; s0 < A.n_slots; s0 += 4 * kAccWaves) {
        uint32_t cnt[4];
        const uint32_t *bk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t s = s0 + k * kAccWaves;
            cnt[k] = s < A.n_slots ? min(cnt_base[s], A.cap) : 0u;
            bk[k] = A.buckets + ((size_t)win * A.n_slots + (s < A.n_slots ? s : 0u)) * A.cap;
        }
        const uint32_t max4 = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3])) >> 2;
        for (uint32_t i = lane; i < max4; i += 64) {
            uint4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i < (cnt[k] >> 2)) r[k] = reinterpret_cast<const uint4 *>(bk[k])[i];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i < (cnt[k] >> 2)) {
                    apply_record<UNIQ>(dd, ud, r[k].x);
                    apply_record<UNIQ>(dd, ud, r[k].y);
                    apply_record<UNIQ>(dd, ud, r[k].z);
                    apply_record<UNIQ>(dd, ud, r[k].w);
                }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t rest = (cnt[k] & ~3u) + lane;
            if (rest < cnt[k]) apply_record<UNIQ>(dd, ud, bk[k][rest]);
        }
    }
    __syncthreads();
    // the scratch is clean for the next call
    for (uint32_t s = tid; s < A.n_slots; s += kAccThreads) A.counts[(size_t)win * A.n_slots + s] = 0u;
    if (win == 0 && tid == 0) *A.work_counter = 0u;
    const uint32_t i0 = 4 * tid;
    store4(A.depth_out + w0, i0, nvalid, block_scan4(dd, wave_tot));
    if (UNIQ) store4(A.uniq_out + w0, i0, nvalid, block_scan4(ud, wave_tot));
}

uint32_t scan_lds_bytes(uint32_t n_words) { return (kMaxWin + kWaves * (kQCap + kPCap) + n_words) * 4u; }

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
    const uint32_t n_win = (g.n_segs + kWin - 1) / kWin;
    if (n_win > kMaxWin) return true;
    if (g.n_segs > (1u << 21)) return true;  // a queued run is (start id << 11) | (len - 1)
    const uint32_t n_words = ((g.n_segs + 31) / 32 + 3) & ~3u;
    if (n_words > (1u << 15)) return true;  // a parked claim is (word << 17) | (half << 16) | 16 bits
    if (scan_lds_bytes(n_words) + 64 > kLdsLimit) return true;  // the "seen" bitset must fit one CU's LDS
    hipDeviceProp_t prop;
    int dev = 0;
    FAST_TRY(hipGetDevice(&dev));
    FAST_TRY(hipGetDeviceProperties(&prop, dev));
    fp->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    fp->n_slots = fp->n_cus;
    fp->n_win = n_win;
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
    cap = std::min<uint64_t>(std::max<uint64_t>(cap & ~3ull, 4), max_cap & ~3ull);
    if (cap < 4) return true;
    fp->cap = (uint32_t)cap;
    if (const char *d = getenv("FLATGFA_DEBUG_SKIP")) fp->dbg = (uint32_t)strtoul(d, nullptr, 10);
    // Work items: whole paths, except that a path longer than `piece` steps is cut into pieces so
    // that graphs with few long paths still fill the chip.  Pieces carry a slot for their bitset.
    uint64_t piece = std::max<uint64_t>(65536, (g.n_steps + 2ull * fp->n_cus - 1) / (2ull * fp->n_cus));
    if (const char *forced = getenv("FLATGFA_PIECE_STEPS")) piece = std::max<uint64_t>(256, strtoull(forced, nullptr, 10));
    piece = (piece + 255) & ~255ull;
    std::vector<uint4> items;
    std::vector<uint2> split;
    uint32_t n_piece_slots = 0;
    for (uint32_t p = 0; p < g.n_paths; ++p) {
        const uint64_t b = hb[p], e = he[p], n = e - b;
        if (n == 0) continue;
        if (n <= piece) {
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
    std::stable_sort(items.begin(), items.end(), [](const uint4 &a, const uint4 &b) { return a.y - a.x > b.y - b.x; });
    fp->n_items = (uint32_t)items.size();
    fp->n_split = (uint32_t)split.size();
    if (items.empty()) return true;
    FAST_TRY(hipMalloc(&fp->counts, slots * 4));
    FAST_TRY(hipMemset(fp->counts, 0, slots * 4));
    FAST_TRY(hipMalloc(&fp->buckets, (slots + fp->n_slots) * cap * 4));
    FAST_TRY(hipMalloc(&fp->ovf_d, ((size_t)g.n_segs + 1) * 4));
    FAST_TRY(hipMalloc(&fp->ovf_u, ((size_t)g.n_segs + 1) * 4));
    FAST_TRY(hipMemset(fp->ovf_d, 0, ((size_t)g.n_segs + 1) * 4));
    FAST_TRY(hipMemset(fp->ovf_u, 0, ((size_t)g.n_segs + 1) * 4));
    FAST_TRY(hipMalloc(&fp->ovf_flag, (size_t)n_win * 4));
    FAST_TRY(hipMemset(fp->ovf_flag, 0, (size_t)n_win * 4));
    FAST_TRY(hipMalloc(&fp->items, items.size() * sizeof(uint4)));
    FAST_TRY(hipMemcpy(fp->items, items.data(), items.size() * sizeof(uint4), hipMemcpyHostToDevice));
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
                    (void *)fp->items, (void *)fp->piece_bits, (void *)fp->split, (void *)fp->work_counter})
        if (p) (void)hipFree(p);
    *fp = FastPlan();
}

int fast_seg_depth(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                   uint32_t *status, hipStream_t stream) {
    const uint32_t stride = fp.n_slots * fp.cap;
    ScanArgs sa{g.steps, reinterpret_cast<const uint4 *>(fp.items), fp.piece_bits, fp.n_items, g.n_segs, fp.n_win,
                uniq_out ? fp.n_words : 0u, fp.n_slots, fp.work_counter, fp.counts, fp.buckets, fp.cap, stride,
                fp.n_win * stride, fp.ovf_d, fp.ovf_u, fp.ovf_flag, status, fp.dbg};
    AccArgs aa{g.n_segs, fp.n_win, fp.n_slots, fp.cap, fp.piece_bits, reinterpret_cast<const uint2 *>(fp.split),
               uniq_out ? fp.n_split : 0u, fp.n_words, fp.counts, fp.buckets, fp.ovf_d, fp.ovf_u, fp.ovf_flag,
               fp.work_counter, depth_out, uniq_out};
    const uint32_t grid = std::min<uint32_t>(fp.n_items, fp.n_slots);  // one persistent workgroup per CU
    if (uniq_out) {
        {
            ProfScope ps("k_scan<uniq>", stream);
            if (fp.dbg) hipLaunchKernelGGL((k_scan<true, true>), dim3(grid), dim3(kThreads), fp.lds_bytes_uniq, stream, sa);
            else hipLaunchKernelGGL((k_scan<true, false>), dim3(grid), dim3(kThreads), fp.lds_bytes_uniq, stream, sa);
        }
        {
            ProfScope ps("k_accum<uniq>", stream);
            hipLaunchKernelGGL(k_accum<true>, dim3(fp.n_win), dim3(kAccThreads), 0, stream, aa);
        }
    } else {
        {
            ProfScope ps("k_scan<depth>", stream);
            hipLaunchKernelGGL((k_scan<false, false>), dim3(grid), dim3(kThreads), fp.lds_bytes_depth, stream, sa);
        }
        {
            ProfScope ps("k_accum<depth>", stream);
            hipLaunchKernelGGL(k_accum<false>, dim3(fp.n_win), dim3(kAccThreads), 0, stream, aa);
        }
    }
    if (hipGetLastError() != hipSuccess) {
        set_error("fast_seg_depth: kernel launch failed");
        return FLATGFA_ERR_HIP;
    }
    return FLATGFA_OK;
}

}  // namespace fgfa_dev
