// Pass 2 (seg_depth_with_uniq / seg_depth / path_depth, ops/depth.rs:15-56,88-131): one workgroup per window
// applies the window's run records to LDS difference arrays -- the "seen" bitset of depth.rs:23-34 lives here,
// per (path, window) -- and writes the counts; for path depth it also turns every record into two differences
// of window-local prefix sums.  See depth_fast.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "depth_fast_kernels.hpp"

namespace fgfa_dev {
namespace {

// ------------------------------------------------------------------ pass 2 ---

// Pass 2 keeps two difference arrays over the window in LDS: D for depth and R for revisits (steps
// on a segment their path had already touched).  A record is +1 at its first segment and -1 just
// past its last one in D; the stretches of its segments that were already claimed get the same
// pair in R.  uniq = depth - revisits.  First visits are the common case and cost nothing extra.
// A record of k_scan_short says what it counts for (bit 24: depth, bit 25: uniq).
template <bool UNIQ, int WB>
__device__ __forceinline__ void apply_record(int *D, int *R, uint32_t rec) {
    constexpr uint32_t kW = 1u << WB;
    const uint32_t rel = rec & (kW - 1), end = rel + ((rec >> WB) & ((1u << (kTagShift - WB)) - 1u)) + 1;  // (whatever lies below a tag; k_scan_short's lengths have eleven bits) end <= window size; that cell is a sink
    if (UNIQ) {
        const int d = (int)((rec >> 24) & 1u), rv = d - (int)((rec >> 25) & 1u);
        if (d) {
            atomicAdd(&D[rel], 1);
            atomicAdd(&D[end], -1);
        }
        if (rv) {
            atomicAdd(&R[rel], rv);
            atomicAdd(&R[end], -rv);
        }
    } else {
        atomicAdd(&D[rel], 1);
        atomicAdd(&D[end], -1);
    }
}

// Claiming a record of k_scan (it counts for depth; what it counts for uniq is decided here).
// A record's segments are claimed in the bitset of its path's group word by word, with returning
// ORs; the bits that were already set are revisits, which go to R one stretch at a time.  Most
// records lie in one word and have at most one stretch of revisits, some span thirty words: a
// loop over "the lanes that still have something to do" would run for the longest record of every
// step with most lanes idle.  So a step claims only the first word of its records; what is left
// of a record, and every word that has revisits, is parked on two small per-wave LDS lists and
// worked off 64 at a time, all lanes busy.
constexpr uint32_t kPend = 96;      // entries per list: fewer than kPendRun parked + up to 64 from one step
constexpr uint32_t kPendRun = 32;   // a list is worked off while it holds at least this many
struct Pending {
    uint32_t *m;   // [kPend] rest of a record: first unclaimed segment | last segment << PB | bitset slot << 2 PB (PB = 13; in a tagged call log2 of the window, which leaves eight bits for the slot at 4096 segments)
    uint2 *r;      // [kPend] {window-relative first segment of a bitset word, its revisited bits}
    uint32_t mcnt, rcnt;
    uint32_t moldest;  // ordinal (mod 256) of the path of the oldest entry on m
    int lane;
};

// One word of the lanes' records [p, e]: returns the word's revisited bits, and advances p.
__device__ __forceinline__ uint32_t claim_word(uint32_t *bits, bool act, uint32_t &p, uint32_t e, uint32_t &base) {
    uint32_t rv = 0;
    base = p & ~31u;
    if (act) {
        const uint32_t lo = p & 31u, wl = min(e, p | 31u), width = wl - p + 1u;
        const uint32_t mask = (0xFFFFFFFFu >> (32u - width)) << lo;
        rv = mask & atomicOr(&bits[p >> 5], mask);
        p = wl + 1u;
    }
    return rv;
}
__device__ __forceinline__ void park_rest(Pending &q, bool e, uint32_t val) {
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(e);
    if (e) q.m[q.mcnt + lane_rank(mk)] = val;
    q.mcnt = __builtin_amdgcn_readfirstlane(q.mcnt + (uint32_t)__builtin_popcountll(mk));  // (tells the compiler it is wave-uniform)
}
__device__ __forceinline__ void park_revisit(Pending &q, bool e, uint32_t base, uint32_t rv) {
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(e);
    if (e) q.r[q.rcnt + lane_rank(mk)] = make_uint2(base, rv);
    q.rcnt = __builtin_amdgcn_readfirstlane(q.rcnt + (uint32_t)__builtin_popcountll(mk));
}
// the newest (up to) 64 parked records: one more word each
template <int WB, int PB = 13>
__device__ __forceinline__ void run_rest(Pending &q, uint32_t *mybits, uint32_t dbg) {
    constexpr uint32_t kNW = (1u << WB) / 32u, kPM = (1u << PB) - 1u;
    const uint32_t n = __builtin_amdgcn_readfirstlane(min(q.mcnt, 64u));
    q.mcnt = __builtin_amdgcn_readfirstlane(q.mcnt - n);
    const bool act = (uint32_t)q.lane < n;
    const uint32_t v = act ? q.m[q.mcnt + q.lane] : 0u;
    asm volatile("" ::: "memory");  // the slots read here are written again below, by other lanes
    uint32_t p = v & kPM, base;
    const uint32_t e = (v >> PB) & kPM;
    uint32_t rv = claim_word(mybits + (v >> (2 * PB)) * kNW, act, p, e, base);
    if (dbg & kDbgNoRevisit) rv = 0;
    park_rest(q, act && p <= e, (v & ~kPM) | p);
    park_revisit(q, rv != 0u, base, rv);
}
// the newest (up to) 64 parked words: one stretch of revisited segments each
__device__ __forceinline__ void run_revisits(Pending &q, int *R) {
    const uint32_t n = __builtin_amdgcn_readfirstlane(min(q.rcnt, 64u));
    q.rcnt = __builtin_amdgcn_readfirstlane(q.rcnt - n);
    const bool act = (uint32_t)q.lane < n;
    const uint2 v = act ? q.r[q.rcnt + q.lane] : make_uint2(0u, 0u);
    asm volatile("" ::: "memory");
    uint32_t rv = v.y;
    if (rv) {
        const uint32_t low = rv & (0u - rv), sum = rv + low;  // adding the lowest set bit carries through its stretch
        const uint32_t from = (uint32_t)__builtin_ctz(low), to = sum ? (uint32_t)__builtin_ctz(sum) : 32u;
        rv &= sum;
        atomicAdd(&R[v.x + from], 1);
        atomicAdd(&R[v.x + to], -1);
    }
    park_revisit(q, rv != 0u, v.x, rv);
}
template <int WB, int PB = 13>
__device__ __forceinline__ void run_pending(Pending &q, int *R, uint32_t *mybits, uint32_t dbg, uint32_t at_least) {
    while (q.mcnt >= at_least || q.rcnt >= at_least) {
        if (q.rcnt >= at_least) run_revisits(q, R);
        else run_rest<WB, PB>(q, mybits, dbg);
        if (at_least == 1u && q.mcnt == 0u && q.rcnt == 0u) break;
    }
}

// One step: 64 records (rec == 0: none for this lane), `slot` = the bitset slot of the lane's path.
// In two halves, so that the caller can put independent work (forming the next step) between
// the request of the first word's claim and the use of its answer.
struct Claim {
    uint32_t old, mask, p, e, base, slot;
    bool act;
};
template <int WB, bool POINT = false>
__device__ __forceinline__ Claim claim_begin(int *D, Pending &q, uint32_t *mybits, uint32_t slot, uint32_t hfirst, uint32_t rec, bool valid, uint32_t dbg) {
    constexpr uint32_t kW = 1u << WB, kNW = kW / 32u;
    Claim c;
    const uint32_t rel = rec & (kW - 1);
    if (POINT) {  // one segment: one bit
        c.e = rel;
        if (valid) {
            atomicAdd(&D[rel], 1);
            atomicAdd(&D[rel + 1u], -1);
        }
        c.act = valid;
        c.slot = slot;
        c.base = rel & ~31u;
        c.mask = 1u << (rel & 31u);
        c.p = rel + 1u;
        c.old = valid ? atomicOr(&mybits[slot * kNW + (rel >> 5)], c.mask) : 0u;
        return c;
    }
    c.e = rel + ((rec >> WB) & 1023u);  // last segment of the run
    if (valid && !(dbg & kDbgNoDepth)) {
        atomicAdd(&D[rel], 1);
        atomicAdd(&D[c.e + 1u], -1);
    }
    c.act = valid && !(dbg & kDbgNoClaim);
    c.slot = slot;
    c.base = rel & ~31u;
    const uint32_t lo = rel & 31u, wl = min(c.e, rel | 31u), width = wl - rel + 1u;
    c.mask = (0xFFFFFFFFu >> (32u - width)) << lo;
    c.p = wl + 1u;
    c.old = c.act ? atomicOr(&mybits[slot * kNW + (rel >> 5)], c.mask) : 0u;
    if (q.mcnt == 0u) q.moldest = __builtin_amdgcn_readfirstlane(hfirst);
    return c;
}
template <int WB, bool POINT = false, int PB = 13>
__device__ __forceinline__ void claim_end(int *R, Pending &q, uint32_t *mybits, const Claim &c, uint32_t dbg) {
    uint32_t rv = c.act ? (c.mask & c.old) : 0u;
    if (dbg & kDbgNoRevisit) rv = 0;
    if (POINT) {  // every record is one segment (k_scan_dense): nothing to park, the revisit is the record
        if (rv) {
            atomicAdd(&R[c.e], 1);
            atomicAdd(&R[c.e + 1u], -1);
        }
        return;
    }
    park_rest(q, c.act && c.p <= c.e, c.p | (c.e << PB) | (c.slot << (2 * PB)));
    park_revisit(q, rv != 0u, c.base, rv);
    run_pending<WB, PB>(q, R, mybits, dbg, kPendRun);
}

// inclusive prefix sum of N*1024 values held N per thread by 1024 threads (v[] holds this thread's
// values on entry, their prefix sums on return).
template <typename T, int N>
__device__ __forceinline__ void block_scan(T *wave_tot, T (&v)[N]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 1; k < N; ++k) v[k] += v[k - 1];
    T incl = v[N - 1];
    for (int off = 1; off < 64; off <<= 1) {
        const T t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    T add = incl - v[N - 1];
    for (int k = 0; k < wave; ++k) add += wave_tot[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += add;
}

template <int N>
__device__ __forceinline__ void store_n(uint32_t *out, uint32_t i0, uint32_t nvalid, const uint32_t (&a)[N]) {
    if (N == 2) {
        if (i0 + 1 < nvalid) *reinterpret_cast<uint2 *>(out + i0) = make_uint2(a[0], a[1]);
        else if (i0 < nvalid) out[i0] = a[0];
        return;
    }
#pragma unroll
    for (int k0 = 0; k0 + 3 < N; k0 += 4) {
        if (i0 + k0 + 3 < nvalid) {
            *reinterpret_cast<uint4 *>(out + i0 + k0) = make_uint4(a[k0], a[k0 + 1], a[k0 + 2], a[k0 + 3]);
        } else {
            for (uint32_t k = 0; k < 4; ++k)
                if (i0 + k0 + k < nvalid) out[i0 + k0 + k] = a[k0 + k];
        }
    }
}

// The same for a window shared by several workgroups: each adds what it counted.
template <int N>
__device__ __forceinline__ void add_n(uint32_t *out, uint32_t i0, uint32_t nvalid, const uint32_t (&a)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k)
        if (a[k] && i0 + k < nvalid) atomicAdd(out + i0 + k, a[k]);
}

// Apply the records [0, scnt[slot]) of every sub-bucket of the window as they are (they say what
// they count for).  Each wave takes sixteen sub-buckets per round and requests the first 64 x 16
// bytes of every one before it applies any, so a round pays the memory latency once.
template <bool UNIQ, int WB, int kPerRound = 16>
__device__ __forceinline__ void apply_flat(const AccArgs &A, int *D, int *R, const uint32_t *scnt, const uint32_t *sstart, const uint32_t *wbase) {
    const int lane = threadIdx.x & 63;
    // wave-uniform: sub-bucket addressing stays scalar.  The waves of all of the window's workgroups share the sub-buckets out.
    const uint32_t uw = __builtin_amdgcn_readfirstlane(blockIdx.y * kAccWaves + (threadIdx.x >> 6)), nw = A.parts * kAccWaves;
    for (uint32_t s0 = uw; s0 < A.n_slots; s0 += kPerRound * nw) {
        uint4 r[kPerRound];
        uint32_t cnt[kPerRound];
#pragma unroll
        for (int k = 0; k < kPerRound; ++k) {
            const uint32_t s = s0 + k * nw;
            const uint32_t sc = s < A.n_slots ? s : 0u;
            cnt[k] = s < A.n_slots ? scnt[sc] : 0u;
            // unconditional (slot 0 always exists): a predicated load would be waited for on the spot
            r[k] = reinterpret_cast<const uint4 *>(wbase + sstart[sc])[(uint32_t)lane < (cnt[k] >> 2) ? lane : 0];
        }
#pragma unroll
        for (int k = 0; k < kPerRound; ++k) {
            if ((uint32_t)lane < (cnt[k] >> 2)) {
                apply_record<UNIQ, WB>(D, R, r[k].x);
                apply_record<UNIQ, WB>(D, R, r[k].y);
                apply_record<UNIQ, WB>(D, R, r[k].z);
                apply_record<UNIQ, WB>(D, R, r[k].w);
            }
        }
        // what does not fit the first pass (skewed sub-buckets), and the last 1..3 records
#pragma unroll 1
        for (int k = 0; k < kPerRound; ++k) {
            const uint32_t s = s0 + k * nw;
            if (s >= A.n_slots) break;
            const uint32_t c = scnt[s];
            const uint32_t *bk = wbase + sstart[s];
            for (uint32_t i = 64 + lane; i < (c >> 2); i += 64) {
                const uint4 v = reinterpret_cast<const uint4 *>(bk)[i];
                apply_record<UNIQ, WB>(D, R, v.x);
                apply_record<UNIQ, WB>(D, R, v.y);
                apply_record<UNIQ, WB>(D, R, v.z);
                apply_record<UNIQ, WB>(D, R, v.w);
            }
            const uint32_t rest = (c & ~3u) + lane;
            if (rest < c) apply_record<UNIQ, WB>(D, R, bk[rest]);
        }
    }
}

// Pass 2 requests the records of three steps ahead of their use.  As in k_scan, hipcc cannot keep
// a load in flight across loop iterations (it copies the destination register, which waits for
// the load), so the three landing registers are fixed -- v120, v121, v122, told to the compiler as
// clobbered and checked by tools/check_pinned_vgprs.py -- and a record is taken out after a
// counted wait: the two younger requests are the only other vector-memory operations in flight.
template <int K>
__device__ __forceinline__ void rec_request(const uint32_t *p) {
#ifndef FGFA_REC_POLICY
#define FGFA_REC_POLICY ""
#endif
    if (K == 0) asm volatile("global_load_dword v120, %0, off" FGFA_REC_POLICY ::"v"(p) : "memory", "v120");
    else if (K == 1) asm volatile("global_load_dword v121, %0, off" FGFA_REC_POLICY ::"v"(p) : "memory", "v121");
    else asm volatile("global_load_dword v122, %0, off" FGFA_REC_POLICY ::"v"(p) : "memory", "v122");
}
template <int K>
__device__ __forceinline__ uint32_t rec_take() {
    uint32_t r;
    if (K == 0) asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, v120" : "=v"(r)::"memory");
    else if (K == 1) asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, v121" : "=v"(r)::"memory");
    else asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, v122" : "=v"(r)::"memory");
    return r;
}

// inclusive prefix maximum across the wave
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0u, x, 0x111 /* row_shr:1 */, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0u, x, 0x112 /* row_shr:2 */, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0u, x, 0x114 /* row_shr:4 */, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0u, x, 0x118 /* row_shr:8 */, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0u, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0u, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, true));
    return x;
}

// Walk this wave's stretch of k_scan's items (plus its share of the handed-back ones).  Their
// directory entries are fetched 64 at a time, one per lane; the records of consecutive items are
// then walked as one stream, 64 records per step whatever the items' sizes, every lane knowing
// which item (hence which path's bitset) its record belongs to.  A wave keeps kSlots bitsets: a
// step never spans more paths than that.  Three steps' records are requested ahead of their use.
// SHARED: the stretch [e0, e1) is this wave's share of ONE path's items, `mybits` is the same
// for all waves of the workgroup and has been cleared by it: no hand-backs, no clearing here.
// BIG: before a step's records are mapped to their items, look whether the step lies inside the item
// of the step before (a build of its own: the few instructions cost 3-5 % where paths have a dozen
// records per window, and save 10-35 % where they have hundreds; the plan's creator times both).
template <int WB, bool SHARED, bool POINT, bool BIG>
__device__ __forceinline__ void apply_groups(const AccArgs &A, int *D, int *R, uint32_t *mybits, uint32_t *mark, uint32_t *pend, const uint32_t *wbase, uint32_t win,
                                             uint32_t e0, uint32_t e1, bool have_first = false, uint2 be_first = make_uint2(0u, 0u), uint32_t slf_first = 0u) {
    constexpr uint32_t kNW = (1u << WB) / 32u;             // words per bitset
    constexpr uint32_t kSlots = WB <= 12 ? 8u : 4u;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.y * kAccWaves + (threadIdx.x >> 6)), nw = A.parts * kAccWaves;
    const uint32_t nback = (A.has_pre && !SHARED) ? min(__builtin_amdgcn_readfirstlane(*A.work_counter), A.max_back) : 0u;
    const uint32_t nst = e1 - e0;
    const uint32_t nE = nst + (nback > wave ? (nback - wave + nw - 1u) / nw : 0u);
    uint32_t gbase = 0, carryG = 0;  // path ordinals are 1-based: 0 = none yet
    uint32_t hbase = 0, hdone = 0;   // ordinals (mod 256 where compared) of the paths that have records in this window
    Pending pq;
    pq.m = pend;
    pq.r = reinterpret_cast<uint2 *>(pend + kPend);
    pq.mcnt = pq.rcnt = pq.moldest = 0;
    pq.lane = lane;
    // An item's place in the walk order is where k_scan left its cursors and its sub-bucket: two
    // coalesced reads per 64 items, requested one round ahead (they are older than every record
    // request of the round, so the counted waits on those still hold).
    uint2 be_next = make_uint2(0u, 0u);
    uint32_t slf_next = 0u;
    const auto fetch = [&](uint32_t mb) {
        const uint32_t x = mb + (uint32_t)lane;
        const uint32_t at = x < nst ? e0 + x : A.n_items + wave + nw * (x - nst);
        be_next = x < nE ? A.dir[(size_t)win * A.dstride + at] : make_uint2(0u, 0u);
        slf_next = x < nE ? A.islot[at] : 0u;
    };
    if (have_first) {  // (requested by the kernel before it set itself up)
        be_next = be_first;
        slf_next = slf_first;
    } else {
        fetch(0);
    }
    for (uint32_t mb = 0; mb < nE; mb += 64u) {
        const uint2 be = be_next;
        const uint32_t slf = slf_next;
        fetch(mb + 64u);
        const uint32_t sl = slf & 0x7FFFFFFFu, first = slf >> 31;
        const uint32_t b = min(be.x, A.cap), en = max(b, min(be.y, A.cap));
        const uint32_t n = en - b;
        const uint32_t incl = wave_scan_incl(n), P = incl - n;
        const uint32_t T = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t G = gbase + wave_scan_incl(first);  // which path the item belongs to
        gbase = __builtin_amdgcn_readlane(G, 63);
        // H: the same, counting only paths that have records here, so that the paths a step
        // spans have consecutive ordinals -- their bitset is slot H mod kSlots.
        const unsigned long long ne = __builtin_amdgcn_ballot_w64(n != 0u);
        const unsigned long long below = ne & ((1ull << lane) - 1ull);
        const int pv = below ? 63 - __builtin_clzll(below) : 0;
        const uint32_t gsh = __shfl(G, pv, 64);  // outside the select: every lane must take part
        const uint32_t gprev = below ? gsh : carryG;
        const uint32_t H = hbase + wave_scan_incl((n != 0u && gprev != G) ? 1u : 0u);
        hbase = __builtin_amdgcn_readlane(H, 63);
        if (ne) carryG = __builtin_amdgcn_readlane(G, 63 - __builtin_clzll(ne));
        // the item's first record (an element offset from the window's bucket base, < 2^24) and H mod 256
        const uint32_t offH = (sl * A.cap + b) | (H << 24);
        uint32_t cs = 0, lastE = 0;  // position in the stream; the item the stream's last prepared record lies in
        uint32_t curEnd = 0;         // where that item's records end in the stream
        uint32_t hseen = hdone;      // ordinal of the last record's path in the steps prepared so far
        struct Chunk {
            const uint32_t *src;  // per lane: where its record is (the bucket base for lanes without one)
            uint32_t slot;        // per lane: its path's bitset slot, or kNoSlot for lanes without a record
            uint32_t nv, hl, hf;  // uniform: records in this step, ordinals of the last and the first one's path
        };
        // The next step of the stream: up to 64 records from position cs on.  Every item that
        // starts inside [cs, cs + 64) leaves its index at its start position; a running maximum
        // then tells every position which item it lies in (a scalar loop over the items is as fast
        // when a step holds one or two items, and several times slower when it holds many).  A
        // step stops short of the record that would bring a kSlots-th further path into it.  When
        // the stream is exhausted the step is empty (nv = 0) but is still formed, so that every
        // step requests one load.
        auto prep = [&]() -> Chunk {
            Chunk c;
            const uint32_t q = cs + (uint32_t)lane;
            if (BIG && cs + 64u <= curEnd) {  // (uniform) the whole step lies inside the item of the step before: nothing to look up
                const uint32_t oh = __builtin_amdgcn_readlane(offH, lastE), Ps = __builtin_amdgcn_readlane(P, lastE);
                const uint32_t h = oh >> 24;
                c.hf = c.hl = h;
                c.nv = 64u;
                c.src = wbase + (oh & 0xFFFFFFu) + (q - Ps);
                c.slot = h & (kSlots - 1u);
                hseen = h;
                cs += 64u;
                return c;
            }
            const uint32_t relp = P - cs;
            mark[lane] = 0u;
            if (n != 0u && relp < 64u) mark[relp] = (uint32_t)lane + 1u;
            // Lanes talk to each other through `mark`: without this the compiler forwards the zero a
            // lane has just stored to its own load (it reasons about one thread at a time).
            asm volatile("" ::: "memory");
            const uint32_t sel = max(wave_scan_max(mark[lane]), lastE + 1u) - 1u;
            const uint32_t oh = __shfl(offH, (int)sel, 64), Ps = __shfl(P, (int)sel, 64);
            const uint32_t h = oh >> 24;
            c.hf = __builtin_amdgcn_readfirstlane(h);
            const bool valid = q < T && ((h - c.hf) & 0xFFu) < kSlots;
            c.nv = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(valid));  // a prefix of the lanes
            c.src = wbase + (valid ? (oh & 0xFFFFFFu) + (q - Ps) : 0u);
            c.slot = valid ? (h & (kSlots - 1u)) : kNoSlot;
            const uint32_t last = c.nv ? c.nv - 1u : 0u;
            c.hl = c.nv ? __builtin_amdgcn_readlane(h, last) : hseen;
            lastE = c.nv ? __builtin_amdgcn_readlane(sel, last) : lastE;
            if (BIG) curEnd = __builtin_amdgcn_readlane(incl, lastE);
            hseen = c.hl;
            cs += c.nv;
            return c;
        };
        auto begin = [&](const Chunk &c, uint32_t loaded) -> Claim {
            // The paths met for the first time in this step get clean bitsets.  A slot is reused
            // every kSlots paths: whatever is still parked for its previous owner goes first.
            const uint32_t fresh = (c.hl - hdone) & 0xFFu;
            if (fresh && pq.mcnt && ((c.hl - pq.moldest) & 0xFFu) >= kSlots) {
                while (pq.mcnt) {
                    if (pq.rcnt >= kPendRun) run_revisits(pq, R);
                    else run_rest<WB>(pq, mybits, A.dbg);
                }
                run_pending<WB>(pq, R, mybits, A.dbg, kPendRun);
            }
            for (uint32_t k = 1; !SHARED && k <= fresh; ++k) {
                uint32_t *bs = mybits + ((hdone + k) & (kSlots - 1u)) * kNW;
                for (uint32_t i = lane; i < kNW / 2u; i += 64) reinterpret_cast<uint2 *>(bs)[i] = make_uint2(0u, 0u);
            }
            hdone = c.hl;
            const bool has = c.slot != kNoSlot;
            return claim_begin<WB, POINT>(D, pq, mybits, has ? c.slot : 0u, c.hf, loaded, has && loaded != 0u, A.dbg);
        };
        Chunk c0 = prep();
        rec_request<0>(c0.src);
        Chunk c1 = prep();
        rec_request<1>(c1.src);
        Chunk c2 = prep();
        rec_request<2>(c2.src);
        // One step: take its records, request the claim of their first words, form the step three
        // ahead and request its records while that claim is under way, then use the claim's answer.
#define FGFA_ACC_STEP(K, CK)                                  \
    if (CK.nv == 0u) break;                                   \
    {                                                         \
        const Claim cl = begin(CK, rec_take<K>());            \
        CK = prep();                                          \
        rec_request<K>(CK.src);                               \
        claim_end<WB, POINT>(R, pq, mybits, cl, A.dbg);       \
    }
        while (true) {
            FGFA_ACC_STEP(0, c0)
            FGFA_ACC_STEP(1, c1)
            FGFA_ACC_STEP(2, c2)
        }
#undef FGFA_ACC_STEP
    }
    run_pending<WB>(pq, R, mybits, A.dbg, 1u);
}

// minimum / maximum across the wave, uniform
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(~0u, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(~0u, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(~0u, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(~0u, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(~0u, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(~0u, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) { return __builtin_amdgcn_readlane(wave_scan_max(x), 63); }

// The tagged walk's record requests: as rec_request / rec_take, but the address is a wave-uniform
// pointer (an SGPR pair) plus the lane's own four bytes -- nothing to compute per lane and step.
// (the build with two workgroups per CU has 64 registers: its landing registers are v61 .. v63)
template <int K>
__device__ __forceinline__ void rec_request_lo(const uint32_t *p, uint32_t lane4) {
    if (K == 0) asm volatile("global_load_dword v61, %0, %1" ::"v"(lane4), "s"(p) : "memory", "v61");
    else if (K == 1) asm volatile("global_load_dword v62, %0, %1" ::"v"(lane4), "s"(p) : "memory", "v62");
    else asm volatile("global_load_dword v63, %0, %1" ::"v"(lane4), "s"(p) : "memory", "v63");
}
template <int K>
__device__ __forceinline__ uint32_t rec_take_lo() {
    uint32_t r;
    if (K == 0) asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, v61" : "=v"(r)::"memory");
    else if (K == 1) asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, v62" : "=v"(r)::"memory");
    else asm volatile("s_waitcnt vmcnt(2)\n\tv_mov_b32 %0, v63" : "=v"(r)::"memory");
    return r;
}
#ifndef FGFA_TAG_DEPTH
#define FGFA_TAG_DEPTH 3  /* steps the tagged walk requests ahead (3 .. 8): landing registers v(123 - depth) .. v122 */
#endif
constexpr int kTagDepth = FGFA_TAG_DEPTH;
static_assert(kTagDepth >= 3 && kTagDepth <= 8, "FGFA_TAG_DEPTH");
template <int K>
__device__ __forceinline__ void rec_request_s(const uint32_t *p, uint32_t lane4) {
#define FGFA_REQ_S(REG) asm volatile("global_load_dword " REG ", %0, %1" ::"v"(lane4), "s"(p) : "memory", REG)
    constexpr int kReg = 123 - kTagDepth + K;
    if (kReg == 115) FGFA_REQ_S("v115");
    else if (kReg == 116) FGFA_REQ_S("v116");
    else if (kReg == 117) FGFA_REQ_S("v117");
    else if (kReg == 118) FGFA_REQ_S("v118");
    else if (kReg == 119) FGFA_REQ_S("v119");
    else if (kReg == 120) FGFA_REQ_S("v120");
    else if (kReg == 121) FGFA_REQ_S("v121");
    else FGFA_REQ_S("v122");
#undef FGFA_REQ_S
}
template <int K>
__device__ __forceinline__ uint32_t rec_take_s() {  // (the kTagDepth - 1 younger requests are the only other vector-memory operations in flight)
    uint32_t r;
#define FGFA_TAKE_S(REG) asm volatile("s_waitcnt vmcnt(%1)\n\tv_mov_b32 %0, " REG : "=v"(r) : "n"(kTagDepth - 1) : "memory")
    constexpr int kReg = 123 - kTagDepth + K;
    if (kReg == 115) FGFA_TAKE_S("v115");
    else if (kReg == 116) FGFA_TAKE_S("v116");
    else if (kReg == 117) FGFA_TAKE_S("v117");
    else if (kReg == 118) FGFA_TAKE_S("v118");
    else if (kReg == 119) FGFA_TAKE_S("v119");
    else if (kReg == 120) FGFA_TAKE_S("v120");
    else if (kReg == 121) FGFA_TAKE_S("v121");
    else FGFA_TAKE_S("v122");
#undef FGFA_TAKE_S
    return r;
}

// What one step does with its 64 records once each lane knows its path's bitset (sb: the bitset's
// LDS byte address; vm: the lanes that hold a record): +1/-1 into D, the run's bits ORed into
// the bitset one word at a time -- the first word by all lanes, further words by the lanes that
// have any (a run of ten segments crosses a word boundary one time in three) -- and every stretch
// of bits that were already set a +1/-1 pair into R (adding the lowest set bit carries through
// its stretch; a word rarely has two).  Hand-written: as hipcc renders the same C++ a step costs
// 100 vector and 120 scalar instructions, and a CU issues one of each per cycle for its sixteen
// waves -- the walk was bound by instruction issue, scalar before vector (FLATGFA_ACC_SKIP
// ablations).  This is 19 + 2 vector instructions for the first word, 12 per turn of the
// revisit loop, 8 per further word, and a dozen scalar ones.  Nothing is parked, so a bitset can
// change hands at any step.
template <int WB, bool NC>
__device__ __forceinline__ void claim_step(uint32_t rec, uint32_t sb, unsigned long long vm, unsigned long long cm, uint32_t dbase, uint32_t rbase,
                                           uint32_t one, uint32_t mone) {  // (cm, a subset of vm: the lanes whose record needs a claim -- the others' items never meet a segment twice, see kTagNoClaim)
    constexpr uint32_t kRelMask = (1u << WB) - 1u, kBaseMask = kRelMask & ~31u;
    uint32_t n, a, w, tt, m, k, mask, base, old, rv, low, sum, f, g;
    unsigned long long sv, s2;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[vm]\n\t"
        "v_and_b32 %[a], %[relmask], %[rec]\n\t"              // the run's first segment, window-relative
        "v_bfe_u32 %[n], %[rec], %[wb], 10\n\t"               // its length - 1
        "v_lshl_add_u32 %[a], %[a], 2, %[dbase]\n\t"
        "ds_add_u32 %[a], %[one]\n\t"                         // D[first] += 1
        "v_lshl_add_u32 %[a], %[n], 2, %[a]\n\t"
        "ds_add_u32 %[a], %[mone] offset:4\n\t"               // D[last + 1] -= 1
#if FGFA_TAG_ABLATE & 4
        "s_branch 4f\n\t"
#endif
        ".if %[nc]\n\t"
        "s_mov_b64 exec, %[cm]\n\t"                            // the lanes that claim (NC builds only)
        ".endif\n\t"
        "v_bfe_u32 %[w], %[rec], 5, %[wb5]\n\t"               // the first segment's word in the bitset
        "v_lshl_add_u32 %[w], %[w], 2, %[sb]\n\t"
        "v_and_b32 %[k], 31, %[rec]\n\t"                      // its bit in that word
        "v_add_u32 %[tt], %[k], %[n]\n\t"                     // the last segment's bit, counted from bit 0 of the first word
        "v_min_u32 %[m], 31, %[tt]\n\t"
        "v_sub_u32 %[m], 31, %[m]\n\t"
        "v_lshrrev_b32 %[m], %[m], -1\n\t"
        "v_lshlrev_b32 %[k], %[k], -1\n\t"
        "v_and_b32 %[mask], %[m], %[k]\n\t"
        "v_and_b32 %[base], %[basemask], %[rec]\n\t"
        "v_lshl_add_u32 %[base], %[base], 2, %[rbase]\n\t"    // R's cell of the word's first segment
        "1:\n\t"
        "ds_or_rtn_b32 %[old], %[w], %[mask]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
#if FGFA_TAG_ABLATE & 8
        "s_branch 4f\n\t"
#endif
        "v_and_b32 %[rv], %[old], %[mask]\n\t"                // the segments this path had already visited
        "v_cmp_ne_u32 vcc, 0, %[rv]\n\t"
        "s_cbranch_vccz 3f\n\t"
        "s_mov_b64 %[s2], exec\n\t"
        "2:\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "v_sub_u32 %[low], 0, %[rv]\n\t"
        "v_and_b32 %[low], %[rv], %[low]\n\t"                 // the lowest revisited segment
        "v_add_u32 %[sum], %[rv], %[low]\n\t"                 // (the carry runs through its stretch)
        "v_ffbl_b32 %[f], %[low]\n\t"
        "v_ffbl_b32 %[g], %[sum]\n\t"
        "v_min_u32 %[g], 32, %[g]\n\t"                        // (no bit left: the stretch ends with the word)
        "v_lshl_add_u32 %[f], %[f], 2, %[base]\n\t"
        "v_lshl_add_u32 %[g], %[g], 2, %[base]\n\t"
        "ds_add_u32 %[f], %[one]\n\t"
        "ds_add_u32 %[g], %[mone]\n\t"
        "v_and_b32 %[rv], %[rv], %[sum]\n\t"
        "v_cmp_ne_u32 vcc, 0, %[rv]\n\t"
        "s_cbranch_vccnz 2b\n\t"
        "s_mov_b64 exec, %[s2]\n\t"
        "3:\n\t"
#if FGFA_TAG_ABLATE & 16
        "s_branch 4f\n\t"
#endif
        "v_cmp_lt_u32 vcc, 31, %[tt]\n\t"                     // the lanes whose run goes on into the next word
        "s_cbranch_vccz 4f\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "v_subrev_u32 %[tt], 32, %[tt]\n\t"
        "v_add_u32 %[w], 4, %[w]\n\t"
        "v_add_u32 %[base], 0x80, %[base]\n\t"
        "v_min_u32 %[m], 31, %[tt]\n\t"
        "v_sub_u32 %[m], 31, %[m]\n\t"
        "v_lshrrev_b32 %[mask], %[m], -1\n\t"
        "s_branch 1b\n\t"
        "4:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [n] "=&v"(n), [a] "=&v"(a), [w] "=&v"(w), [tt] "=&v"(tt), [m] "=&v"(m), [k] "=&v"(k), [mask] "=&v"(mask), [base] "=&v"(base),
          [old] "=&v"(old), [rv] "=&v"(rv), [low] "=&v"(low), [sum] "=&v"(sum), [f] "=&v"(f), [g] "=&v"(g), [sv] "=&s"(sv), [s2] "=&s"(s2)
        : [rec] "v"(rec), [sb] "v"(sb), [vm] "s"(vm), [cm] "s"(cm), [dbase] "s"(dbase), [rbase] "s"(rbase), [one] "v"(one), [mone] "v"(mone),
          [relmask] "i"(kRelMask), [basemask] "i"(kBaseMask), [wb] "i"(WB), [wb5] "i"(WB - 5), [nc] "i"(NC ? 1 : 0)
        : "vcc", "memory");
}

// A step none of whose records needs a claim: the two depth updates and nothing else.
template <int WB>
__device__ __forceinline__ void depth_step(uint32_t rec, unsigned long long vm, uint32_t dbase, uint32_t one, uint32_t mone) {
    constexpr uint32_t kRelMask = (1u << WB) - 1u;
    uint32_t n, a;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[vm]\n\t"
        "v_and_b32 %[a], %[relmask], %[rec]\n\t"
        "v_bfe_u32 %[n], %[rec], %[wb], 10\n\t"
        "v_lshl_add_u32 %[a], %[a], 2, %[dbase]\n\t"
        "ds_add_u32 %[a], %[one]\n\t"
        "v_lshl_add_u32 %[a], %[n], 2, %[a]\n\t"
        "ds_add_u32 %[a], %[mone] offset:4\n\t"
        "s_mov_b64 exec, %[sv]"
        : [n] "=&v"(n), [a] "=&v"(a), [sv] "=&s"(sv)
        : [rec] "v"(rec), [vm] "s"(vm), [dbase] "s"(dbase), [one] "v"(one), [mone] "v"(mone), [relmask] "i"(kRelMask), [wb] "i"(WB)
        : "memory");
}

// The same for records of one segment each (k_scan_dense's: ids without runs): two depth updates, one returning OR of
// the segment's bit, and under the lanes that found it set the revisit's two updates.  Eleven vector instructions,
// where the C++ rendering had two predicated regions with their exec bookkeeping.
template <int WB>
__device__ __forceinline__ void claim_point(uint32_t rec, uint32_t sb, unsigned long long vm, unsigned long long cm, uint32_t dbase, uint32_t rbase, uint32_t one, uint32_t mone) {
    constexpr uint32_t kRelMask = (1u << WB) - 1u;
    uint32_t rel, a, w, k, bit, old;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[vm]\n\t"
        "v_and_b32 %[rel], %[relmask], %[rec]\n\t"
        "v_lshl_add_u32 %[a], %[rel], 2, %[dbase]\n\t"
        "ds_add_u32 %[a], %[one]\n\t"
        "ds_add_u32 %[a], %[mone] offset:4\n\t"
        "s_mov_b64 exec, %[cm]\n\t"
        "v_bfe_u32 %[w], %[rec], 5, %[wb5]\n\t"
        "v_lshl_add_u32 %[w], %[w], 2, %[sb]\n\t"
        "v_and_b32 %[k], 31, %[rec]\n\t"
        "v_lshlrev_b32 %[bit], %[k], 1\n\t"
        "ds_or_rtn_b32 %[old], %[w], %[bit]\n\t"
        "v_lshl_add_u32 %[a], %[rel], 2, %[rbase]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_and_b32 %[old], %[old], %[bit]\n\t"
        "v_cmp_ne_u32 vcc, 0, %[old]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "ds_add_u32 %[a], %[one]\n\t"
        "ds_add_u32 %[a], %[mone] offset:4\n\t"
        "s_mov_b64 exec, %[sv]"
        : [rel] "=&v"(rel), [a] "=&v"(a), [w] "=&v"(w), [k] "=&v"(k), [bit] "=&v"(bit), [old] "=&v"(old), [sv] "=&s"(sv)
        : [rec] "v"(rec), [sb] "v"(sb), [vm] "s"(vm), [cm] "s"(cm), [dbase] "s"(dbase), [rbase] "s"(rbase), [one] "v"(one), [mone] "v"(mone),
          [relmask] "i"(kRelMask), [wb5] "i"(WB - 5)
        : "vcc", "scc", "memory");
}

// Pass 2 of a tagged call: every record of k_scan says whose it is (see kTagShift), so a wave
// walks whole sub-buckets, 64 consecutive records per step, three steps' records requested ahead
// of their use, and needs no directory: nothing to fetch before the first record, no mapping of
// records to items.  The waves of a workgroup take its sub-buckets from an LDS counter, one ahead
// (the first is their own index), so that none is left with the heavy ones.
// `bits` holds the workgroup's kAccWaves * kTagSlots private bitsets, then one per split path
// (shared by all waves; cleared by the kernel).  A private slot changes hands when a tag beyond
// the highest seen so far shows up (k_scan guarantees that the slot's previous owner, kTagSlots
// items earlier, has no record behind that point), and when the wave opens its next sub-bucket.
// The walk is bound by instruction issue -- a CU issues one scalar and one vector instruction per
// cycle for all its sixteen waves -- so the common step is kept short: the tags of a step are
// nearly always between the highest met so far and the tag of the step's last record, and then
// the hand-over is a couple of compares; anything else (items interleaved by waves that ran ahead,
// more items in a step than a wave has bitsets) takes the general route below it.
#ifndef FGFA_TAG_ABLATE
#define FGFA_TAG_ABLATE 0  /* measurements only (results are then wrong): 1 = no claims, 2 = no bitset hand-overs, 4 = claims stop behind the depth updates, 8 = behind the first word's OR, 16 = no further words */
#endif
// OWN: the wave keeps track of which tag owns each of its bitsets (see FGFA_TAG_MISS_OWN below) instead of taking them for the tags of
// (hmax - kSlots, hmax]: for sub-buckets whose tags lie far apart -- a workgroup of pass 1 that took hundreds of items few of which
// visit any one window -- where a step of 64 records holds three tags twenty apart and the other build claims them one after the other.
template <int WB, bool POINT, bool SHARED, bool LOW = false, int SLOTS = (int)kTagSlots, bool NC = false, bool OWN = false>  // (NC: the plan has items whose records need no claim, kTagNoClaim -- a build of its own: the test costs cfg-L 4 %)
__device__ __forceinline__ void apply_tagged(const AccArgs &A, int *D, int *R, uint32_t *bits, const uint2 *scnt2, const uint32_t *wbase,
                                             uint32_t *grab) {
    constexpr uint32_t kW = 1u << WB, kNW = kW / 32u;
    constexpr uint32_t kSlots = (uint32_t)SLOTS;  // private bitsets per wave: kTagSlots, or twice as many where the LDS allows (k_scan's order of the tags holds for any multiple)
    constexpr uint32_t kPriv = kAccWaves * kSlots;  // slot ids: the waves' private bitsets first, the shared ones behind
    const int lane = threadIdx.x & 63;
    const uint32_t lane4 = 4u * (uint32_t)lane;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t y16 = blockIdx.y * kAccWaves, nw = A.parts * kAccWaves;
    const uint32_t shlo = kTagCount - 1u - A.n_shared;  // tags from here up name split paths (and, the highest, items that need no claim)
    const uint32_t bits0 = lds_addr(bits), priv_b = bits0 + ((wv * kSlots) << (WB - 3));  // (LDS byte addresses)
    const uint32_t dbase = lds_addr(D), rbase = lds_addr(R);
    uint32_t one = 1u, mone = ~0u;  // (the LDS adds take their operand from a register)
    asm volatile("" : "+v"(one), "+v"(mone));
    // this workgroup's sub-buckets: the i-th is y16 + (i & 15) + (i >> 4) * nw, while that is below n_slots
    const uint32_t imax = A.n_slots > y16 ? kAccWaves * ((A.n_slots - y16 + nw - 1u) / nw) : 0u;
    const uint32_t *sp = wbase;  // (uniform) the next record of the open sub-bucket ...
    uint32_t left = 0;           // ... and how many it has left
    uint32_t plain = 0;          // ... and whether its private tags all have a bitset of their own (2) or not (0)
    uint32_t cur_i = wv;
    // (scnt2: {where k_scan's records start, counted from the window's first bucket; how many there are | plain << 31})
    const auto open = [&](uint32_t i) {
        const uint32_t s = y16 + (i & (kAccWaves - 1u)) + (i / kAccWaves) * nw;
        left = 0;
        if (i < imax && s < A.n_slots) {
            const uint2 c = scnt2[s];
            sp = wbase + __builtin_amdgcn_readfirstlane(c.x);
            const uint32_t n = __builtin_amdgcn_readfirstlane(c.y);
            left = n & 0x7FFFFFFFu;
            plain = (n >> 31) << 1;
        }
    };
    open(cur_i);
    // The sub-bucket after the open one is taken when that one is opened: the LDS round trip is
    // long over when it is needed.  (By hand: hipcc turns an atomicAdd by one lane into its wave-aggregated
    // form, a dozen instructions.)
    const uint32_t grab_a = lds_addr(grab);
    uint32_t nxt = 0;
#define FGFA_TAG_GRAB()                                                                                         \
    do {                                                                                                        \
        unsigned long long sv_;                                                                                 \
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1" \
                     : "+v"(nxt), "=&s"(sv_)                                                                    \
                     : "v"(grab_a), "v"(one)                                                                    \
                     : "memory");                                                                               \
    } while (0)
#define FGFA_TAG_TAKEN(OUT) asm volatile("s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %0, %1" : "=s"(OUT) : "v"(nxt) : "memory")
    FGFA_TAG_GRAB();
    // The next step of this wave's stream: up to 64 records of the open sub-bucket, or of the next one
    // that has any.  Lanes beyond the last record read what lies behind it (the bucket array ends
    // with a window nobody reads); a step behind the end of the stream is empty but still requests its load.
#define FGFA_TAG_GEN(K, NV, FR)                                  \
    do {                                                         \
        FR = 0u;                                                 \
        while (left == 0u && cur_i < imax) {                     \
            FGFA_TAG_TAKEN(cur_i);                               \
            FGFA_TAG_GRAB();                                     \
            open(cur_i);                                         \
            FR = 1u | plain;                                     \
        }                                                        \
        NV = min(64u, left);                                     \
        if (LOW) rec_request_lo<K>(sp, lane4);                   \
        else rec_request_s<K>(sp, lane4);                        \
        sp += NV;                                                \
        left -= NV;                                              \
    } while (0)
    int hmax = -1;  // (uniform) the highest private tag met in the open sub-bucket (the wave's first one is walked the general way: nothing says it is new)
    // OWN: (lane s < kSlots) the tag whose bits slot s holds, or kFree; `hmax` is then the highest tag that owns a slot, and
    // `dense` says that slot s is owned by the tag of (hmax - kSlots, hmax] congruent to s, for every s (or that no slot is owned at all)
    constexpr uint32_t kFree = ~0u;
    uint32_t ownv = kFree;
    bool dense = true;
    const auto clear_slots = [&](int from, int to) {  // the bitsets of the tags from .. to change hands
        for (int t = from; t <= to; ++t) {
            uint32_t *bs = bits + (wv * kSlots + ((uint32_t)t & (kSlots - 1u))) * kNW;
            for (uint32_t i = lane; i < kNW / 2u; i += 64) reinterpret_cast<uint2 *>(bs)[i] = make_uint2(0u, 0u);
        }
    };
    const auto claim = [&](uint32_t rec, uint32_t tag, unsigned long long act, unsigned long long cact, bool any_shared) {
        // each lane's bitset: its wave's slot tag mod kSlots, or its split path's  (cact: the lanes of act that claim at all)
        uint32_t sb = priv_b + ((tag & (kSlots - 1u)) << (WB - 3));
        if (SHARED && any_shared) sb = tag >= shlo ? bits0 + ((kPriv + kTagCount - 2u - tag) << (WB - 3)) : sb;
        if (POINT) {  // every record is one segment (k_scan_dense)
            claim_point<WB>(rec, sb, act, cact, dbase, rbase, one, mone);
        } else if (!(FGFA_TAG_ABLATE & 1)) {
            claim_step<WB, NC>(rec, sb, act, cact, dbase, rbase, one, mone);
        }
    };
    constexpr int kDepth = LOW ? 3 : kTagDepth;
    uint32_t nv0 = 0, nv1 = 0, nv2 = 0, nv3 = 0, nv4 = 0, nv5 = 0, nv6 = 0, nv7 = 0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0, f6 = 0, f7 = 0;
    FGFA_TAG_GEN(0, nv0, f0);
    FGFA_TAG_GEN(1, nv1, f1);
    FGFA_TAG_GEN(2, nv2, f2);
    if (kDepth > 3) FGFA_TAG_GEN((kDepth > 3 ? 3 : 0), nv3, f3);
    if (kDepth > 4) FGFA_TAG_GEN((kDepth > 4 ? 4 : 0), nv4, f4);
    if (kDepth > 5) FGFA_TAG_GEN((kDepth > 5 ? 5 : 0), nv5, f5);
    if (kDepth > 6) FGFA_TAG_GEN((kDepth > 6 ? 6 : 0), nv6, f6);
    if (kDepth > 7) FGFA_TAG_GEN((kDepth > 7 ? 7 : 0), nv7, f7);
    // OWN: the records of a step whose tag does not own its bitset (slot = tag mod kSlots, as ever) are taken a BURST at a time -- the
    // lowest such tag T and every other one within kSlots of it, up to T2: the slots of T .. T2 change hands (cleared; tags
    // of that stretch without a record get theirs too, as in the other build), and all of the step's records claim at once.  k_scan's
    // gate (a tag's records all lie before those of a tag kTagSlots beyond it) says the bitsets' previous owners are finished -- unless
    // their last records are in this very step, unclaimed: those claim first.  Tags twenty apart share a slot one
    // time in kSlots, where the other build takes every such step in as many claims as it has tags.
#define FGFA_TAG_MISS_OWN(REC, TAG, VM, CVM, SHM, PVM)                                                                 \
    do {                                                                                                               \
        constexpr uint32_t kMask_ = kSlots - 1u;                                                                       \
        unsigned long long miss_;                                                                                      \
        if (dense) {                                                                                                   \
            miss_ = PVM & ~__builtin_amdgcn_ballot_w64((uint32_t)(hmax - (int)TAG) < kSlots);                          \
            if (miss_) {                                                                                               \
                /* the common hand-over, as in the other build: the step's tags end with its highest and span fewer than */ \
                /* kSlots -- the bitsets of (hmax, c] change hands, and the owners are those of c's stretch            */ \
                const int c_ = (int)__builtin_amdgcn_readlane(TAG, 63 - (int)__builtin_clzll(PVM));                    \
                if ((__builtin_amdgcn_ballot_w64((int)TAG > c_ || (int)(TAG + kSlots) <= c_) & PVM) == 0ull) {         \
                    clear_slots(max(hmax + 1, c_ - (int)(kSlots - 1u)), c_);                                           \
                    hmax = c_;                                                                                         \
                    ownv = (uint32_t)c_ - (((uint32_t)c_ - (uint32_t)lane) & kMask_);                                  \
                    miss_ = 0ull;                                                                                      \
                }                                                                                                      \
            }                                                                                                          \
        } else {                                                                                                       \
            const uint32_t o_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((TAG & kMask_) << 2), (int)ownv);          \
            miss_ = PVM & ~__builtin_amdgcn_ballot_w64(o_ == TAG);                                                     \
        }                                                                                                              \
        unsigned long long todo_ = VM;                                                                                 \
        while (miss_) {                                                                                                \
            /* the lowest tag without a bitset first (not the first lane's: neighbouring items interleave) */          \
            const uint32_t t_ = wave_min_u32((miss_ >> lane) & 1ull ? TAG : ~0u);                                      \
            const uint32_t rel_ = TAG - t_;                                                                            \
            const unsigned long long bm_ = __builtin_amdgcn_ballot_w64(rel_ < kSlots) & miss_;                         \
            uint32_t t2_ = t_;                                                                                         \
            if (__builtin_amdgcn_ballot_w64(rel_ != 0u) & bm_) t2_ = wave_max_u32((bm_ >> lane) & 1ull ? TAG : 0u);    \
            const uint32_t n_ = t2_ - t_ + 1u;  /* 1 .. kSlots */                                                      \
            /* unclaimed records of older tags whose bitset the burst takes (their last: k_scan's gate): they claim first */ \
            const unsigned long long busy_ = __builtin_amdgcn_ballot_w64((int)TAG < (int)t_ && (rel_ & kMask_) < n_) & todo_ & PVM; \
            if (busy_) {                                                                                               \
                claim(REC, TAG, busy_, busy_ & CVM, false);                                                            \
                todo_ &= ~busy_;                                                                                       \
            }                                                                                                          \
            const uint32_t off_ = ((uint32_t)lane - t_) & kMask_;  /* (lane s stands for slot s) */                     \
            const uint32_t nt_ = t_ + off_;                                                                            \
            /* (a tag of the stretch that owns its slot already -- its records came first: waves of k_scan that ran ahead -- keeps it) */ \
            const bool upd_ = off_ < n_ && (uint32_t)lane < kSlots && ownv != nt_;                                     \
            unsigned long long um_ = __builtin_amdgcn_ballot_w64(upd_);                                                \
            if (__builtin_amdgcn_ballot_w64(upd_ && ownv != kFree && (int)(ownv + kTagSlots) > (int)nt_))              \
                atomicOr(A.status, kStInternal);  /* cannot happen: k_scan's gate */                                   \
            ownv = upd_ ? nt_ : ownv;                                                                                  \
            while (um_) {                                                                                              \
                uint32_t *bs_ = bits + (wv * kSlots + (uint32_t)__builtin_ctzll(um_)) * kNW;                           \
                um_ &= um_ - 1ull;                                                                                     \
                for (uint32_t i_ = lane; i_ < kNW / 2u; i_ += 64) reinterpret_cast<uint2 *>(bs_)[i_] = make_uint2(0u, 0u); \
            }                                                                                                          \
            hmax = max(hmax, (int)t2_);                                                                                \
            const uint32_t wt_ = (uint32_t)hmax - (((uint32_t)hmax - (uint32_t)lane) & kMask_);                        \
            dense = (__builtin_amdgcn_ballot_w64(ownv == wt_) & ((1ull << kSlots) - 1ull)) == ((1ull << kSlots) - 1ull); \
            miss_ &= ~bm_;                                                                                             \
        }                                                                                                              \
        claim(REC, TAG, todo_, todo_ & CVM, SHM != 0ull);                                                              \
    } while (0)
#define FGFA_TAG_STEP(K, NV, FR)                                                                                       \
    if (NV == 0u) break;                                                                                               \
    {                                                                                                                  \
        const uint32_t rec = LOW ? rec_take_lo<K>() : rec_take_s<K>();                                                 \
        const unsigned long long vm = NV >= 64u ? ~0ull : (1ull << NV) - 1ull;  /* the lanes that hold a record */     \
        const uint32_t tag = rec >> kTagShift;                                                                         \
        /* the lanes whose item never meets a segment twice (kTagNoClaim): depth updates, no bitset */                 \
        const unsigned long long cvm = NC ? vm & ~__builtin_amdgcn_ballot_w64(tag == kTagNoClaim) : vm;                \
        if (FR & 2u) {  /* a sub-bucket whose private tags each have a slot: all of them cleared here, none changes hands */ \
            if (cvm) {                                                                                                 \
                for (uint32_t i = lane; i < kSlots * kNW / 4u; i += 64) reinterpret_cast<uint4 *>(bits + wv * kSlots * kNW)[i] = make_uint4(0u, 0u, 0u, 0u); \
                hmax = (int)kSlots - 1;                                                                                \
                if (OWN) ownv = (uint32_t)lane;                                                                        \
            } else {                                                                                                   \
                hmax = -1;  /* (nothing to claim in its first step: the slots are cleared as their tags show up) */    \
                if (OWN) ownv = kFree;                                                                                 \
            }                                                                                                          \
            if (OWN) dense = true;                                                                                     \
        } else if (FR) {                                                                                               \
            hmax = -1;  /* the private slots start over with this sub-bucket */                                        \
            if (OWN) ownv = kFree, dense = true;                                                                       \
        }                                                                                                              \
        FGFA_TAG_GEN(K, NV, FR);                                                                                       \
        if (NC && cvm == 0ull) {                                                                                       \
            depth_step<WB>(rec, vm, dbase, one, mone);                                                                 \
        } else {                                                                                                       \
        const unsigned long long shm = SHARED ? __builtin_amdgcn_ballot_w64(tag >= shlo) & cvm : 0ull;                 \
        const unsigned long long pvm = cvm & ~shm;  /* the lanes whose tag names an item of their own */               \
        if constexpr (OWN) {                                                                                           \
            FGFA_TAG_MISS_OWN(rec, tag, vm, cvm, shm, pvm);                                                            \
        } else {                                                                                                       \
        bool general = false;                                                                                          \
        if (!(FGFA_TAG_ABLATE & 2) && (__builtin_amdgcn_ballot_w64((uint32_t)hmax - tag >= kSlots) & pvm)) {        \
            /* tags beyond those met so far (or, which cannot be, kSlots behind): their bitsets change hands. */     \
            /* The last private record's tag is the highest unless waves that ran ahead have interleaved the items */  \
            const int c = (int)__builtin_amdgcn_readlane(tag, 63 - (int)__builtin_clzll(pvm));                         \
            general = (__builtin_amdgcn_ballot_w64((int)tag > c || (int)(tag + kSlots) <= c) & pvm) != 0ull;           \
            if (!general) {                                                                                            \
                clear_slots(max(hmax + 1, c - (int)(kSlots - 1u)), c);                                              \
                hmax = c;                                                                                              \
            }                                                                                                          \
        }                                                                                                              \
        if (!general) {                                                                                                \
            claim(rec, tag, vm, cvm, shm != 0ull);                                                                     \
        } else {                                                                                                       \
            unsigned long long todo = vm;                                                                              \
            do {                                                                                                       \
                unsigned long long act = todo;                                                                         \
                const bool pv = ((todo & pvm) >> lane) & 1ull;                                                         \
                if (todo & pvm) {                                                                                      \
                    /* the lanes before the first one whose tag is kSlots beyond the lowest go first: */            \
                    /* records of such tags lie in order */                                                            \
                    const uint32_t tmin = wave_min_u32(pv ? tag : ~0u);                                                \
                    const unsigned long long beyond = __builtin_amdgcn_ballot_w64(pv && tag >= tmin + kSlots);      \
                    if (beyond) act = todo & ((1ull << __builtin_ctzll(beyond)) - 1ull);                               \
                    if ((int)(tmin + kSlots) <= hmax || !act) {  /* cannot happen: k_scan's gate */                 \
                        atomicOr(A.status, kStInternal);                                                               \
                        act = todo;                                                                                    \
                    }                                                                                                  \
                    const int hnew = max(hmax, (int)wave_max_u32(((act & pvm) >> lane) & 1ull ? tag : 0u));            \
                    clear_slots(max(hmax + 1, hnew - (int)(kSlots - 1u)), hnew);                                    \
                    hmax = hnew;                                                                                       \
                }                                                                                                      \
                claim(rec, tag, act, act & cvm, shm != 0ull);                                                          \
                todo &= ~act;                                                                                          \
            } while (todo);                                                                                            \
        }                                                                                                              \
        }                                                                                                              \
        }                                                                                                              \
    }
    while (true) {
        FGFA_TAG_STEP(0, nv0, f0)
        FGFA_TAG_STEP(1, nv1, f1)
        FGFA_TAG_STEP(2, nv2, f2)
        if (kDepth > 3) { FGFA_TAG_STEP((kDepth > 3 ? 3 : 0), nv3, f3) }
        if (kDepth > 4) { FGFA_TAG_STEP((kDepth > 4 ? 4 : 0), nv4, f4) }
        if (kDepth > 5) { FGFA_TAG_STEP((kDepth > 5 ? 5 : 0), nv5, f5) }
        if (kDepth > 6) { FGFA_TAG_STEP((kDepth > 6 ? 6 : 0), nv6, f6) }
        if (kDepth > 7) { FGFA_TAG_STEP((kDepth > 7 ? 7 : 0), nv7, f7) }
    }
#undef FGFA_TAG_STEP
#undef FGFA_TAG_MISS_OWN
#undef FGFA_TAG_GEN
#undef FGFA_TAG_GRAB
#undef FGFA_TAG_TAKEN
}

// the sum of a 64-bit value over the wave, uniform, by DPP adds on its halves
__device__ __forceinline__ unsigned long long wave_total_u64(unsigned long long x) {
#define FGFA_DPP_ADD64(CTRL, ROWMASK, BC)                                                                            \
    x += ((unsigned long long)(uint32_t)__builtin_amdgcn_update_dpp(0u, (uint32_t)(x >> 32), CTRL, ROWMASK, 0xf, BC) << 32) | \
         (uint32_t)__builtin_amdgcn_update_dpp(0u, (uint32_t)x, CTRL, ROWMASK, 0xf, BC)
    FGFA_DPP_ADD64(0x111 /* row_shr:1 */, 0xf, true);
    FGFA_DPP_ADD64(0x112 /* row_shr:2 */, 0xf, true);
    FGFA_DPP_ADD64(0x114 /* row_shr:4 */, 0xf, true);
    FGFA_DPP_ADD64(0x118 /* row_shr:8 */, 0xf, true);
    FGFA_DPP_ADD64(0x142 /* row_bcast:15 */, 0xa, true);
    FGFA_DPP_ADD64(0x143 /* row_bcast:31 */, 0xc, true);
#undef FGFA_DPP_ADD64
    return ((unsigned long long)__builtin_amdgcn_readlane((uint32_t)(x >> 32), 63) << 32) | __builtin_amdgcn_readlane((uint32_t)x, 63);
}

// Eight such sums at once: x[k] holds this lane's share of sum k; every lane returns the wave's
// total of sum (lane & 7).  Three butterfly steps that halve the number of values a lane holds
// (a lane keeps the sums whose index agrees with its own on bit j and passes the others to its
// partner 2^j lanes away) and three that add what is left across the groups of eight: 18
// instructions per sum where eight reductions of their own take 50.
#define FGFA_DPP64(V, CTRL)                                                                                              \
    (((unsigned long long)(uint32_t)__builtin_amdgcn_update_dpp(0u, (uint32_t)((V) >> 32), CTRL, 0xf, 0xf, true) << 32) | \
     (uint32_t)__builtin_amdgcn_update_dpp(0u, (uint32_t)(V), CTRL, 0xf, 0xf, true))
__device__ __forceinline__ unsigned long long wave_totals8_u64(const unsigned long long (&x)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    unsigned long long y[4], z[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long keep = b0 ? x[2 * j + 1] : x[2 * j], send = b0 ? x[2 * j] : x[2 * j + 1];
        y[j] = keep + FGFA_DPP64(send, 0xB1 /* quad_perm:[1,0,3,2] */);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned long long keep = b1 ? y[2 * j + 1] : y[2 * j], send = b1 ? y[2 * j] : y[2 * j + 1];
        z[j] = keep + FGFA_DPP64(send, 0x4E /* quad_perm:[2,3,0,1] */);
    }
    const unsigned long long keep = b2 ? z[1] : z[0], send = b2 ? z[0] : z[1];
    const unsigned long long up = FGFA_DPP64(send, 0x104 /* row_shl:4: from the lane four above */), dn = FGFA_DPP64(send, 0x114 /* row_shr:4: from four below */);
    unsigned long long w = keep + (b2 ? dn : up);      // sum (lane & 7) over the lane's group of eight
    w += FGFA_DPP64(w, 0x128 /* row_ror:8 */);        // ... over its row of sixteen
    w += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(w >> 32), 16, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)w, 16, 64);
    w += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(w >> 32), 32, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)w, 32, 64);
    return w;
}
#undef FGFA_DPP64

// measure_path (depth.rs:116-131) without a second walk of the steps: a record (first segment,
// length) of path p contributes sum(len) and sum(depth * len) over its segments, which are two
// differences of the window's prefix sums Lw / Ww (built in LDS once the window's depth is
// final).  Each wave walks its stretch of k_scan's items as in apply_groups; the first 64 records
// of eight items are requested before any is used; an item's contribution in this window is
// reduced across the wave and stored -- plain stores: an atomic per item would sit in the way of
// the loads behind it until memory had acknowledged it -- and k_path_reduce adds the windows up.
template <int WB>
__device__ __forceinline__ void sum_groups(const AccArgs &A, const ulonglong2 *LW,
                                           const uint32_t *wbase, uint32_t win, uint32_t e0, uint32_t e1) {
    constexpr uint32_t kW = 1u << WB;
    constexpr int kAhead = 8;
    const int lane = threadIdx.x & 63;
    ulonglong2 *part = A.psum_part + (size_t)win * A.dstride;
    const auto add = [&](uint32_t rec, unsigned long long &ls, unsigned long long &ws) {
        const uint32_t rel = rec & (kW - 1), e1x = rel + ((rec >> WB) & 1023u) + 1u;  // one past the run's last segment
        const ulonglong2 hi = LW[e1x], lo = LW[rel];  // (both sums of a prefix side by side: two 16-byte LDS reads per record, not four of 8)
        ls += hi.x - lo.x;
        ws += hi.y - lo.y;
    };
    for (uint32_t mb = e0; mb < e1; mb += 64u) {
        const uint32_t cntE = min(64u, e1 - mb);
        const bool have = (uint32_t)lane < cntE;
        const uint32_t at = mb + (uint32_t)lane;
        const uint2 be = have ? A.dir[(size_t)win * A.dstride + at] : make_uint2(0u, 0u);
        const uint32_t sl = have ? A.islot[at] & 0x7FFFFFFFu : 0u;
        const uint32_t b = min(be.x, A.cap), en = max(b, min(be.y, A.cap));
        const uint32_t n = en - b, off = sl * A.cap + b;
        unsigned long long myL = 0, myW = 0;  // lane i: item i's sums in this window
        for (uint32_t i0 = 0; i0 < cntE; i0 += kAhead) {
            uint32_t r[kAhead], nn[kAhead], oo[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const uint32_t i = min(i0 + (uint32_t)k, cntE - 1u);
                nn[k] = i0 + k < cntE ? __builtin_amdgcn_readlane(n, i) : 0u;
                oo[k] = __builtin_amdgcn_readlane(off, i);
                r[k] = wbase[oo[k] + ((uint32_t)lane < nn[k] ? (uint32_t)lane : 0u)];  // unconditional: a predicated load would be waited for on the spot
            }
            unsigned long long ls[kAhead], ws[kAhead];  // this lane's share of each of the eight items' sums
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                ls[k] = ws[k] = 0ull;
                if (nn[k] == 0u) continue;
                if ((uint32_t)lane < nn[k]) add(r[k], ls[k], ws[k]);
                for (uint32_t c = 64u; c < nn[k]; c += 64u)
                    if (c + (uint32_t)lane < nn[k]) add(wbase[oo[k] + c + lane], ls[k], ws[k]);
            }
            // all eight reduced together (by DPP: __shfl_down would go through LDS twelve times per value);
            // lane i0 + k -- i0 is a multiple of eight -- finds item k's totals in its own registers
            const unsigned long long tl = wave_totals8_u64(ls, lane), tw = wave_totals8_u64(ws, lane);
            if ((uint32_t)lane >= i0 && (uint32_t)lane < i0 + (uint32_t)kAhead) {
                myL = tl;
                myW = tw;
            }
        }
        if (have) part[at] = make_ulonglong2(myL, myW);
    }
}

// Adds an item's per-window sums up and credits them to its path.  One wave per item.
__global__ __launch_bounds__(256) void k_path_reduce(const uint4 *__restrict__ items, const uint32_t *__restrict__ elist, uint32_t n_items, uint32_t n_win,
                                                     uint32_t dstride, const ulonglong2 *__restrict__ part,
                                                     unsigned long long *__restrict__ psum_len,
                                                     unsigned long long *__restrict__ psum_w) {
    const uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (j >= n_items) return;
    unsigned long long l = 0, w = 0;
    for (uint32_t wdw = lane; wdw < n_win; wdw += 64u) {
        const ulonglong2 v = part[(size_t)wdw * dstride + j];
        l += v.x;
        w += v.y;
    }
    l = wave_total_u64(l);
    w = wave_total_u64(w);
    if (lane == 0 && (l | w)) {
        const uint32_t p = items[elist[j] & 0x7FFFFFFFu].w;  // (j is a position in pass 2's walk order)
        atomicAdd(&psum_len[p], l);
        atomicAdd(&psum_w[p], w);
    }
}

// the "seen" bitsets of a tagged call: dynamic shared memory, (kAccWaves * kTagSlots + n_shared) * window / 8 bytes
extern __shared__ __attribute__((aligned(16))) uint32_t tag_bits[];

// PAIR (tagged, unique depth, no split paths): TWO workgroups per window, each with half of its
// sub-buckets, and both resident on a CU (64 registers, under 80 KB of LDS): the walk issues about
// one instruction per cycle and CU where two are possible, and eight waves per SIMD hide more of its
// LDS round trips than four.  Both leave their partial vectors in scratch; the second one to
// finish adds the other's to its own and writes the results.
template <bool UNIQ, int WB, bool PSUM, bool POINT, bool BIG, bool TAGGED, bool PAIR, int SLOTS = (int)kTagSlots, bool LOWREG = PAIR, bool NC = false, bool OWN = false>
__device__ __forceinline__ void accum_body(const AccArgs &A) {
    constexpr uint32_t kW = 1u << WB;
    constexpr int kPer = kW / kAccThreads;  // cells per thread: 4 or 8
    constexpr uint32_t kSlots = WB <= 12 ? 8u : 4u;
    // difference arrays over the window: depth, and (with unique depth) revisits
    __shared__ __attribute__((aligned(16))) int cells[(UNIQ ? 2 : 1) * (kW + 64)];
    __shared__ unsigned long long wave_tot[kAccWaves];
    __shared__ uint32_t scnt[kMaxSlots];
    __shared__ uint32_t sstart[kMaxSlots];  // where each sub-bucket starts, counted from wbase (s * cap, or a packed plan's own table)
    __shared__ __attribute__((aligned(8))) uint2 scnt2[TAGGED && UNIQ ? kMaxSlots : 1];  // tagged: where k_scan's records start and end in each sub-bucket
    __shared__ __attribute__((aligned(16))) uint32_t bits[UNIQ && !TAGGED ? kAccWaves * kSlots * (kW / 32) : 4];
    __shared__ uint32_t marks[UNIQ && !TAGGED ? kAccWaves * 64 : 4];
    __shared__ __attribute__((aligned(8))) uint32_t pend[UNIQ && !TAGGED ? kAccWaves * 3 * kPend : 4];
    __shared__ uint32_t grab;  // tagged: the next of the workgroup's sub-buckets nobody has taken yet
    __shared__ __attribute__((aligned(16))) ulonglong2 LW[PSUM ? kW + 1 : 1];  // prefix sums of len and of depth * len, side by side
    int *D = cells, *R = cells + (UNIQ ? kW + 64 : 0);
    const int tid = threadIdx.x, wave = tid >> 6;
    AccTimer tm;
    tm.start(A.tprof);
    const uint32_t win = blockIdx.x, w0 = win * kW;
    // this window's record counts, one per sub-bucket: staged in LDS, and zeroed in place so that
    // the scratch is clean for the next call.  With unique depth, k_scan's records are found
    // through the directory; the counts staged are those of the records that came before them.
    const bool flat = !UNIQ || A.has_pre;  // (without unique depth every record is applied as it is, tagged or not)
    const uint32_t nw = A.parts * kAccWaves;  // a window's sub-buckets are shared out to the waves of its A.parts workgroups
    // The first round of directory entries of this wave's items is requested right away: it is on
    // its way while the workgroup clears its arrays (a round trip of the twelve microseconds a
    // launch costs before it has counted anything).
    const uint32_t vwave = __builtin_amdgcn_readfirstlane(blockIdx.y * kAccWaves + wave);
    uint32_t ge0 = 0, ge1 = 0, slf_first = 0;
    uint2 be_first = make_uint2(0u, 0u);
    if (UNIQ && !TAGGED) {
        ge0 = __builtin_amdgcn_readfirstlane(A.wave_off[vwave]);
        ge1 = __builtin_amdgcn_readfirstlane(A.wave_off[vwave + 1]);
        const uint32_t nback = A.has_pre ? min(__builtin_amdgcn_readfirstlane(*A.work_counter), A.max_back) : 0u;
        const uint32_t nst = ge1 - ge0, nE = nst + (nback > vwave ? (nback - vwave + nw - 1u) / nw : 0u);
        const uint32_t x = (uint32_t)(tid & 63);
        const uint32_t at = x < nst ? ge0 + x : A.n_items + vwave + nw * (x - nst);
        if (x < nE) {
            be_first = A.dir[(size_t)win * A.dstride + at];
            slf_first = A.islot[at];
        }
    }
    for (uint32_t sl = tid; sl < A.n_slots; sl += kAccThreads) {
        if ((sl % nw) / kAccWaves != blockIdx.y) {  // (whoever walks a sub-bucket reads and clears its count)
            sstart[sl] = A.pk ? A.pk[(size_t)win * A.n_slots + sl].x : sl * A.cap;  // (apply_flat's unconditional loads may look at sub-bucket 0)
            continue;
        }
        uint32_t *c = A.counts + (size_t)win * A.n_slots + sl;
        uint32_t v = *c;
        *c = 0u;
        // where the sub-bucket starts (counted from wbase) and how much room it has
        uint32_t start = sl * A.cap, room = A.cap;
        if (A.pk) {
            const uint2 e = A.pk[(size_t)win * A.n_slots + sl];
            start = e.x;
            room = e.y;
        } else if (v > (A.cap >> 1)) {
            // (a sub-bucket more than half full: flatgfa_dev_status makes room before a later call -- whose
            // items k_scan may deal to other workgroups -- runs out of it)
            atomicMax(A.status + 2, v);
            if (A.fullest) atomicMax(A.fullest, v);  // (which range of the plan it was: only that one is given more room)
        }
        sstart[sl] = start;
        if (TAGGED && UNIQ) {
            const uint32_t c1 = min(v, room);
            v = min(A.has_pre == 1 ? A.counts0[(size_t)win * A.n_slots + sl] : A.has_pre ? v : 0u, c1);  // (2: k_scan did not run, all are earlier records)
            // (bit 31 of the count: the sub-bucket's private tags are 0 .. kTagSlots - 1 at most, so no bitset changes hands inside it)
            scnt2[sl] = make_uint2(start + v, (c1 - v) | (A.taken && A.taken[sl] <= (uint32_t)SLOTS ? 0x80000000u : 0u));  // k_scan's records: where they start in the window's buckets, how many
        } else if (UNIQ && A.has_pre == 1) {
            v = A.counts0[(size_t)win * A.n_slots + sl];
        }
        scnt[sl] = min(v, room);
    }
    if (tid == 0) grab = kAccWaves;
    if (TAGGED && UNIQ)  // the split paths' bitsets (the private ones are cleared when they change hands)
        for (uint32_t i = tid; i < A.n_shared * (kW / 32); i += kAccThreads) tag_bits[kAccWaves * (uint32_t)SLOTS * (kW / 32) + i] = 0u;
    const uint32_t nvalid = min(kW, A.n_segs - w0);
    for (uint32_t i = tid; i < (UNIQ ? 2u : 1u) * (kW + 64); i += kAccThreads) cells[i] = 0;
    __syncthreads();
    const uint32_t *wbase = A.pk ? A.buckets : A.buckets + (size_t)win * A.n_slots * A.cap;  // (a packed plan's starts count from the array's)
    tm.mark(0);
    if (flat) apply_flat<UNIQ, WB, (LOWREG ? 4 : 16)>(A, D, R, scnt, sstart, wbase);
    tm.mark(1);
    if (UNIQ && TAGGED) {
        if (PAIR) apply_tagged<WB, POINT, false, true, (int)kTagSlots, NC>(A, D, R, tag_bits, scnt2, wbase, &grab);
        else if (LOWREG && A.n_shared) apply_tagged<WB, POINT, true, true, SLOTS, NC>(A, D, R, tag_bits, scnt2, wbase, &grab);
        else if (LOWREG) apply_tagged<WB, POINT, false, true, SLOTS, NC>(A, D, R, tag_bits, scnt2, wbase, &grab);
        else if (A.n_shared) apply_tagged<WB, POINT, true, false, SLOTS, NC, OWN>(A, D, R, tag_bits, scnt2, wbase, &grab);
        else apply_tagged<WB, POINT, false, false, SLOTS, NC, OWN>(A, D, R, tag_bits, scnt2, wbase, &grab);
    } else if (UNIQ) {
        apply_groups<WB, false, POINT, BIG>(A, D, R, bits + wave * (kSlots * (kW / 32)), marks + wave * 64, pend + wave * (3 * kPend), wbase, win,
                                ge0, ge1, true, be_first, slf_first);
        // the long paths, one after the other, all waves on each: the bitset is slot 1 of wave 0's
        for (uint32_t f = A.fat_off[blockIdx.y]; f < A.fat_off[blockIdx.y + 1]; ++f) {
            __syncthreads();
            for (uint32_t i = tid; i < kW / 32; i += kAccThreads) bits[kW / 32 + i] = 0u;
            __syncthreads();
            const uint32_t *wo = A.fat_woff + (size_t)f * (kAccWaves + 1) + wave;
            apply_groups<WB, true, POINT, BIG>(A, D, R, bits, marks + wave * 64, pend + wave * (3 * kPend), wbase, win,
                                   __builtin_amdgcn_readfirstlane(wo[0]), __builtin_amdgcn_readfirstlane(wo[1]));
        }
    }
    tm.mark(2);
    __syncthreads();
    tm.mark(3);
    const uint32_t i0 = kPer * tid;
    uint32_t d[kPer], u[kPer];
    if (UNIQ) {
        // one scan for both: depth in the low word, revisits in the high word of a 64-bit value
        // (every prefix has both counts non-negative, so the words do not disturb each other)
        unsigned long long v[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) v[k] = (unsigned long long)(long long)D[i0 + k] + ((unsigned long long)(long long)R[i0 + k] << 32);
        block_scan<unsigned long long, kPer>(wave_tot, v);
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            d[k] = (uint32_t)v[k];
            u[k] = d[k] - (uint32_t)(v[k] >> 32);
        }
        if (PAIR) {
            // this workgroup's half: depth and revisits, in scratch; whoever finds the other half there adds it up
            __shared__ uint32_t second;
            uint32_t rv[kPer];
#pragma unroll
            for (int k = 0; k < kPer; ++k) rv[k] = (uint32_t)(v[k] >> 32);
            uint32_t *mine = A.pair_part + ((size_t)win * 2u + blockIdx.y) * (2u * kW);
            const uint32_t *theirs = A.pair_part + ((size_t)win * 2u + (1u - blockIdx.y)) * (2u * kW);
            // The two workgroups may sit on different XCDs, whose L2s do not see each other's lines within
            // a kernel: the halves are written and read with device-scope accesses (they go through to
            // memory), the writes are waited for, and only then is the half counted -- a release fence
            // would write the whole L2 back instead (tried: 0.25 ms per launch).
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                __hip_atomic_store(mine + i0 + k, d[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + kW + i0 + k, rv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) second = __hip_atomic_fetch_add(&A.pair_flag[win], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (second) {
#pragma unroll
                for (int k = 0; k < kPer; ++k) {
                    d[k] += __hip_atomic_load(theirs + i0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    rv[k] += __hip_atomic_load(theirs + kW + i0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < kPer; ++k) u[k] = d[k] - rv[k];
                if (A.accumulate) {
                    add_n<kPer>(A.depth_out + w0, i0, nvalid, d);
                    add_n<kPer>(A.uniq_out + w0, i0, nvalid, u);
                } else {
                    store_n<kPer>(A.depth_out + w0, i0, nvalid, d);
                    store_n<kPer>(A.uniq_out + w0, i0, nvalid, u);
                }
                if (tid == 0) __hip_atomic_store(&A.pair_flag[win], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (clean for the next call)
            }
        } else if (A.parts > 1 || A.accumulate) {
            add_n<kPer>(A.depth_out + w0, i0, nvalid, d);
            add_n<kPer>(A.uniq_out + w0, i0, nvalid, u);
        } else {
            store_n<kPer>(A.depth_out + w0, i0, nvalid, d);
            store_n<kPer>(A.uniq_out + w0, i0, nvalid, u);
        }
        tm.mark(4);
        tm.finish();
    } else {
        int v[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) v[k] = D[i0 + k];
        block_scan<int, kPer>(reinterpret_cast<int *>(wave_tot), v);
#pragma unroll
        for (int k = 0; k < kPer; ++k) d[k] = (uint32_t)v[k];
        if (!PSUM && (A.parts > 1 || A.accumulate)) add_n<kPer>(A.depth_out + w0, i0, nvalid, d);
        else store_n<kPer>(A.depth_out + w0, i0, nvalid, d);
        if (PSUM) {
            unsigned long long l[kPer], w[kPer];
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const uint32_t len = i0 + k < nvalid ? A.seg_len[w0 + i0 + k] : 0u;
                l[k] = len;
                w[k] = (unsigned long long)d[k] * len;
            }
            block_scan<unsigned long long, kPer>(wave_tot, l);
            block_scan<unsigned long long, kPer>(wave_tot, w);
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                LW[i0 + k + 1] = make_ulonglong2(l[k], w[k]);
            }
            if (tid == 0) LW[0] = make_ulonglong2(0ull, 0ull);
            __syncthreads();
            sum_groups<WB>(A, LW, wbase, win, __builtin_amdgcn_readfirstlane(A.wave_off[vwave]), __builtin_amdgcn_readfirstlane(A.wave_off[vwave + 1]));
            for (uint32_t f = A.fat_off[blockIdx.y]; f < A.fat_off[blockIdx.y + 1]; ++f) {
                const uint32_t *wo = A.fat_woff + (size_t)f * (kAccWaves + 1) + wave;
                sum_groups<WB>(A, LW, wbase, win, __builtin_amdgcn_readfirstlane(wo[0]), __builtin_amdgcn_readfirstlane(wo[1]));
            }
        }
    }
}

template <bool UNIQ, int WB, bool PSUM = false, bool POINT = false, bool BIG = false, bool TAGGED = false, int SLOTS = (int)kTagSlots, bool NC = false, bool OWN = false>
__global__ __launch_bounds__(kAccThreads) void k_accum(const AccArgs A) {
    accum_body<UNIQ, WB, PSUM, POINT, BIG, TAGGED, false, SLOTS, false, NC, OWN>(A);
}
#ifdef FGFA_MEASURE  // (measurement builds only, tools/variants.sh: profiles/NOTES.md R4.6, R4.9 -- no plan of the product library picks them)
template <int WB, bool POINT>
__global__ __launch_bounds__(kAccThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_accum_pair(const AccArgs A) {
    accum_body<true, WB, false, POINT, false, true, true, (int)kTagSlots, true, true>(A);
}
// Half-size windows, TWO of them resident on a CU (64 registers): eight waves per SIMD without a second
// workgroup's setup, scan and exchange per window.
template <int WB>
__global__ __launch_bounds__(kAccThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_accum_small(const AccArgs A) {
    accum_body<true, WB, false, false, false, true, false, (int)kTagSlots, true, true>(A);
}
#endif

}  // namespace

bool accum_kernels_setup() {
    // pass 2 of a tagged call keeps its bitsets in dynamic shared memory (next to about 60 KB of static arrays, 93 KB with 8192-segment windows)
    static OncePerDevice once;
    const bool ok = once([] {
        bool good = true;
        const auto set = [&](const void *k, uint32_t bytes) { good = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess && good; };
        set((const void *)k_accum<true, 11, false, false, false, true>, tagged_lds_bytes(11, kMaxShared));
        set((const void *)k_accum<true, 12, false, false, false, true>, tagged_lds_bytes(12, kMaxShared));
        set((const void *)k_accum<true, 12, false, true, false, true>, tagged_lds_bytes(12, kMaxShared));
        set((const void *)k_accum<true, 12, false, false, false, true, 8>, tagged_lds_bytes(12, 64, 8));
        set((const void *)k_accum<true, 13, false, false, false, true>, tagged_lds_bytes(13, 0));
        set((const void *)k_accum<true, 13, false, true, false, true>, tagged_lds_bytes(13, 0));
        // (... and their builds for plans with items that need no claim)
        set((const void *)k_accum<true, 11, false, false, false, true, (int)kTagSlots, true>, tagged_lds_bytes(11, kMaxShared));
        set((const void *)k_accum<true, 12, false, false, false, true, (int)kTagSlots, true>, tagged_lds_bytes(12, kMaxShared));
        set((const void *)k_accum<true, 12, false, true, false, true, (int)kTagSlots, true>, tagged_lds_bytes(12, kMaxShared));
        set((const void *)k_accum<true, 12, false, false, false, true, 8, true>, tagged_lds_bytes(12, 64, 8));
        set((const void *)k_accum<true, 13, false, false, false, true, (int)kTagSlots, true>, tagged_lds_bytes(13, 0));
        set((const void *)k_accum<true, 13, false, true, false, true, (int)kTagSlots, true>, tagged_lds_bytes(13, 0));
        // (... and the builds that keep track of the bitsets' owners, for sub-buckets of sparse tags: FastPlan::acc_own)
        set((const void *)k_accum<true, 12, false, false, false, true, (int)kTagSlots, false, true>, tagged_lds_bytes(12, kMaxShared));
        set((const void *)k_accum<true, 12, false, false, false, true, 8, false, true>, tagged_lds_bytes(12, 64, 8));
        set((const void *)k_accum<true, 13, false, false, false, true, (int)kTagSlots, false, true>, tagged_lds_bytes(13, 0));
        set((const void *)k_accum<true, 12, false, false, false, true, (int)kTagSlots, true, true>, tagged_lds_bytes(12, kMaxShared));
        set((const void *)k_accum<true, 12, false, false, false, true, 8, true, true>, tagged_lds_bytes(12, 64, 8));
        set((const void *)k_accum<true, 13, false, false, false, true, (int)kTagSlots, true, true>, tagged_lds_bytes(13, 0));
#ifdef FGFA_MEASURE
        set((const void *)k_accum_small<11>, tagged_lds_bytes(11, kMaxShared));
        set((const void *)k_accum_pair<12, false>, tagged_lds_bytes(12, 0));
        set((const void *)k_accum_pair<12, true>, tagged_lds_bytes(12, 0));
#endif
        return good;
    });
    if (!ok) set_error("hipFuncSetAttribute(k_accum): dynamic shared memory");
    return ok;
}

// (a tagged call's pass 2 in the build for the plan's items: NC where some of them need no claim, kTagNoClaim)
#define FGFA_TAGGED_LAUNCH(GRID, LDS, WB_, POINT_, SLOTS_)                                                                      \
    do {                                                                                                                       \
        if (nc) hipLaunchKernelGGL((k_accum<true, WB_, false, POINT_, false, true, SLOTS_, true>), GRID, dim3(kAccThreads), LDS, stream, aa);  \
        else hipLaunchKernelGGL((k_accum<true, WB_, false, POINT_, false, true, SLOTS_, false>), GRID, dim3(kAccThreads), LDS, stream, aa);    \
    } while (0)
// (... that keeps track of the bitsets' owners: plans whose sub-buckets hold sparse tags, FastPlan::acc_own)
#define FGFA_OWN_LAUNCH(GRID, LDS, WB_, SLOTS_)                                                                                \
    do {                                                                                                                       \
        if (nc) hipLaunchKernelGGL((k_accum<true, WB_, false, false, false, true, SLOTS_, true, true>), GRID, dim3(kAccThreads), LDS, stream, aa);  \
        else hipLaunchKernelGGL((k_accum<true, WB_, false, false, false, true, SLOTS_, false, true>), GRID, dim3(kAccThreads), LDS, stream, aa);    \
    } while (0)
void launch_accum(const FastPlan &fp, AccArgs &aa, bool uniq, bool tagged, bool psum, hipStream_t stream) {
    const dim3 agrid(fp.n_win, fp.acc_parts);
    const bool pair = uniq && tagged && fp.acc_pair;
    const bool nc = fp.n_noclaim != 0 || fp.n_flag_chunks != 0;
#ifdef FGFA_MEASURE
    if (pair) {
        aa.parts = 2;
        const uint32_t tl = tagged_lds_bytes(fp.wb, 0);
        const dim3 pgrid(fp.n_win, 2);
        if (fp.dense) hipLaunchKernelGGL((k_accum_pair<12, true>), pgrid, dim3(kAccThreads), tl, stream, aa);
        else hipLaunchKernelGGL((k_accum_pair<12, false>), pgrid, dim3(kAccThreads), tl, stream, aa);
    } else
#else
    (void)pair;
#endif
    if (uniq && tagged) {
        const uint32_t tl = tagged_lds_bytes(fp.wb, fp.n_shared);
        if (fp.dense && fp.wb == 12) FGFA_TAGGED_LAUNCH(agrid, tl, 12, true, (int)kTagSlots);
        else if (fp.dense && fp.wb == 13) FGFA_TAGGED_LAUNCH(agrid, tl, 13, true, (int)kTagSlots);
#ifdef FGFA_MEASURE
        else if (fp.wb == 11 && test_hook("FLATGFA_ACC_SMALL")) hipLaunchKernelGGL((k_accum_small<11>), agrid, dim3(kAccThreads), tl, stream, aa);
#endif
        else if (fp.wb == 11) FGFA_TAGGED_LAUNCH(agrid, tl, 11, false, (int)kTagSlots);
        else if (fp.acc_own && fp.wb == 12 && fp.acc_slots == 8 && fp.n_shared <= 64) FGFA_OWN_LAUNCH(agrid, tagged_lds_bytes(12, fp.n_shared, 8), 12, 8);
        else if (fp.acc_own && fp.wb == 12) FGFA_OWN_LAUNCH(agrid, tl, 12, (int)kTagSlots);
        else if (fp.acc_own && fp.wb == 13) FGFA_OWN_LAUNCH(agrid, tl, 13, (int)kTagSlots);
        else if (fp.wb == 12 && fp.acc_slots == 8 && fp.n_shared <= 64) FGFA_TAGGED_LAUNCH(agrid, tagged_lds_bytes(12, fp.n_shared, 8), 12, false, 8);
        else if (fp.wb == 12) FGFA_TAGGED_LAUNCH(agrid, tl, 12, false, (int)kTagSlots);
        else FGFA_TAGGED_LAUNCH(agrid, tl, 13, false, (int)kTagSlots);
    } else if (uniq) {
        // (k_scan_dense's records are single segments: the walk has nothing to park)
        const bool big = fp.big_groups;
        if (fp.dense && fp.wb == 12 && big) hipLaunchKernelGGL((k_accum<true, 12, false, true, true>), agrid, dim3(kAccThreads), 0, stream, aa);
        else if (fp.dense && fp.wb == 12) hipLaunchKernelGGL((k_accum<true, 12, false, true, false>), agrid, dim3(kAccThreads), 0, stream, aa);
        else if (fp.dense && fp.wb == 13) hipLaunchKernelGGL((k_accum<true, 13, false, true, true>), agrid, dim3(kAccThreads), 0, stream, aa);
        else if (fp.wb == 12 && big) hipLaunchKernelGGL((k_accum<true, 12, false, false, true>), agrid, dim3(kAccThreads), 0, stream, aa);
        else if (fp.wb == 13 && big) hipLaunchKernelGGL((k_accum<true, 13, false, false, true>), agrid, dim3(kAccThreads), 0, stream, aa);
        else if (fp.wb == 11) hipLaunchKernelGGL((k_accum<true, 11>), agrid, dim3(kAccThreads), 0, stream, aa);
        else if (fp.wb == 12) hipLaunchKernelGGL((k_accum<true, 12>), agrid, dim3(kAccThreads), 0, stream, aa);
        else hipLaunchKernelGGL((k_accum<true, 13>), agrid, dim3(kAccThreads), 0, stream, aa);
    } else {
        if (fp.wb == 11) hipLaunchKernelGGL((k_accum<false, 11>), agrid, dim3(kAccThreads), 0, stream, aa);
        else if (fp.wb == 12 && psum) hipLaunchKernelGGL((k_accum<false, 12, true>), agrid, dim3(kAccThreads), 0, stream, aa);
        else if (fp.wb == 12) hipLaunchKernelGGL((k_accum<false, 12>), agrid, dim3(kAccThreads), 0, stream, aa);
        else hipLaunchKernelGGL((k_accum<false, 13>), agrid, dim3(kAccThreads), 0, stream, aa);
    }
}
#undef FGFA_TAGGED_LAUNCH
#undef FGFA_OWN_LAUNCH

void launch_path_reduce(const FastPlan &fp, unsigned long long *len_out, unsigned long long *weighted_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_path_reduce, dim3((fp.n_items + 3) / 4), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(fp.items), fp.elist, fp.n_items,
                       fp.n_win, fp.dstride, reinterpret_cast<const ulonglong2 *>(fp.psum_part), len_out, weighted_out);
}

}  // namespace fgfa_dev
