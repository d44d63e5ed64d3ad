// Pass 1 over long items (seg_depth_with_uniq / seg_depth, ops/depth.rs:15-56): k_scan turns the steps of a path,
// or of a piece of a long path, into run records bucketed by segment window; k_scan_dense does the same for graphs
// whose ids have next to no runs (every step a record, partitioned by window in LDS).  See depth_fast.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "depth_fast_kernels.hpp"

#ifndef FGFA_TPROF_ALL
#define FGFA_TPROF_ALL 0  /* measurement builds: FLATGFA_SCAN_TIME's timeline from every build of k_scan, not the plain one alone */
#endif

namespace fgfa_dev {
namespace {

// ============================================================ pass 1, long items ===
//
// k_scan keeps no per-path state.  A wave's run queue holds (start id, position) pairs; the
// positions are block-relative (16 * lane + step), so a run's length is the distance to the next
// entry's position modulo 1024: a full block's last run is closed by the first entry of whatever
// the wave queues next (position 0), anything shorter appends a terminator entry (kInvalid,
// number of steps).  Entries leave the queue oldest first, 64 at a time, each as one record.

struct RWave {
    const uint32_t *poff;  // packed buckets: the workgroup's row of ScanArgs::pk_off, in LDS
    uint32_t dir;  // +1 / -1 (as unsigned): which way the current item's runs go (uniform; the queue never holds two items)
    uint32_t tagc; // what every record of the current item carries besides its range: 1 << 24, or the item's tag << kTagShift
    uint2 *q;
    uint32_t fill;
    uint32_t vm[3];
    int lane;
    bool epoch_ok;  // the item before the current one is complete: this wave may append records
    uint32_t tacc[8], tlast;  // kDbgTime (diagnostic): cycles per phase of this wave (scalar registers; a kernel is far shorter than 2^32 cycles)
};

// kDbgTime: charge the cycles since the last mark to phase `ph`
template <bool DBG>
__device__ __forceinline__ void tmark(const ScanArgs &A, RWave &w, int ph) {
    if (DBG && (A.dbg & kDbgTime)) {
        const uint32_t t = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_readcyclecounter());
        w.tacc[ph] = __builtin_amdgcn_readfirstlane(w.tacc[ph] + (t - w.tlast));
        w.tlast = t;
    }
}

// Pass A for four consecutive steps of every lane: Mj (a lane mask in an SGPR pair) = "step j
// starts a run" = its id is not the id before it plus DIR; CNT += Mj per lane.  Three vector
// instructions per step.  PM is the id before step 0.  DIR (an SGPR) is +1 or -1: the way the
// item's path mostly runs through the segment ids (a contig on the reverse strand walks them
// downwards; its runs are found just the same and emitted from their low end).
#define FGFA_PA_STEP(PMJ, XJ, MJ)                            \
    "v_add_u32 %[t], %[dir], %[" PMJ "]\n\t"                 \
    "v_cmp_ne_u32 %[" MJ "], %[" XJ "], %[t]\n\t"            \
    "v_addc_co_u32_e64 %[cnt], vcc, 0, %[cnt], %[" MJ "]\n\t"
#define FGFA_PA4(CNT, DIR, PM, X0, X1, X2, X3, M0, M1, M2, M3)                                                          \
    do {                                                                                                             \
        uint32_t t_;                                                                                                 \
        asm volatile(FGFA_PA_STEP("pm", "x0", "m0") FGFA_PA_STEP("x0", "x1", "m1") FGFA_PA_STEP("x1", "x2", "m2")    \
                         FGFA_PA_STEP("x2", "x3", "m3")                                                              \
                     : [cnt] "+v"(CNT), [t] "=&v"(t_), [m0] "=&s"(M0), [m1] "=&s"(M1), [m2] "=&s"(M2), [m3] "=&s"(M3) \
                     : [dir] "s"(DIR), [pm] "v"(PM), [x0] "v"(X0), [x1] "v"(X1), [x2] "v"(X2), [x3] "v"(X3)          \
                     : "vcc");                                                                                       \
    } while (0)

// Pass B for four consecutive steps of every lane: for step j, the lanes of ACT where a run
// starts (mask Mj) append (step j's id, step j's position) at their queue cursor `p`.  One scalar,
// one LDS and one vector instruction per step, no branches; exec is restored before the statement
// ends.  Pj holds the position of the lane's step j in the block for the whole kernel.
#define FGFA_PB_STEP(XJ, PJ, MJ)                                       \
    "s_and_b64 exec, %[act], %[" MJ "]\n\t"                            \
    "ds_write2_b32 %[p], %[" XJ "], %[" PJ "] offset1:1\n\t"           \
    "v_add_u32 %[p], 8, %[p]\n\t"
#define FGFA_PB4(P, ACT, X0, X1, X2, X3, P0, P1, P2, P3, M0, M1, M2, M3)                                             \
    do {                                                                                                             \
        unsigned long long sv_;                                                                                      \
        asm volatile("s_mov_b64 %[sv], exec\n\t" FGFA_PB_STEP("x0", "p0", "m0") FGFA_PB_STEP("x1", "p1", "m1")       \
                         FGFA_PB_STEP("x2", "p2", "m2") FGFA_PB_STEP("x3", "p3", "m3") "s_mov_b64 exec, %[sv]"       \
                     : [p] "+v"(P), [sv] "=&s"(sv_)                                                                  \
                     : [act] "s"(ACT), [x0] "v"(X0), [x1] "v"(X1), [x2] "v"(X2), [x3] "v"(X3), [p0] "v"(P0),         \
                       [p1] "v"(P1), [p2] "v"(P2), [p3] "v"(P3), [m0] "s"(M0), [m1] "s"(M1), [m2] "s"(M2),           \
                       [m3] "s"(M3)                                                                                  \
                     : "memory", "scc");                                                                             \
    } while (0)

__device__ __forceinline__ uint32_t epoch_now(uint32_t *ctl) {
    return __hip_atomic_load(ctl + kCtlEpoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Emit `n` queue entries starting at `base`, one per lane; entry base + n must exist (it closes
// the last run).  The whole run must lie below n_segs: that is the bounds check of every step in
// it.  A run that crosses into the next window (at most one: runs are shorter than a window) is
// emitted as two records.
// (the builds whose records' tags come from the blocks' ids, see k_scan's block_flag: builds of their own -- the flags' loads and the tag's select cost the others' k_scan 1-1.5 %)
template <int MODE>
constexpr bool kBlockFlags = mode_flags(MODE);
template <int MODE, int K>
__device__ __forceinline__ void emit_raw(const ScanArgs &A, RWave &w, uint32_t *bcur, uint32_t *mine, uint32_t base, uint32_t n) {
    constexpr bool DBG = MODE == kModeDbg;
    // K chunks of 64 entries side by side (n counts the entries of the last one; the others are
    // full): each chunk is a chain of LDS read, cursor atomic, permute and store, and the waves of
    // a CU are too few to hide one chain at a time when most steps start a run.
    const uint32_t wb = A.wb, wmask = (1u << wb) - 1u;
    const uint32_t down = w.dir == 1u ? 0u : ~0u;
    uint2 e[K], s[K];
    bool valid[K], cross[K];
    uint32_t win[K], rel[K], lenm1[K], pos[K], ncm[K];
    bool bad = false, any_cross = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        valid[k] = k + 1 < K || (uint32_t)w.lane < n;
        const uint32_t idx = base + 64u * (uint32_t)k + (valid[k] ? (uint32_t)w.lane : 0u);
        e[k] = w.q[idx];
        s[k] = w.q[idx + 1u];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        lenm1[k] = (s[k].y - e[k].y - 1u) & 1023u;
        valid[k] = valid[k] && e[k].x != kInvalid;
        uint32_t id = e[k].x - (lenm1[k] & down);  // a downward run is emitted from its low end
        if constexpr (kBlockFlags<MODE>) {
            ncm[k] = (uint32_t)((int32_t)id >> 31) & (kTagNoClaim << kTagShift);  // (bit 31 of a block's ids: its records need no claim -- all tag bits set)
            id &= 0x7FFFFFFFu;
        } else {
            ncm[k] = 0u;
        }
        if (mode_ranged(MODE)) {  // (uniform) the run's part inside this walk's range, if any; beyond the graph: the bounds check below
            const uint32_t hi = id + lenm1[k];
            const bool outside = hi >= A.n_total;
            const uint32_t lo2 = max(id, A.seg_base), hi2 = min(hi, A.seg_base + A.n_segs - 1u);
            const bool keep = lo2 <= hi2;
            valid[k] = valid[k] && (keep || outside);
            id = outside ? A.n_segs : lo2 - A.seg_base;
            lenm1[k] = outside ? 0u : hi2 - lo2;
        }
        const bool b = valid[k] && id + lenm1[k] >= A.n_segs;
        bad |= b;
        valid[k] = valid[k] && !b;
        win[k] = id >> wb;
        rel[k] = id & wmask;
        cross[k] = valid[k] && rel[k] + lenm1[k] > wmask;
        any_cross |= cross[k];
    }
    flag_if_any(A, bad, kStBounds);
#pragma unroll
    for (int k = 0; k < K; ++k) pos[k] = take_slots(bcur, w.lane, valid[k], win[k]);
    bool ovf = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t l1 = cross[k] ? wmask - rel[k] : lenm1[k];
        ovf |= put<DBG, mode_big(MODE), mode_packed(MODE)>(A, w, mine, valid[k], pos[k], win[k], rel[k] | (l1 << wb) | w.tagc | ncm[k]);
    }
    if (__builtin_amdgcn_ballot_w64(any_cross)) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t pos2 = cross[k] ? atomicAdd(&bcur[win[k] + 1u], 1u) : 0u;
            ovf |= put<DBG, mode_big(MODE), mode_packed(MODE)>(A, w, mine, cross[k], pos2, win[k] + 1u, ((lenm1[k] - (wmask - rel[k]) - 1u) << wb) | w.tagc | ncm[k]);
        }
    }
    if constexpr (mode_packed(MODE)) {
        // A packed call that runs out of room does so all the time -- the counting call, in which no sub-bucket has any: every
        // chunk of every wave would OR into the one status word (a million atomics on one address: 4.9 ms of a 0.3 ms kernel on
        // 200 M steps).  The workgroup says it once: a spare control word (behind kCtlEpoch, cleared with the others) remembers.
        if (__builtin_amdgcn_ballot_w64(ovf)) {
            uint32_t *said = bcur + 2u * A.nwp + kCtlEpoch + 2u;
            if (w.lane == 0 && __hip_atomic_load(said, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                __hip_atomic_store(said, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                atomicOr(A.status, kStOverflow);
            }
        }
    } else {
        flag_if_any(A, ovf, kStOverflow);
    }
}

// Emit the oldest entries, 64 at a time, while at least 65 are queued, then move what is left to
// the front of the queue.  With `all`, a terminator is appended and everything is emitted.
template <int MODE, int WIDE = 1>
__device__ __forceinline__ void drain_raw(const ScanArgs &A, RWave &w, uint32_t *bcur, uint32_t *mine, bool all, bool to_one = false) {  // (to_one: down to the one entry that closes the last run)
    if (all) {
        if (w.lane == 0) w.q[w.fill] = make_uint2(kInvalid, 0u);
        w.fill += 1u;
    }
    uint32_t base = 0;
    if (WIDE > 1) {
        while (w.fill - base >= 64u * WIDE + 1u) {
            emit_raw<MODE, WIDE>(A, w, bcur, mine, base, 64u);
            base += 64u * WIDE;
        }
    }
    while (w.fill - base >= 65u || ((all || to_one) && w.fill - base >= 2u)) {
        const uint32_t n = min(64u, w.fill - 1u - base);
        emit_raw<MODE, 1>(A, w, bcur, mine, base, n);
        base += n;
    }
    if (all) {
        w.fill = 0;
    } else if (base) {
        const uint32_t rem = w.fill - base;  // 1..64
        const bool mv = (uint32_t)w.lane < rem;
        const uint2 v = mv ? w.q[base + w.lane] : make_uint2(0u, 0u);
        if (mv) w.q[w.lane] = v;
        w.fill = rem;
    }
}

// Up to 64 consecutive steps, one per lane (what lies before an item's first 64-byte boundary
// and behind its last block).  Queued as a segment of its own, terminator included.
__device__ __forceinline__ void tile_narrow_raw(const ScanArgs &A, RWave &w, uint64_t t, uint32_t count) {
    const bool valid = (uint32_t)w.lane < count;
    const uint32_t id = valid ? A.steps[t + w.lane] >> 1 : 0u;
    const uint32_t prev = __builtin_amdgcn_update_dpp(0u, id, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const bool s = valid && (w.lane == 0 || id != prev + w.dir);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(s);
    if (s) w.q[w.fill + lane_rank(m)] = make_uint2(id, (uint32_t)w.lane);
    const uint32_t ns = (uint32_t)__builtin_popcountll(m);
    if (w.lane == 0) w.q[w.fill + ns] = make_uint2(kInvalid, count);
    w.fill += ns + 1u;
}

// How one work item (a path, or a piece of a long one) is cut up: the few steps [b, t0) before
// the first 64-byte boundary, then `nblk` blocks of 1024 steps starting at t0, the last of which
// may hold only `nl_last` lanes' worth of 16-step chunks, then fewer than 16 steps [tail, e).
// Blocks are not assigned to waves in advance: a wave takes the next free one from an LDS counter
// whenever one of its landing sets is free (its first three are its own index plus 0, 16 and 32).
struct Item {
    uint64_t b, e, t0, tail;
    uint32_t nblk, nl_last;
    uint32_t dir;      // +1, or -1 for an item marked as running down the segment ids (items[].z & 1)
    uint32_t shared;   // items[].z >> 1 (bits 1 .. 30): 0, or 1 + the ordinal of the split path this item is a piece of
    uint32_t noclaim;  // items[].z >> 31: the item's path never visits a segment twice (see kTagNoClaim)
    const uint4 *src;  // this lane's first 16 bytes of block 0
};

// Which item a workgroup takes in its r-th turn.  Items are sorted longest first and dealt out
// in snake order (0..G-1, then G-1..0, ...), which balances a sorted list well and needs no
// queue: a returning global atomic per item sat on the critical path of every path.
__device__ __forceinline__ uint32_t item_of(uint32_t round, uint32_t wg, uint32_t n_wg) {
    return round * n_wg + ((round & 1u) ? n_wg - 1u - wg : wg);
}

// An item's descriptor (and where pass 2 looks for it) is the same for all lanes: read through the
// scalar cache.  As a vector load hipcc waited for it with vmcnt(0) on the spot -- which also waits
// for every block the wave has in flight: a drained pipeline plus a round trip per item and wave
// (nothing next to a 100 k-step item; with 10 k-step ones k_scan is 3.5 % faster this way).  The lists
// are written before this kernel starts.
#ifndef FGFA_ITEM_RING
#define FGFA_ITEM_RING 1  /* 0: every wave reads its next item's descriptor from memory (measurements) */
#endif
#ifndef FGFA_ITEM_SLOAD
#define FGFA_ITEM_SLOAD 1  /* 1: where a workgroup has more than eight items; 0 / 2: never / always (measurements) */
#endif
__device__ __forceinline__ uint4 sload_item(const uint4 *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t sload_u32(const uint32_t *p) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

__device__ __forceinline__ Item make_item(const ScanArgs &A, bool have, uint4 d, int lane) {
    Item it;
    it.b = it.e = it.t0 = it.tail = 0;
    it.nblk = 0;
    it.nl_last = 64;
    it.dir = 1u;
    it.shared = 0u;
    it.noclaim = 0u;
    it.src = nullptr;
    if (have) {
        it.dir = (d.z & 1u) ? ~0u : 1u;
        it.shared = (d.z & ~kItemNoClaim) >> 1;
        it.noclaim = d.z >> 31;
        it.b = d.x;
        it.e = d.y;
        const uint64_t up = (it.b + 15) & ~(uint64_t)15;
        it.t0 = up < it.e ? up : it.e;
        uint64_t chunks = (it.e - it.t0) / 16;
        // a block is read as four whole KiB: a last, partial block that would reach past the step
        // array is left to the tail tiles instead
        if ((chunks % 64) && it.t0 + ((chunks + 63) / 64) * 1024 > A.n_steps) chunks -= chunks % 64;
        it.tail = it.t0 + chunks * 16;
        it.nblk = (uint32_t)((chunks + 63) / 64);
        it.nl_last = (chunks % 64) ? (uint32_t)(chunks % 64) : 64u;
        // kDbgHotLoads (diagnostic): every item reads the same cache-resident megabyte
        it.src = reinterpret_cast<const uint4 *>(A.steps + ((A.dbg & kDbgHotLoads) ? (it.t0 & 0x3FFF0u) : it.t0)) + lane;
    }
    return it;
}

// One block: 1024 consecutive steps, of which the first `nsteps` (a multiple of 16) count.  Lane l
// holds four groups of four consecutive steps: group k = steps 256k + 4l .. + 3 (a[4k .. 4k + 3]).
// Pass A marks the run starts and counts them per lane and group (a group's first step compares
// with the last step of the lane below; lane 0 always starts a run); two wave prefix sums (two
// 16-bit counts each) give every (group, lane) its own stretch of the queue, in path order; pass B
// appends.  If the queue cannot take the block's starts, or holds a chunk's worth and this wave may
// emit, the oldest entries are emitted first, one chunk at a time; a block that queues a lot is
// followed by a wide drain.
template <int MODE>
__device__ __forceinline__ void block16r(const ScanArgs &A, RWave &w, uint32_t *bcur, uint32_t *mine, uint32_t *ctl, uint32_t need,
                                         uint32_t (&a)[16], const uint32_t (&pj)[16], uint32_t nsteps) {
    constexpr bool DBG = MODE == kModeDbg;
    constexpr uint32_t kQ = kQueueOf<MODE>;
    unsigned long long m[16], act[4];
    uint32_t cnt[4];
    const bool partial = nsteps < 1024u;  // (wave-uniform) a partial block ends with a terminator
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t prev = __builtin_amdgcn_update_dpp(0u, a[4 * k + 3], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        cnt[k] = 0;
        FGFA_PA4(cnt[k], w.dir, prev, a[4 * k], a[4 * k + 1], a[4 * k + 2], a[4 * k + 3], m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]);
        const uint32_t add0 = 1u & ~(uint32_t)m[4 * k];  // lane 0's first step of the group starts a run whatever is below
        m[4 * k] |= 1ull;
        cnt[k] += (w.lane == 0) ? add0 : 0u;
        act[k] = ~0ull;
    }
    if (partial) {  // which (group, lane) pairs hold steps at all
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = 256u * (uint32_t)k + 4u * (uint32_t)w.lane < nsteps;
            act[k] = __builtin_amdgcn_ballot_w64(on);
            cnt[k] = on ? cnt[k] : 0u;
        }
    }
    const uint32_t s01 = wave_scan_incl(cnt[0] | (cnt[1] << 16)), s23 = wave_scan_incl(cnt[2] | (cnt[3] << 16));
    const uint32_t t01 = __builtin_amdgcn_readlane(s01, 63), t23 = __builtin_amdgcn_readlane(s23, 63);
    const uint32_t t0 = t01 & 0xFFFFu, t1 = t01 >> 16, t2 = t23 & 0xFFFFu, t3 = t23 >> 16;
    const uint32_t total = t0 + t1 + t2 + t3 + (partial ? 1u : 0u);
    tmark<DBG>(A, w, 2);
    if (FGFA_SKIP(kDbgNoEmit)) {
        w.fill = 0;
    } else if (w.fill >= 65u || w.fill + total + 2u > kQ) {
        if (!w.epoch_ok) {
            if (epoch_now(ctl) >= need) {
                w.epoch_ok = true;
            } else if (w.fill + total + 2u > kQ) {
                while (epoch_now(ctl) < need) __builtin_amdgcn_s_sleep(2);
                w.epoch_ok = true;
            }
            tmark<DBG>(A, w, 1);
        }
        if (w.epoch_ok) drain_raw<MODE>(A, w, bcur, mine, false);
        if (mode_packed(MODE) && w.fill + total + 2u > kQ) {  // (the shorter queue of a packed call; epoch_ok holds here)
            drain_raw<MODE>(A, w, bcur, mine, false, true);
            if (w.fill + total + 2u > kQ) {
                if (w.lane == 0) atomicOr(A.status, kStBackOverflow);
                return;
            }
        }
        tmark<DBG>(A, w, 3);
    }
    const uint32_t off[4] = {(s01 & 0xFFFFu) - cnt[0], t0 + (s01 >> 16) - cnt[1], t0 + t1 + (s23 & 0xFFFFu) - cnt[2],
                             t0 + t1 + t2 + (s23 >> 16) - cnt[3]};
    if (!FGFA_SKIP(kDbgNoPassB)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t p = lds_addr(w.q + w.fill + off[k]);
            FGFA_PB4(p, act[k], a[4 * k], a[4 * k + 1], a[4 * k + 2], a[4 * k + 3], pj[4 * k], pj[4 * k + 1], pj[4 * k + 2], pj[4 * k + 3],
                     m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]);
        }
    }
    if (partial && w.lane == 0) w.q[w.fill + total - 1u] = make_uint2(kInvalid, nsteps);
    w.fill += total;
    tmark<DBG>(A, w, 2);
    // A block that leaves kWide chunks' worth in the queue (paths whose runs are short) has them
    // emitted here, where the block's ids are dead and there are registers for kWide chunks side
    // by side.  (Draining only here was measured: no gain on such paths, and short items -- 32 k
    // steps -- lost 15 %: their waves more often find the item before them not wrapped up yet.)
    // (the diagnostic build keeps cycle counters in registers and has room for one chunk at a time only)
    if (mode_wide(MODE) > 1 && !FGFA_SKIP(kDbgNoEmit) && w.fill >= 64u * (DBG ? 1 : mode_wide(MODE)) + 1u) {
        if (!w.epoch_ok && epoch_now(ctl) >= need) w.epoch_ok = true;
        tmark<DBG>(A, w, 1);
        if (w.epoch_ok) drain_raw<MODE, (DBG ? 1 : mode_wide(MODE))>(A, w, bcur, mine, false);
        tmark<DBG>(A, w, 3);
    }
}

template <int MODE, bool TAGGED>
__global__ __launch_bounds__(kThreads) void k_scan(const ScanArgs A) {
    constexpr bool DBG = MODE == kModeDbg;
    constexpr uint32_t kRing = TAGGED ? kCtlRing - 1u : 1u;  // which cell of the control rings an item uses: its ordinal & kRing
    extern __shared__ uint32_t lds[];
    // layout: [bcur: nwp][snap: nwp, untagged only][control words][run queues: kWaves * kQ2 entries of 8 bytes]
    constexpr bool PACKED = mode_packed(MODE);
    static_assert(!PACKED || TAGGED, "packed buckets are for tagged calls");
    constexpr uint32_t kTables = TAGGED && !PACKED ? 1u : 2u;
    uint32_t *bcur = lds;
    uint32_t *snap = lds + A.nwp;  // the cursors when the current item started (not kept in a tagged call; a packed one keeps its sub-buckets' offsets there: n_win + 1 <= nwp words)
    uint32_t *ctl = lds + kTables * A.nwp;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the span math on the scalar unit
    uint32_t *mine = PACKED ? A.buckets + A.pk_base[blockIdx.x] : A.buckets + (size_t)blockIdx.x * A.cap;  // this workgroup's sub-bucket of window 0 (packed: its region)
    if (TAGGED && (mode_plain(MODE) || FGFA_TPROF_ALL) && A.tprof && threadIdx.x == 0) {
        A.tprof[kTprofRow * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        // where it runs: HW_ID (wave, SIMD, CU, shader array and engine) and XCC_ID
        A.tprof[kTprofRow * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
    RWave w;
    w.poff = snap;
    w.q = reinterpret_cast<uint2 *>(lds + kTables * A.nwp + kCtlWords) + (uint32_t)wave * kQueueOf<MODE>;
    w.fill = 0;
    w.vm[0] = w.vm[1] = w.vm[2] = 0;
    w.lane = lane;
    w.epoch_ok = true;
    for (int k = 0; k < 8; ++k) w.tacc[k] = 0;
    w.tlast = (DBG && (A.dbg & kDbgTime)) ? __builtin_amdgcn_readfirstlane((uint32_t)__builtin_readcyclecounter()) : 0u;
    if (A.zero_a) {  // small graphs: pass 2 adds to the outputs (AccArgs::parts)
        for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < A.n_segs; i += gridDim.x * kThreads) {
            A.zero_a[i] = 0u;
            if (A.zero_b) A.zero_b[i] = 0u;
        }
    }
    if (A.zero_c) {  // path depth: two memset launches less per call
        for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < A.n_zero64; i += gridDim.x * kThreads) A.zero_c[i] = A.zero_d[i] = 0ull;
    }
    // the cursors continue where k_scan_short (if it ran) left this workgroup's sub-buckets
    for (uint32_t i = threadIdx.x; i < A.nwp; i += kThreads) {
        const uint32_t c = i < A.n_win ? A.counts[(size_t)i * A.n_slots + blockIdx.x] : 0u;
        bcur[i] = c;
        if (!TAGGED) snap[i] = c;
        if (PACKED) snap[i] = i <= A.n_win ? A.pk_off[(size_t)blockIdx.x * (A.n_win + 1u) + i] : 0u;
        if (A.has_pre && i < A.n_win) A.counts0[(size_t)i * A.n_slots + blockIdx.x] = c;
    }
    if (threadIdx.x < kCtlWords)
        ctl[threadIdx.x] = threadIdx.x < kCtlArrive ? 4u * kWaves : threadIdx.x < kCtlJobs ? 0u : threadIdx.x == kCtlJobs + 1u ? gridDim.x + blockIdx.x : kJobEmpty;
    if (TAGGED && threadIdx.x >= kCtlDesc + 4u && threadIdx.x < kCtlDesc + 8u)  // the workgroup's second item is known from the start; its descriptor, if there is such an item (nobody looks otherwise)
        ctl[threadIdx.x] = gridDim.x + blockIdx.x < A.n_items + A.max_back ? reinterpret_cast<const uint32_t *>(A.items + gridDim.x + blockIdx.x)[threadIdx.x - (kCtlDesc + 4u)] : 0u;
    // block-relative positions of this lane's sixteen steps; opaque, so that they stay in registers
    uint32_t pj[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        pj[j] = 256u * (uint32_t)(j >> 2) + 4u * (uint32_t)lane + (uint32_t)(j & 3);
        asm volatile("" : "+v"(pj[j]));
    }
    __syncthreads();
    const uint32_t back = A.n_short ? min(*A.work_counter, A.max_back) : 0u;  // what k_scan_short handed back
    const uint32_t n_items = A.n_items + back;

    // The first blocks of an item are requested while the previous item is being wrapped up, and
    // its descriptor while the previous item is being walked.
    uint32_t rr = 0;  // this workgroup's items so far
    // (Few long items: reading the next item's descriptor from memory in every wave, which drains the
    // wave's loads once per item, measures 1-1.5 % FASTER on cfg-L's k_scan than the LDS ring or the
    // scalar read; with ten times as many items per workgroup those are 2-5 % faster.)
    const bool many_items = n_items > 8u * gridDim.x;
    // Which items a workgroup walks: untagged, the r-th is fixed (item_of: pass 2 finds it through the
    // directory anyway); tagged, the first two are (its own index, and that plus the number of
    // workgroups) and the rest come off a global counter, longest first, as the workgroups get to
    // them -- the slower ones take fewer.  A wave asks two items ahead, so that the answer and the
    // item's descriptor are there long before they are needed.
    uint32_t job = TAGGED ? blockIdx.x : item_of(0, blockIdx.x, gridDim.x);
    Item it = make_item(A, job < n_items, job < n_items ? A.items[job] : make_uint4(0u, 0u, 0u, 0u), lane);
    // where pass 2 looks for the item: fetched with its descriptor, long before it is needed
    uint32_t place = (TAGGED || job >= A.n_items) ? job | 0x80000000u : A.perm[job];
    // Records of item rr may be appended once `need` items of this workgroup are complete: the item
    // before it (its cursor snapshot is taken then), or -- tagged -- the one kTagSlots before it.
    uint32_t need = 0;
    w.dir = __builtin_amdgcn_readfirstlane(it.dir);
#define FGFA_ITEM_TAG() (TAGGED ? __builtin_amdgcn_readfirstlane((it.noclaim && !kBlockFlags<MODE>) ? kTagNoClaim : it.shared ? kTagCount - 1u - it.shared : rr) << kTagShift : 1u << 24)
    w.tagc = FGFA_ITEM_TAG();
    // Whether block j of the current item makes no-claim records: the item's path never meets a segment twice (items[].z bit 31),
    // or every 16-step chunk of the block lies in windows its path enters once and walks one way (ScanArgs::cflags: the bit of the
    // block's first chunk, set by the plan's k_block_flags).  The bit rides on top of the block's ids (take_block_flagged)
    // into the tag of every record they make (emit_raw).
    typedef __attribute__((address_space(4))) const uint32_t cu32;
    // (in two halves: the word is asked for when a landing set is given its next block, and looked at when the block is taken --
    // a scalar load's quarter of a microsecond would otherwise be waited for on the spot, once per block)
    const auto flag_fetch = [&](uint32_t j) -> uint32_t {
        uint32_t word = 0u;
        j = __builtin_amdgcn_readfirstlane(j);
        if (TAGGED && kBlockFlags<MODE> && A.cflags && j < (uint32_t)__builtin_amdgcn_readfirstlane(it.nblk)) {
            const uint32_t c0 = __builtin_amdgcn_readfirstlane((uint32_t)(it.t0 >> 4)) + 64u * j;
            word = *reinterpret_cast<cu32 *>(reinterpret_cast<uintptr_t>(A.cflags + (c0 >> 5))) >> (c0 & 31u);
        }
        return word;
    };
    const auto flag_eval = [&](uint32_t word) -> uint32_t { return __builtin_amdgcn_readfirstlane((it.noclaim | word) & 1u); };
    uint32_t fwb[3] = {0u, 0u, 0u};  // (uniform) bit 0: the mark of the block each landing set holds -- asked for with the block, looked at when it is taken
    uint32_t blk[3];  // the block each landing set holds (or will hold next)
    uint32_t resv;    // the block this wave takes after those
    // (a partial block is read whole: make_item has made sure that stays inside the step array)
#define FGFA_BLOCK_PTR(j) (it.src + (size_t)(j) * 256)
#define FGFA_LOAD_BLOCK(SET, j)                                                                   \
    do {                                                                                          \
        if (MODE != kModeDbg && it.t0 + (uint64_t)(j) * 1024u < A.mall_steps) load_block_coal_plain<SET>(w, FGFA_BLOCK_PTR(j)); \
        else load_block_coal<SET>(w, FGFA_BLOCK_PTR(j));                                          \
    } while (0)
    // (An item's first 64 blocks are its waves' own: wave w takes blocks j, j + 16, j + 32 and j + 48,
    // j = w less the blocks of the items before, mod 16 -- items of ten blocks would otherwise leave
    // the same six waves without work every time.)
    uint32_t rot = 0;
#define FGFA_PRELOAD()                                                      \
    do {                                                                    \
        blk[0] = ((uint32_t)wave - rot) & (kWaves - 1u);                    \
        blk[1] = blk[0] + kWaves;                                           \
        blk[2] = blk[0] + 2u * kWaves;                                      \
        resv = blk[0] + 3u * kWaves;                                        \
        if (blk[0] < it.nblk) FGFA_LOAD_BLOCK(0, blk[0]);                   \
        if (blk[1] < it.nblk) FGFA_LOAD_BLOCK(1, blk[1]);                   \
        if (blk[2] < it.nblk) FGFA_LOAD_BLOCK(2, blk[2]);                   \
        if (TAGGED && kBlockFlags<MODE>) fwb[0] = flag_fetch(blk[0]), fwb[1] = flag_fetch(blk[1]), fwb[2] = flag_fetch(blk[2]); \
    } while (0)
    // one block: wait for its data, take the next free block for its register set, process it
#define FGFA_BLOCK(SET)                                                                       \
    if (blk[SET] < it.nblk) {                                                                 \
        tmark<DBG>(A, w, 4);                                                                  \
        wait_block<SET>(w);                                                                   \
        tmark<DBG>(A, w, 0);                                                                  \
        uint32_t a[16];                                                                       \
        if constexpr (TAGGED && kBlockFlags<MODE>) take_block_flagged<SET>(a, flag_eval(fwb[SET])); \
        else take_block<SET>(a);                                                              \
        const uint32_t mine_now = blk[SET];                                                   \
        blk[SET] = resv;  /* taken one block ago, so that the LDS round trip is off this path */ \
        if (blk[SET] < it.nblk) FGFA_LOAD_BLOCK(SET, blk[SET]);                               \
        if (TAGGED && kBlockFlags<MODE>) fwb[SET] = flag_fetch(blk[SET]);                     \
        uint32_t got = 0;                                                                     \
        if (lane == 0) got = atomicAdd(&ctl[kCtlNext + (rr & kRing)], 1u);                    \
        if (!FGFA_SKIP(kDbgNoTiles)) {                                                        \
            block16r<MODE>(A, w, bcur, mine, ctl, need, a, pj, mine_now + 1 == it.nblk ? 16u * it.nl_last : 1024u); \
        } else if (a[0] == 0x3FFFFFFFu) {                                                     \
            atomicOr(A.status, kStDebug);                                                     \
        }                                                                                     \
        resv = __builtin_amdgcn_readfirstlane(got);                                           \
    }
    FGFA_PRELOAD();

    while (job < n_items) {
        uint32_t next_job;
        if (PACKED) {
            next_job = (rr + 1u) * gridDim.x + blockIdx.x;  // a fixed deal: the layout of the buckets was counted on it
        } else if (TAGGED) {
            uint32_t *ahead = &ctl[kCtlJobs + ((rr + 2u) & kRing)];
            uint32_t st = 0;
            if (lane == 0) st = atomicCAS(ahead, kJobEmpty, kJobPending);  // who fetches the item after the next?
            do {
                next_job = __hip_atomic_load(&ctl[kCtlJobs + ((rr + 1u) & kRing)], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } while (next_job >= kJobPending);  // (asked for an item ago: it is there, but for items of a handful of steps)
            if ((uint32_t)__builtin_amdgcn_readfirstlane(st) == kJobEmpty) {
                // (A workgroup has tag_limit private tags: its item of that ordinal would wrap into the split
                // paths' tags, or onto a bitset slot still in use.  It takes no further item then -- the
                // others do; should they all run out of tags, the last workgroup out reports it.)
                uint32_t got = kJobPending - 1u - 2u * gridDim.x;
                if (rr + 2u < A.tag_limit) {
                    if (lane == 0) got = atomicAdd(A.work_counter + 1, 1u);
                    got = __builtin_amdgcn_readfirstlane(got);
                }
                got = min(got + 2u * gridDim.x, kJobPending - 1u);
                // ... and its descriptor, for all the waves (each reading it from memory was a round trip per item and wave)
                if (FGFA_ITEM_RING && many_items && got < n_items && lane < 4) ctl[kCtlDesc + 4u * ((rr + 2u) & kRing) + lane] = reinterpret_cast<const uint32_t *>(A.items + got)[lane];
                if (lane == 0) __hip_atomic_store(ahead, got, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            next_job = item_of(rr + 1u, blockIdx.x, gridDim.x);
        }
        const uint32_t next_job_s = __builtin_amdgcn_readfirstlane(next_job);
        uint4 next_item = make_uint4(0u, 0u, 0u, 0u);
        uint32_t next_place = next_job_s | 0x80000000u;
        if (TAGGED && !PACKED && FGFA_ITEM_RING && many_items) {
            if (next_job_s < n_items) {
                const uint32_t *dc = ctl + kCtlDesc + 4u * ((rr + 1u) & kRing);
                next_item = make_uint4(dc[0], dc[1], dc[2], dc[3]);
            }
        } else if (!DBG && (FGFA_ITEM_SLOAD == 1 ? many_items : FGFA_ITEM_SLOAD != 0)) {  // (the diagnostic build has no registers to spare)
            if (next_job_s < n_items) next_item = sload_item(A.items + next_job_s);
            if (!TAGGED && next_job_s < A.n_items) next_place = sload_u32(A.perm + next_job_s);
        } else {
            if (next_job_s < n_items) next_item = A.items[next_job_s];
            if (!TAGGED && next_job_s < A.n_items) next_place = A.perm[next_job_s];
        }
        // The few steps outside the blocks are walked on their own, by two waves that change with
        // the item: their loads are plain ones, and the wait for them also waits for every block the
        // wave has in flight -- with items of ten blocks the same two waves paid that for every block
        // they took, and the others waited for them at the gate.
        const uint32_t head_wave = TAGGED ? rr & (kWaves - 1u) : 0u, tail_wave = TAGGED ? (rr + kWaves / 2u) & (kWaves - 1u) : kWaves - 1u;
        if ((uint32_t)wave == head_wave && it.t0 > it.b) tile_narrow_raw(A, w, it.b, (uint32_t)(it.t0 - it.b));
        if ((uint32_t)wave == tail_wave) {
            for (uint64_t t = it.tail; t < it.e; t += 64) {  // fewer than 16 steps, but for a block left out by make_item
                if (w.fill + 66u > kQueueOf<MODE>) {
                    while (epoch_now(ctl) < need) __builtin_amdgcn_s_sleep(2);
                    w.epoch_ok = true;
                    drain_raw<MODE>(A, w, bcur, mine, false);
                }
                tile_narrow_raw(A, w, t, (uint32_t)min((uint64_t)64, it.e - t));
            }
        }
#pragma unroll 1
        while (blk[0] < it.nblk || blk[1] < it.nblk || blk[2] < it.nblk) {
            FGFA_BLOCK(0)
            FGFA_BLOCK(1)
            FGFA_BLOCK(2)
        }
        // Records of this item may only be appended once every wave has left the item before it
        // (its cursor snapshot is taken then; tagged: the item kTagSlots before it); a wave that got
        // ahead has been queueing until now.
        // (In a tagged call a wave that has queued nothing -- the item had no block for it -- has
        // nothing to wait for: it goes on, up to kIdleAhead items ahead of the slowest.)
        tmark<DBG>(A, w, 4);
        if (TAGGED && w.fill == 0u) {
            while (rr >= kIdleAhead && epoch_now(ctl) + kIdleAhead <= rr) __builtin_amdgcn_s_sleep(2);
        } else {
            if (!w.epoch_ok) {
                while (epoch_now(ctl) < need) __builtin_amdgcn_s_sleep(2);
                w.epoch_ok = true;
            }
            tmark<DBG>(A, w, 1);
            drain_raw<MODE, (DBG ? 1 : mode_wide(MODE))>(A, w, bcur, mine, true);
        }
        tmark<DBG>(A, w, 3);
        // This wave is done with the item: it requests its first two blocks of the next one right away.
        const uint32_t pe = place;
        place = next_place;
        job = next_job;
        rot = (rot + it.nblk) & (kWaves - 1u);
        it = make_item(A, job < n_items, next_item, lane);
        w.dir = __builtin_amdgcn_readfirstlane(it.dir);  // the queue is empty here
        FGFA_PRELOAD();
        // The last wave to leave the item snapshots the cursors: dir[window][item] = the item's
        // stretch of this workgroup's sub-bucket, which is how pass 2 tells the paths apart --
        // in a tagged call the records say so themselves, and the last wave only counts the item off.
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&ctl[kCtlArrive + (rr & kRing)], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == kWaves - 1u) {
            if (!TAGGED) {
                // (pass 2 finds the item at its place in ITS walk order: one coalesced read per 64 items there)
                const uint32_t at = __builtin_amdgcn_readfirstlane(pe & 0x7FFFFFFFu);
                for (uint32_t i = lane; i < A.n_win; i += 64) {
                    const uint32_t c = bcur[i];
                    A.dir[(size_t)i * A.dstride + at] = make_uint2(snap[i], c);
                    snap[i] = c;
                }
                if (lane == 0) A.islot[at] = blockIdx.x | (pe & 0x80000000u);
            }
            if (lane == 0) {
                ctl[kCtlArrive + (rr & kRing)] = 0u;
                ctl[kCtlNext + (rr & kRing)] = 4u * kWaves;  // for the next item that uses these cells (no wave is there yet)
                if (TAGGED) ctl[kCtlJobs + (rr & kRing)] = kJobEmpty;
            }
            __hip_atomic_store(ctl + kCtlEpoch, rr + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // (items complete in order)
        }
        rr += 1u;
        w.tagc = FGFA_ITEM_TAG();
        need = TAGGED ? (rr >= kTagSlots ? rr - (kTagSlots - 1u) : 0u) : rr;
        w.epoch_ok = TAGGED && epoch_now(ctl) >= need;
        tmark<DBG>(A, w, 5);
    }
    if (DBG && (A.dbg & kDbgTime) && lane == 0) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(A.status + 8);
        for (int k = 0; k < 8; ++k) atomicAdd(&acc[k], (unsigned long long)w.tacc[k]);
    }
#undef FGFA_PRELOAD
#undef FGFA_BLOCK
#undef FGFA_BLOCK_PTR
#undef FGFA_LOAD_BLOCK
#undef FGFA_ITEM_TAG
    // publish how many records this workgroup left in each window's sub-bucket
    if (TAGGED && (mode_plain(MODE) || FGFA_TPROF_ALL) && A.tprof && lane == 0) A.tprof[kTprofRow * blockIdx.x + 4 + wave] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    for (uint32_t wdw = threadIdx.x; wdw < A.n_win; wdw += kThreads)
        A.counts[(size_t)wdw * A.n_slots + blockIdx.x] = bcur[wdw];
    if (TAGGED && threadIdx.x == 0) {  // the last workgroup out leaves the item counter clean for the next call
        A.taken[blockIdx.x] = rr;
        const uint32_t out = atomicAdd(A.work_counter + 2, 1u);
        if (out == gridDim.x - 1u) {
            // items nobody took (every workgroup out of tags): the call is completed through the atomic kernels
            const uint32_t taken = __hip_atomic_load(A.work_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!PACKED && (uint64_t)taken + 2ull * gridDim.x < n_items) atomicOr(A.status, kStBackOverflow);
            A.work_counter[1] = 0u;
            A.work_counter[2] = 0u;
        }
    }
    if (TAGGED && (mode_plain(MODE) || FGFA_TPROF_ALL) && A.tprof && threadIdx.x == 0) {
        A.tprof[kTprofRow * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        A.tprof[kTprofRow * blockIdx.x + 3] = rr;  // the items it took
    }
}

// ================================================== pass 1 for graphs without runs ===
//
// k_scan_dense: when nearly every step starts a run (ids that jump about: the plan counts
// more than three records for four steps), finding runs is wasted work and k_scan's emit -- 64
// records, 64 windows, 64 scattered 4-byte stores -- is bound by the L2s' request rate (0.48 ms
// for 100 M steps).  Here the workgroup partitions a tile of 8192 steps by window in LDS --
// one returning LDS atomic per step gives its rank within its window's bin, a prefix sum gives the
// bins their places -- and writes the sorted tile out, so that the records of a window leave as
// stretches of consecutive addresses.  Every step is a record of length one.
// The cursors, their snapshots per item and everything pass 2 reads are k_scan's.

// A barrier for LDS traffic only: __syncthreads() also waits for every global load in flight, and
// the next tile's steps are meant to stay in flight across the barriers of this tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// A tile goes through four phases -- P1 count (a returning LDS atomic per step), P2 the bins' prefix
// sum (one wave), P3 scatter into the stage, P4 write-out -- each needing the one before it finished
// by all waves.  Run one tile at a time that is five barriers a tile and the phases' times add up
// (per tile and wave, cycles: steps 600-1500, P1 1700, P2 1100 with fifteen waves idle, P3 1500, P4
// 1400-2700; FGFA_DENSE_PROF).  So two tiles are in flight, two phases between barriers: P1 of tile
// t + 1 beside P3 of tile t, then P2 of t + 1 (wave 0) beside P4 of t -- two barriers a tile, and
// LDS round trips of one phase behind the other's instructions.  `delta` and the tile's total are
// double-buffered (P2 of t + 1 writes them while P4 of t reads its own).
__global__ __launch_bounds__(kThreads) void k_scan_dense(const ScanArgs A) {
    extern __shared__ uint32_t lds[];
    uint32_t *bcur = lds, *snap = lds + A.nwp, *base = lds + 2u * A.nwp, *delta0 = lds + 3u * A.nwp, *hist = lds + 4u * A.nwp, *stage = lds + 5u * A.nwp + 64u;
    uint32_t *delta1 = stage + kDenseTile + 64u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (A.zero_a) {
        for (uint32_t i = blockIdx.x * kThreads + tid; i < A.n_segs; i += gridDim.x * kThreads) {
            A.zero_a[i] = 0u;
            if (A.zero_b) A.zero_b[i] = 0u;
        }
    }
    if (A.zero_c) {
        for (uint32_t i = blockIdx.x * kThreads + tid; i < A.n_zero64; i += gridDim.x * kThreads) A.zero_c[i] = A.zero_d[i] = 0ull;
    }
    for (uint32_t i = tid; i < A.nwp; i += kThreads) {
        const uint32_t c = i < A.n_win ? A.counts[(size_t)i * A.n_slots + blockIdx.x] : 0u;
        bcur[i] = c;
        snap[i] = c;
        hist[i] = 0u;
        if (A.has_pre && i < A.n_win) A.counts0[(size_t)i * A.n_slots + blockIdx.x] = c;
    }
    __syncthreads();
    const uint32_t back = A.n_short ? min(*A.work_counter, A.max_back) : 0u;
    const uint32_t n_items = A.n_items + back;
    const uint32_t wb = A.wb, wmask = (1u << wb) - 1u;
    uint32_t *mine = A.buckets + (size_t)blockIdx.x * A.cap;
    const bool small = ((uint64_t)A.n_win + 1u) * A.stride < (1ull << 32);  // a record's place in the bucket array fits 32 bits
    bool bad = false, ovf = false;
#ifdef FGFA_DENSE_PROF
    unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
#define DP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_readcyclecounter(); tp[i] += n_ - tl; tl = n_; } while (0)
#else
#define DP(i)
#endif
    for (uint32_t rr = 0;; ++rr) {
        const uint32_t job = item_of(rr, blockIdx.x, gridDim.x);
        if (job >= n_items) break;
        const uint4 d = A.items[job];
        const uint32_t place = job < A.n_items ? A.perm[job] : job | 0x80000000u;
        const uint32_t shared_ = (d.z & ~kItemNoClaim) >> 1;
        const uint32_t tagc = A.tagged ? ((d.z >> 31) ? kTagNoClaim : shared_ ? kTagCount - 1u - shared_ : rr) << kTagShift : 1u << 24;  // (see kTagShift)
        // The steps of a full tile land in v112..v119, two dwordx4 per thread (steps 4 tid .. 4 tid + 3 of either half),
        // requested a tile ahead: in C++ hipcc waits for a load as soon as its registers are copied into the next
        // iteration's, and its vmcnt(0) waits for the record stores of the tile before as well.  As in k_scan: the
        // registers are pinned (tools/check_pinned_vgprs.py), the wait is counted by hand -- eight stores at most have
        // been issued since -- and the ids are taken out by the shifts that drop the orientation bit.
        static_assert(kDensePer == 8 && kDenseTile == 8 * kThreads, "the landing registers hold eight steps per thread");
        const auto issue = [&](uint64_t t0) {
            const uint32_t *p0 = A.steps + t0 + 4u * tid, *p1 = p0 + kDenseTile / 2;
            asm volatile("global_load_dwordx4 v[112:115], %0, off nt\n\tglobal_load_dwordx4 v[116:119], %1, off nt" ::"v"(p0), "v"(p1)
                         : "memory", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
        };
        bool stores8 = false;  // the only vector-memory operations issued since the landing loads are eight record stores
        // P1: a step's (range-relative) id, or ~0 if it does not count, and its rank in its window's bin.
        // A full tile of a plan without ranges takes the plain form of every phase: no step of it is left out, so
        // nothing is predicated -- the general form costs 46 vector and 56 scalar instructions a step (a branch
        // around every atomic and every store), and a CU issues one scalar instruction per cycle for all its waves.
        uint32_t cur[kDensePer], lr[kDensePer], ncur[kDensePer], nlr[kDensePer];
        const auto is_plain = [&](uint64_t t0) { return !A.ranged && t0 + kDenseTile <= (uint64_t)d.y; };
        const auto count = [&](uint64_t t0) {
            if (is_plain(t0)) {
                if (stores8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("v_lshrrev_b32 %0, 1, v112\n\tv_lshrrev_b32 %1, 1, v113\n\tv_lshrrev_b32 %2, 1, v114\n\tv_lshrrev_b32 %3, 1, v115\n\t"
                             "v_lshrrev_b32 %4, 1, v116\n\tv_lshrrev_b32 %5, 1, v117\n\tv_lshrrev_b32 %6, 1, v118\n\tv_lshrrev_b32 %7, 1, v119"
                             : "=v"(ncur[0]), "=v"(ncur[1]), "=v"(ncur[2]), "=v"(ncur[3]), "=v"(ncur[4]), "=v"(ncur[5]), "=v"(ncur[6]), "=v"(ncur[7])
                             :
                             : "memory");
                if (is_plain(t0 + kDenseTile)) issue(t0 + kDenseTile);
                stores8 = false;
#pragma unroll
                for (int k = 0; k < kDensePer; ++k) {
                    bad |= ncur[k] >= A.n_segs;
                    ncur[k] = min(ncur[k], A.n_segs - 1u);  // (a bad id: the call fails, and until it does everything stays in bounds)
                    nlr[k] = atomicAdd(&hist[ncur[k] >> wb], 1u);
                }
                return;
            }
            const uint32_t cnt = (uint32_t)min((uint64_t)kDenseTile, (uint64_t)d.y - t0);
#pragma unroll
            for (int k = 0; k < kDensePer; ++k) {
                const uint32_t i = (uint32_t)k * kThreads + tid;
                bool valid = i < cnt;
                uint32_t id = (valid ? A.steps[t0 + i] : 0u) >> 1;
                if (A.ranged) {
                    bad |= valid && id >= A.n_total;
                    valid = valid && id - A.seg_base < A.n_segs;
                    id -= A.seg_base;
                } else {
                    bad |= valid && id >= A.n_segs;
                    valid = valid && id < A.n_segs;
                }
                ncur[k] = valid ? id : ~0u;
                nlr[k] = valid ? atomicAdd(&hist[id >> wb], 1u) : 0u;
            }
            if (is_plain(t0 + kDenseTile)) issue(t0 + kDenseTile);  // (cannot be: a tile that is not full is its item's last)
            stores8 = false;
        };
        // P2, by one wave: exclusive prefix sum of the bins, nwp / 64 consecutive bins per lane; the cursors move
        // on and the bins are empty again.  delta[bin] = what turns a place in the stage into the record's place in
        // the bucket array, counted from this workgroup's first sub-bucket (less than 2^30: fast_plan_create).
        const auto prefix = [&](uint32_t par) {
            if (tid < 64u) {
                uint32_t *delta = par ? delta1 : delta0;
                const uint32_t per = A.nwp >> 6;
                uint32_t sum = 0;
                for (uint32_t k = 0; k < per; ++k) sum += hist[lane * per + k];
                const uint32_t incl = wave_scan_incl(sum);
                uint32_t run = incl - sum;
                bool over = false;
                for (uint32_t k = 0; k < per; ++k) {
                    const uint32_t bin = lane * per + k;
                    const uint32_t hk = hist[bin], bk = bcur[bin];
                    base[bin] = run;
                    delta[bin] = bin * A.stride + bk - run;  // (modulo 2^32 in a plan with more records than that: P4 then takes the window's part off again)
                    bcur[bin] = bk + hk;
                    over |= bk + hk > A.cap;
                    hist[bin] = 0u;
                    run += hk;
                }
                const bool any_over = __builtin_amdgcn_ballot_w64(over) != 0ull;
                if (lane == 63u) {
                    stage[kDenseTile + 1u + par] = incl;  // the tile's records
                    stage[kDenseTile + 3u + par] = any_over ? 1u : 0u;  // ... and whether any of them is beyond its sub-bucket's end
                }
            }
        };
        if (is_plain(d.x)) issue(d.x);
        count(d.x);
        lds_barrier();
        prefix(0u);
        lds_barrier();
        uint32_t par = 0;
        for (uint64_t t0 = d.x; t0 < d.y; t0 += kDenseTile, par ^= 1u) {
            const bool more = t0 + kDenseTile < d.y;
            const bool plain = !A.ranged && t0 + kDenseTile <= d.y;
#pragma unroll
            for (int k = 0; k < kDensePer; ++k) cur[k] = ncur[k], lr[k] = nlr[k];
            DP(0);
            if (more) count(t0 + kDenseTile);
            DP(1);
            // P3 (branch free, so that a thread's lookups are in flight together: a step that does not count goes to a sink)
            if (plain) {
#pragma unroll
                for (int k = 0; k < kDensePer; ++k) stage[base[cur[k] >> wb] + lr[k]] = cur[k];  // (the id: its window and its place in the window)
            } else {
#pragma unroll
                for (int k = 0; k < kDensePer; ++k) {
                    const bool valid = cur[k] != ~0u;
                    const uint32_t at = base[valid ? cur[k] >> wb : 0u] + lr[k];
                    stage[valid ? at : kDenseTile] = cur[k];
                }
            }
            DP(2);
            lds_barrier();
            DP(3);
            if (more) prefix(par ^ 1u);
            DP(4);
            // P4.  The stage holds the tile sorted by window: consecutive places are consecutive slots of
            // a sub-bucket until the window changes.  (A wave per bin instead -- uniform addresses, no
            // bin lookup per record -- was measured 20 % slower: sixteen bins in a row, each waiting
            // for its own LDS reads.)
            const uint32_t *delta = par ? delta1 : delta0;
            const uint32_t total = stage[kDenseTile + 1u + par];
            const bool over = stage[kDenseTile + 3u + par] != 0u;
            uint32_t sid[kDensePer], dl[kDensePer];
#pragma unroll
            for (int k = 0; k < kDensePer; ++k) sid[k] = stage[(uint32_t)k * kThreads + tid];
            if (plain && !over && small) {
#pragma unroll
                for (int k = 0; k < kDensePer; ++k) dl[k] = delta[sid[k] >> wb];
#pragma unroll
                for (int k = 0; k < kDensePer; ++k) mine[dl[k] + ((uint32_t)k * kThreads + tid)] = (sid[k] & wmask) | tagc;
                stores8 = true;
            } else {
#pragma unroll
                for (int k = 0; k < kDensePer; ++k) {
                    const uint32_t j = (uint32_t)k * kThreads + tid;
                    sid[k] = j < total ? sid[k] : 0u;
                    dl[k] = delta[sid[k] >> wb];
                }
#pragma unroll
                for (int k = 0; k < kDensePer; ++k) {
                    const uint32_t j = (uint32_t)k * kThreads + tid;
                    const uint32_t wn = sid[k] >> wb, pos = dl[k] + j - wn * A.stride;
                    if (j < total) {
                        if (pos < A.cap) mine[(size_t)wn * A.stride + pos] = (sid[k] & wmask) | tagc;
                        else ovf = true;
                    }
                }
            }
            DP(5);
            lds_barrier();
            DP(6);
        }
        const uint32_t at = place & 0x7FFFFFFFu;
        for (uint32_t i = tid; i < A.n_win; i += kThreads) {
            const uint32_t c = bcur[i];
            A.dir[(size_t)i * A.dstride + at] = make_uint2(snap[i], c);
            snap[i] = c;
        }
        if (tid == 0) A.islot[at] = blockIdx.x | (place & 0x80000000u);
        __syncthreads();
    }
#ifdef FGFA_DENSE_PROF
    if ((blockIdx.x == 0 || blockIdx.x == 100) && (tid == 0 || tid == 1000)) printf("dense wg %u tid %u: between %llu count %llu scatter %llu barrierA %llu prefix %llu writeout %llu barrierB %llu\n", blockIdx.x, tid, tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6]);
#endif
    flag_if_any(A, bad, kStBounds);
    flag_if_any(A, ovf, kStOverflow);
    if (A.tagged && tid == 0) A.taken[blockIdx.x] = (n_items + gridDim.x - 1u) / gridDim.x;  // (item_of: no workgroup takes more)
    for (uint32_t i = tid; i < A.n_win; i += kThreads) A.counts[(size_t)i * A.n_slots + blockIdx.x] = bcur[i];
}

}  // namespace

bool scan_kernels_setup() {
    static OncePerDevice once;
    const bool ok = once([] {
        bool good = true;
        for (const void *k : {(const void *)k_scan<kModePlain, false>, (const void *)k_scan<kModePlain, true>, (const void *)k_scan<kModeRanged, false>,
                              (const void *)k_scan<kModeRanged, true>, (const void *)k_scan<kModeBig, true>, (const void *)k_scan<kModeRangedBig, true>,
                              (const void *)k_scan<kModePacked, true>, (const void *)k_scan<kModePackedRanged, true>,
                              (const void *)k_scan<kModePlainFlags, true>, (const void *)k_scan<kModePackedFlags, true>, (const void *)k_scan<kModePlainNarrow, true>,
#ifdef FGFA_MEASURE
                              (const void *)k_scan<kModeDbg, false>,
#endif
                              (const void *)k_scan_dense})
            good = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit) == hipSuccess && good;
        return good;
    });
    if (!ok) set_error("hipFuncSetAttribute(k_scan): dynamic shared memory");
    return ok;
}

int launch_scan(const FastPlan &fp, const ScanArgs &sa, bool tagged, uint32_t grid, hipStream_t stream) {
    const dim3 g(grid), b(kThreads);
    const uint32_t lds = fp.lds_bytes_scan;
    if (fp.dense) hipLaunchKernelGGL(k_scan_dense, g, b, dense_lds_bytes(fp.nwp), stream, sa);
#ifdef FGFA_MEASURE
    else if (fp.dbg) hipLaunchKernelGGL((k_scan<kModeDbg, false>), g, b, lds, stream, sa);
#endif
    else if (sa.cflags && tagged && !sa.ranged && fp.packed) hipLaunchKernelGGL((k_scan<kModePackedFlags, true>), g, b, lds, stream, sa);  // (a packed call's offsets are its region's: `big` is not its business)
    else if (sa.cflags && tagged && !sa.ranged && !sa.big && !fp.packed) hipLaunchKernelGGL((k_scan<kModePlainFlags, true>), g, b, lds, stream, sa);
    else if (fp.packed && sa.ranged) hipLaunchKernelGGL((k_scan<kModePackedRanged, true>), g, b, lds, stream, sa);
    else if (fp.packed) hipLaunchKernelGGL((k_scan<kModePacked, true>), g, b, lds, stream, sa);
    else if (sa.big && !tagged) { set_error("fast_seg_depth: a bucket array this large needs a tagged call"); return FLATGFA_ERR_ARG; }
    else if (sa.big && sa.ranged) hipLaunchKernelGGL((k_scan<kModeRangedBig, true>), g, b, lds, stream, sa);
    else if (sa.big) hipLaunchKernelGGL((k_scan<kModeBig, true>), g, b, lds, stream, sa);
    else if (sa.ranged && tagged) hipLaunchKernelGGL((k_scan<kModeRanged, true>), g, b, lds, stream, sa);
    else if (sa.ranged) hipLaunchKernelGGL((k_scan<kModeRanged, false>), g, b, lds, stream, sa);
    else if (tagged && fp.narrow_emit) hipLaunchKernelGGL((k_scan<kModePlainNarrow, true>), g, b, lds, stream, sa);
    else if (tagged) hipLaunchKernelGGL((k_scan<kModePlain, true>), g, b, lds, stream, sa);
    else hipLaunchKernelGGL((k_scan<kModePlain, false>), g, b, lds, stream, sa);
    return FLATGFA_OK;
}

}  // namespace fgfa_dev
