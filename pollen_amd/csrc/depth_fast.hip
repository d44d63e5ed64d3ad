// The bucketed node-depth path for gfx950 (seg_depth_with_uniq / seg_depth, ops/depth.rs:15-56):
// no global atomics on the data path.  DESIGN.md section 3 is the long version.
//
//   k_scan        (pass 1, stateless)  persistent workgroups walk one path (or one piece of a long
//                 path) at a time.  The steps are cut into blocks of 1024; a wave takes a block
//                 with four fully coalesced 1 KiB reads (streamed past the L2: nt), so that every
//                 lane holds four groups of four consecutive steps, finds where the maximal +1
//                 runs of segment ids start (-1 runs in an item that mostly walks the ids
//                 downwards: a contig on the reverse strand), and queues (start id, position) per run.
//                 A run's length is the distance to the next queued start, so each run becomes
//                 ONE range record (start, length) instead of `length` histogram updates.  A
//                 record goes to the bucket of its segment window; buckets are split into one
//                 private sub-bucket per workgroup, so the append cursor is an LDS counter and a
//                 workgroup's partial lines stay in its own XCD's L2.  The kernel keeps no
//                 per-path state: after the last wave has left an item it snapshots the cursors,
//                 which tells pass 2 which records of a sub-bucket belong to which path.
//                 A graph beyond 16 M segments is walked once per range of 16 M (k_scan<ranged>
//                 clips every run to the range).
//   k_scan_dense  (pass 1 for graphs with next to no runs)  every step is a record of length one;
//                 the workgroup partitions tiles of 8192 steps by window in LDS, so that a window's
//                 records leave as stretches of consecutive addresses instead of 64 scattered stores.
//   k_scan_short  (pass 1 for paths of at most 2048 steps, and "medium" paths with few runs)
//                 every wave walks whole paths on its own and claims the path's segments in a
//                 per-wave hash set of bitset words; its records carry what they count for.
//   k_accum       (pass 2)  one workgroup per window (several, adding their counts up, when the
//                 graph has fewer windows than CUs).  The "seen" bitset of depth.rs:23-34 lives
//                 here, per (path, window): 512 bytes of LDS instead of one bit per segment of the
//                 whole graph.  A wave walks the records of one path's group after the other (a
//                 path too long for one wave is walked by all sixteen on a shared bitset),
//                 claims each record's segments with returning LDS ORs (the bits that were already
//                 set are revisits), and applies the record as a +1/-1 pair to an LDS difference
//                 array for depth -- and its revisited stretches to a second one; uniq = depth -
//                 revisits -- which are prefix-summed and written with 16-byte stores.  For path
//                 depth the same kernel, once the window's depth is final, turns every record
//                 into two differences of window-local prefix sums (sum len, sum depth * len).
//
// Exactness: every step lies in exactly one run, so it contributes +1 to exactly one depth
// record; every (path, segment) pair that occurs sets exactly one bit of its path's bitset: the
// lane whose OR found it clear met a first visit, every other a revisit.  Sums of +1s are order-independent, hence the
// results equal depth.rs bit for bit under any scheduling.  Sub-buckets have a fixed capacity; a
// record that does not fit raises a flag, and flatgfa_dev_status completes the call on a larger
// plan (or through the atomic kernels) before it reports success.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <queue>
#include <cstdlib>
#include <cstring>
#include <cstdio>
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
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kMaxSlots = 1024;  // sub-buckets per window (= workgroups of pass 1) k_accum can stage
constexpr int kAccThreads = 1024;
constexpr uint32_t kAccWaves = kAccThreads / 64;
constexpr uint32_t kLdsLimit = 160 * 1024;
// status word bits (flatgfa_dev_status)
constexpr uint32_t kStBounds = 1u, kStDebug = 2u, kStOverflow = 4u, kStInternal = 8u, kStBackOverflow = 16u;  // (16: more short paths handed back than the list holds: larger buckets would not help, the atomic kernels complete the call)  // (8: an invariant between the two passes did not hold -- a bug, reported as an error rather than as counts)

// ---- the wave-per-path kernels (k_scan_short): windows of 4096 segments, at most 256 of them ----
constexpr uint32_t kRunBits = 11;  // a run of a depth-only call is cut at multiples of 2^11 ids (it then never crosses a window's end)
constexpr uint32_t kRunSpan = 1u << kRunBits;
constexpr uint32_t kPosBits = 10;                        // a queued entry is (id << 10) | the step's position in its block: a run STARTS there
constexpr uint32_t kTermEntry = 0xFFFFFFFFu << kPosBits;  // ... or, with this id, ends there: what closes a block's last run (never emitted)
// Where their runs are cut: depth-only runs at ids that are multiples of 2048 (a record never
// crosses a window); with unique depth at multiples of 32, so that a run lies inside ONE word of
// the "seen" bitset and is claimed with a single returning LDS OR.
template <bool UNIQ>
constexpr uint32_t kCutMask = UNIQ ? 31u : kRunSpan - 1u;
constexpr uint32_t kShortWinBits = 12;
constexpr uint32_t kShortMaxWin = 256;     // LDS cursor table entries of k_scan_short
constexpr uint32_t kShortMaxSegs = 1u << 20;
constexpr uint32_t kShortMax = 2048;       // steps; longer paths go through k_scan (or the medium variant)
constexpr uint32_t kDummyBase = 1u << 20;  // ids from here up stand in for steps outside the path (never emitted)
constexpr int kShortHash = 9;              // per-wave hash set of 512 (bitset word index + 1, bits) pairs
// The medium-path variant: hash sets of 2048 entries, for paths whose run count (known to the plan) fits
// one -- seven sets and fourteen waves per workgroup, two waves per path (eight and eight, one wave per
// path, when built with -DFGFA_MEDIUM_PAIRED=0).
#ifndef FGFA_MEDIUM_PAIRED
#define FGFA_MEDIUM_PAIRED 1
#endif
constexpr bool kMediumPaired = FGFA_MEDIUM_PAIRED != 0;  // two waves per path and hash set (k_scan_short<..., PAIRED>)
constexpr int kMediumHash = 11, kMediumWaves = kMediumPaired ? 14 : 8;
constexpr uint32_t kQPaired = 328;  // a paired wave's run queue: 64 left over + up to 256 from sixteen lanes + the entry that closes a block
#ifndef FGFA_SHORT_WAVES
#define FGFA_SHORT_WAVES 16
#endif
constexpr int kShortWaves = FGFA_SHORT_WAVES;  // waves of a k_scan_short workgroup (measurements: fewer = lower occupancy)
constexpr uint32_t kMediumRuns = 1500;
constexpr uint32_t kMaxHandBack = 4096;    // short paths k_scan_short may hand back to k_scan per call

// ---- tagged records ----
// A record of k_scan is (window-relative first segment) | (length - 1) << wb.  In a *tagged* call it
// also says whose it is, from bit 23 up: pass 2 then needs no directory of which records of a
// sub-bucket belong to which item, and k_scan no cursor snapshot (hence no turnaround) per item.
//   private tag  = the item's ordinal among its workgroup's items (a whole path): its records lie in
//                  ONE sub-bucket per window, and its "seen" bitset is one of the kTagSlots private
//                  bitsets of the pass-2 wave that walks that sub-bucket, slot = tag mod kTagSlots.
//                  k_scan guarantees that all records of tag t precede all records of tag T in a
//                  sub-bucket whenever T - t >= kTagSlots (a wave may only emit for item rr once
//                  every wave has left item rr - kTagSlots), so a slot is free when its next owner
//                  shows up;
//   shared tag   = kTagCount - 1 - (ordinal of a path that is cut into pieces): the pieces are walked
//                  by different workgroups, their records lie in many sub-buckets, and all waves of
//                  pass 2 claim in ONE bitset per such path (LDS ORs are atomic across waves).
constexpr uint32_t kTagShift = 23, kTagCount = 1u << (32 - kTagShift);
constexpr uint32_t kTagSlots = 4;
constexpr uint32_t kMaxShared = 128;      // bitsets of split paths pass 2 has LDS for (4096-segment windows)

// ---- k_scan ----
constexpr uint32_t kMaxWin = 2048;        // windows per launch (LDS tables: the cursors and their snapshots per item)
constexpr uint32_t kMaxWinTagged = 4096;  // ... of a plan whose calls are always tagged: k_scan keeps no snapshots then
constexpr uint32_t kInvalid = 0xFFFFFFFFu;  // queue entry that starts no run (terminates the one before it)
constexpr uint32_t kQ2 = 64 + 1024 + 8;   // queue entries per wave: what is left over + one all-starts block

// diagnostic ablations (FLATGFA_DEBUG_SKIP, results are then wrong by construction)
constexpr uint32_t kDbgNoStore = 1, kDbgNoTiles = 8, kDbgHotLoads = 16, kDbgTime = 32, kDbgNoEmit = 2, kDbgNoPassB = 4, kDbgNoClaim = 64, kDbgNoRevisit = 128, kDbgNoDepth = 256, kDbgHotStores = 512;
// the ablation checks exist only in the DBG instantiation of the kernel
#define FGFA_SKIP(bit) (DBG && (A.dbg & (bit)))

struct ScanArgs {
    const uint32_t *path_begin, *path_end;  // the graph's spans: a handed-back path that was read from its reversed copy is walked by k_scan from the graph's own steps
    const uint32_t *rev_steps;  // the wave-per-path kernels: the reversed copies, which the paths from n_fwd on in the list are read from (one launch for both)
    uint32_t n_fwd;
    uint32_t seg_base, n_total, ranged;  // ranged: this walk keeps what falls into [seg_base, seg_base + n_segs) of the graph's n_total segments
    uint32_t *zero_a, *zero_b;  // k_scan clears these vectors of n_segs counts first (pass 2 adds to them when windows are shared); or null
    unsigned long long *zero_c, *zero_d;  // ... and these two of n_zero64 sums (the paths' sums k_path_reduce adds to); or null
    uint32_t n_zero64;
    const uint32_t *steps;
    uint4 *items;        // work items, longest first: {begin, end, -, path}; room behind the first n_items
                         // for the short paths k_scan_short hands back (counted in *work_counter)
    const uint4 *short_items;  // paths of at most kShortMax steps, longest first
    uint32_t n_short;
    uint64_t n_steps;
    uint32_t n_items, n_segs, n_win, n_slots;
    uint32_t wb;         // log2 of the window size (k_scan; k_scan_short always uses 12)
    uint32_t nwp;        // n_win rounded up to a multiple of 64 (LDS table size)
    uint32_t has_pre;    // k_scan_short ran before: keep its cursors for pass 2
    uint32_t max_back;   // items k_scan_short may hand back
    uint32_t *work_counter;
    uint32_t *counts;    // [n_win][n_slots] cursors: what k_scan_short left, then what k_scan left
    uint32_t *counts0;   // [n_win][n_slots] copy of the cursors k_scan started from
    uint32_t *buckets;   // [n_win + 1][n_slots][cap]; window n_win is a write sink
    uint2 *dir;          // [n_win][dstride] {cursor before, cursor after} item j in its workgroup's sub-bucket
    uint32_t *islot;     // [dstride] the sub-bucket (workgroup) that walked the item at each position of pass 2's walk order | first of its path << 31
    const uint32_t *perm;  // [n_items] item j's position in that order | first of its path << 31 (handed-back items keep their index)
    uint32_t dstride;
    uint32_t cap;
    uint32_t stride;     // n_slots * cap: elements between consecutive windows (< 2^24)
    uint32_t sink;       // n_win * stride (meaningful while the bucket array holds fewer than 2^30 records)
    uint32_t big;        // the bucket array holds 2^30 records or more: 64-bit offsets in put() (k_scan<kModeBig / kModeRangedBig>; k_scan_dense reads this)
    uint32_t *status;
    uint32_t dbg;
    uint32_t tagged;     // records carry their item's tag (see kTagShift); k_scan_dense reads this, k_scan is a build of its own
    // Packed buckets (k_scan<kModePacked...>): every (window, workgroup) sub-bucket has exactly the room its records
    // need -- counted once, with the items dealt in a fixed order -- and a workgroup's sub-buckets lie back to back:
    const uint32_t *pk_off;   // [n_slots][n_win + 1] where each of the workgroup's sub-buckets starts in its region (the last entry: the region's end, a sink)
    const uint64_t *pk_base;  // [n_slots] where the workgroup's region starts in `buckets`
    uint64_t mall_steps; // blocks that start below this step index are read without the nt hint, so that they stay in the Infinity Cache from one call to the next (FastPlan::mall_steps)
    uint32_t *taken;     // tagged: [n_slots] how many items each workgroup took (its private tags are 0 .. taken - 1): pass 2 clears all of a wave's
                         // bitsets at once where a sub-bucket has no more tags than the wave has bitsets, and none changes hands inside it
    uint32_t tag_limit;  // tagged: how many items a workgroup may take (its private tags are 0 .. tag_limit - 1; the split paths' lie above)
    unsigned long long *tprof;  // FLATGFA_SCAN_TIME (diagnostic): per workgroup, when it started, when it ended, when each of its waves ran out of work (10 ns units)
};

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

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

// LDS byte address of a pointer into the workgroup's shared memory
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(lds_u32 *)p; }

// Store a record at slot `pos` of this workgroup's sub-bucket of window `win`.  Branch free:
// lanes with nothing to store (or no room) write to the sink window.  Returns whether the record
// did not fit (the call is then completed on a larger plan, see flatgfa_dev_status).
template <bool DBG, bool BIG = false, bool PACKED = false, typename W>
__device__ __forceinline__ bool put(const ScanArgs &A, W &w, uint32_t *mine, bool e, uint32_t pos, uint32_t win, uint32_t rec) {
    if constexpr (PACKED) {
        // the sub-bucket's start and end in the workgroup's region (two neighbouring LDS words); what does not fit
        // (a call that makes other records than the one the layout was counted on) goes to the region's sink
        const uint32_t wq = e ? win : 0u;
        const uint32_t lo = w.poff[wq], hi = w.poff[wq + 1u];
        const bool fits = e && pos < hi - lo;
        mine[fits ? lo + pos : w.poff[A.n_win]] = rec;
        w.vm[0] += 1;
        w.vm[1] += 1;
        w.vm[2] += 1;
        return e && !fits;
    }
    const bool ok = e && pos < A.cap;
    if (BIG) {
        // A bucket array of 2^30 records or more (many windows times sub-buckets deep enough for the few
        // workgroups that walk a window's paths, as on a whole-genome graph whose paths run along it):
        // the offset takes 64 bits -- one quarter-rate multiply-add per chunk of 64 records.
        const unsigned long long slot = (unsigned long long)(ok ? win : A.n_win) * A.stride + (ok ? pos : 0u);
        mine[slot] = rec;
        w.vm[0] += 1;  // exactly one store instruction, executed by the whole wave
        w.vm[1] += 1;
        w.vm[2] += 1;
        return e && !ok;
    }
    // The bucket array holds fewer than 2^30 records, so a 32-bit byte offset from a uniform base
    // suffices.  window * stride + pos as one full-rate 24-bit multiply-add (the plan keeps the
    // stride below 2^24; hipcc would otherwise pick the quarter-rate 64-bit mad).
    uint32_t slot;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(slot) : "v"(win), "s"(A.stride), "v"(pos));
    uint32_t boff = (ok ? slot : A.sink) << 2;
    if (FGFA_SKIP(kDbgHotStores)) boff = (pos & 1023u) << 2;  // diagnostic: the same instructions, but the lines stay in L2
    if (!FGFA_SKIP(kDbgNoStore)) {
#ifdef FGFA_NT_STORE
        __builtin_nontemporal_store(rec, reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(mine) + boff));
#else
        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(mine) + boff) = rec;
#endif
        w.vm[0] += 1;  // exactly one store instruction, executed by the whole wave
        w.vm[1] += 1;
        w.vm[2] += 1;
    }
    return e && !ok;
}

__device__ __forceinline__ void flag_if_any(const ScanArgs &A, bool b, uint32_t bit) {
    if (__builtin_amdgcn_ballot_w64(b) && b) atomicOr(A.status, bit);
}

// ----------------------------------------------- streaming loads of steps ---
//
// Two blocks per wave (8 KiB; 128 KiB per CU) are kept in flight across loop iterations.  hipcc
// cannot express that: it drains vmcnt to 0 at the top of the loop, and an inline-asm load into a
// compiler-allocated register is unsafe because the compiler may copy the register (to rotate it
// through the loop) while the load is still in flight.  So the landing registers are fixed sets
// of sixteen -- v[96:111] and v[112:127], and v[80:95] in k_scan, which keeps three blocks in
// flight -- which the compiler is told are clobbered and never otherwise allocates (the kernels
// need < 80 VGPRs; 128 is the budget of a 1024-thread workgroup).  tools/check_pinned_vgprs.py checks the generated ISA for exactly that (`make
// check`, and the CPU test suite).
// A block is taken out of its set, already shifted down to segment ids, by v_lshrrevs issued
// after a counted s_waitcnt (wait_block).  On gfx950 vmcnt counts loads and stores alike and they
// return in issue order (hipcc itself relies on that: it waits vmcnt(2) for a load followed by
// two stores), so the wait counts the record stores issued since, too -- otherwise every block
// would wait for the stores of the block before it to be acknowledged.
// k_scan_short's pattern (load_block_async): the four loads of a lane cover its own 64 bytes; the
// wave's four instructions together cover 4 KiB, every 64-byte sector exactly once per instruction (measured at the same 5.9 TB/s as
// fully coalesced loads, tools/loadpat.hip).  No nontemporal hint here: the sectors must survive
// in cache from the first of the four instructions to the last.
#ifndef FGFA_LOAD_POLICY
#define FGFA_LOAD_POLICY ""  /* cache-policy bits of k_scan_short's step loads: they need their lines to survive from the first of a lane's four loads to the last */
#endif
#ifndef FGFA_COAL_POLICY
#define FGFA_COAL_POLICY " nt"  /* k_scan's step loads are whole lines read once: streamed past the L2, whose lines are left to the records (measured: k_scan 116 -> 104 us; " sc1" / " sc0 sc1": no change) */
#endif
#define FGFA_CLOB_A "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"
#define FGFA_CLOB_C "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"
#define FGFA_CLOB_B "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
template <int SET, typename W>
__device__ __forceinline__ void load_block_async(W &w, const uint4 *p) {
#pragma unroll
    for (int k = 0; k < 3; ++k) w.vm[k] = k == SET ? 0u : w.vm[k] + 4u;
    if (SET == 0)
        asm volatile("global_load_dwordx4 v[96:99], %0, off" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[100:103], %0, off offset:16" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[104:107], %0, off offset:32" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[108:111], %0, off offset:48" FGFA_LOAD_POLICY "" ::"v"(p) : "memory", FGFA_CLOB_A);
    else if (SET == 1)
        asm volatile("global_load_dwordx4 v[112:115], %0, off" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[116:119], %0, off offset:16" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[120:123], %0, off offset:32" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[124:127], %0, off offset:48" FGFA_LOAD_POLICY "" ::"v"(p) : "memory", FGFA_CLOB_B);
    else
        asm volatile("global_load_dwordx4 v[80:83], %0, off" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[84:87], %0, off offset:16" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[88:91], %0, off offset:32" FGFA_LOAD_POLICY "\n\t"
                     "global_load_dwordx4 v[92:95], %0, off offset:48" FGFA_LOAD_POLICY "" ::"v"(p) : "memory", FGFA_CLOB_C);
}
// k_scan's pattern: instruction k of a block reads the block's k-th KiB, 16 bytes per lane -- each
// instruction is one fully coalesced 1 KiB read -- so that lane l ends up with four groups of four
// consecutive steps: steps 256k + 4l .. 256k + 4l + 3 of the block in registers 4k .. 4k + 3.
// (the same reads without the nt hint: they allocate in the L2 and the Infinity Cache -- ScanArgs::mall_steps)
template <int SET, typename W>
__device__ __forceinline__ void load_block_coal_plain(W &w, const uint4 *p) {
#pragma unroll
    for (int k = 0; k < 3; ++k) w.vm[k] = k == SET ? 0u : w.vm[k] + 4u;
    if (SET == 0)
        asm volatile("global_load_dwordx4 v[96:99], %0, off\n\t"
                     "global_load_dwordx4 v[100:103], %0, off offset:1024\n\t"
                     "global_load_dwordx4 v[104:107], %0, off offset:2048\n\t"
                     "global_load_dwordx4 v[108:111], %0, off offset:3072" ::"v"(p) : "memory", FGFA_CLOB_A);
    else if (SET == 1)
        asm volatile("global_load_dwordx4 v[112:115], %0, off\n\t"
                     "global_load_dwordx4 v[116:119], %0, off offset:1024\n\t"
                     "global_load_dwordx4 v[120:123], %0, off offset:2048\n\t"
                     "global_load_dwordx4 v[124:127], %0, off offset:3072" ::"v"(p) : "memory", FGFA_CLOB_B);
    else
        asm volatile("global_load_dwordx4 v[80:83], %0, off\n\t"
                     "global_load_dwordx4 v[84:87], %0, off offset:1024\n\t"
                     "global_load_dwordx4 v[88:91], %0, off offset:2048\n\t"
                     "global_load_dwordx4 v[92:95], %0, off offset:3072" ::"v"(p) : "memory", FGFA_CLOB_C);
}
template <int SET, typename W>
__device__ __forceinline__ void load_block_coal(W &w, const uint4 *p) {
#pragma unroll
    for (int k = 0; k < 3; ++k) w.vm[k] = k == SET ? 0u : w.vm[k] + 4u;
    if (SET == 0)
        asm volatile("global_load_dwordx4 v[96:99], %0, off" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[100:103], %0, off offset:1024" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[104:107], %0, off offset:2048" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[108:111], %0, off offset:3072" FGFA_COAL_POLICY "" ::"v"(p) : "memory", FGFA_CLOB_A);
    else if (SET == 1)
        asm volatile("global_load_dwordx4 v[112:115], %0, off" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[116:119], %0, off offset:1024" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[120:123], %0, off offset:2048" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[124:127], %0, off offset:3072" FGFA_COAL_POLICY "" ::"v"(p) : "memory", FGFA_CLOB_B);
    else
        asm volatile("global_load_dwordx4 v[80:83], %0, off" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[84:87], %0, off offset:1024" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[88:91], %0, off offset:2048" FGFA_COAL_POLICY "\n\t"
                     "global_load_dwordx4 v[92:95], %0, off offset:3072" FGFA_COAL_POLICY "" ::"v"(p) : "memory", FGFA_CLOB_C);
}
// Waits until the loads into landing set SET have returned.  `w.vm[SET]` counts the memory
// instructions this wave is known to have issued since (the other set's loads and the record
// stores); they return in issue order, so the loads are back once at most that many operations
// are outstanding.  Rounded down to one of a few immediates; anything issued but not counted
// (rare paths) only makes the wait stricter.
template <int SET, typename W>
__device__ __forceinline__ void wait_block(const W &w) {
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
    else if (SET == 2) FGFA_TAKE16("v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
    else FGFA_TAKE16("v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

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

#ifndef FGFA_QCAP
#define FGFA_QCAP 416
#endif
#ifndef FGFA_SHORT_ABLATE
#define FGFA_SHORT_ABLATE 0  /* measurements only (results are wrong): 1 loads only, 2 runs queued but not emitted, 4 no claims, 8 no record stores, 16 hash set not wiped, 32 partly new claims dropped */
#endif
constexpr uint32_t kQCap = FGFA_QCAP;  // at least 64 left over + up to 257 from sixteen lanes of a block; a short path has at most kQCap - 16 runs, hence bitset words: its 512-entry hash set must not fill up
constexpr uint32_t kPCap = 96;   // parked claims (two words each): 31 left over + up to 64 from one chunk

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

// Slots in the sub-buckets of `win`, one per valid lane.  A path that runs along the graph puts
// neighbouring lanes' runs in the same window, and 64 LDS atomics on one address take 64 turns:
// the first lane of every stretch of equal windows takes the slots of its stretch.  Branch free,
// so that the chunks of a wide drain overlap their LDS round trips.  (FGFA_SLOTS_MODE, measured on
// one box: 1 adds a wave-uniform shortcut for "all lanes one window" and one atomic per lane
// when most lanes differ -- fewer instructions, but a branch between the chunks: +4 % on paths
// along the graph; 2 only the latter: +7 %; cfg-L is the same with all three.)
#ifndef FGFA_SLOTS_MODE
#define FGFA_SLOTS_MODE 0
#endif
__device__ __forceinline__ uint32_t take_slots(uint32_t *bcur, int lane, bool valid, uint32_t win) {
#if FGFA_SLOTS_MODE == 1
    const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
    if (vm == 0) return 0u;
    const uint32_t win0 = __builtin_amdgcn_readlane(win, (int)__builtin_ctzll(vm));
    if (__builtin_amdgcn_ballot_w64(valid && win != win0) == 0) {
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(&bcur[win0], (uint32_t)__builtin_popcountll(vm));
        return __builtin_amdgcn_readfirstlane(first) + lane_rank(vm);
    }
#endif
    const uint32_t key = valid ? win : 0x80000000u | (uint32_t)lane;
    const uint32_t kprev = __builtin_amdgcn_update_dpp(~0u, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const unsigned long long heads = __builtin_amdgcn_ballot_w64(lane == 0 || key != kprev);
#if FGFA_SLOTS_MODE >= 1
    if (__builtin_popcountll(heads) > 48) return valid ? atomicAdd(&bcur[win], 1u) : 0u;
#endif
    const uint32_t head = 63u - (uint32_t)__builtin_clzll(heads & (~0ull >> (63 - lane)));
    const unsigned long long rest = (heads >> 1) >> lane;
    const uint32_t cnt = rest ? (uint32_t)__builtin_ctzll(rest) + 1u : 64u - (uint32_t)lane;
    const uint32_t first = valid && head == (uint32_t)lane ? atomicAdd(&bcur[win], cnt) : 0u;
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(head << 2), (int)first) + ((uint32_t)lane - head);
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
                                           bool valid, uint32_t ent, uint32_t next) {
    // a run lasts until the next entry's position (positions start over with every block: modulo 1024)
    const uint32_t id = ent >> kPosBits, lenm1 = (next - ent - 1u) & ((1u << kPosBits) - 1u), win = id >> kShortWinBits;
    valid = valid && id < kDummyBase;  // runs of placeholder ids and the entries that only close a run are dropped here
    uint32_t kind = 0, pos;
    if (UNIQ) {
        const uint32_t mask = valid ? (0xFFFFFFFFu >> (31u - lenm1)) << (id & 31u) : 0u;
        const uint32_t old = (FGFA_SHORT_ABLATE & 4) ? 0u : claim_hashed<HASH>(A, seen, w.dummy, w.lane, valid, id >> 5, mask);
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
__device__ __forceinline__ void drain(const ScanArgs &A, Wave &w, uint32_t *seen, uint32_t *bcur, uint32_t *mine, bool all) {
    uint32_t base = 0;
    while (w.fill - base >= 65u || (all && w.fill - base >= 2u)) {
        const uint32_t n = min(64u, w.fill - 1u - base);
        const bool valid = (uint32_t)w.lane < n;
        const uint32_t at = base + (valid ? (uint32_t)w.lane : 0u);
        const uint32_t ent = w.q[at], next = w.q[at + 1u];
        emit_chunk<UNIQ, HASH>(A, w, seen, bcur, mine, valid, ent, next);
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
                                        uint32_t (&a)[16], uint32_t nl, uint32_t rel_lo, uint32_t rel_hi, uint32_t blk_pos) {
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
        drain<UNIQ, HASH>(A, w, seen, bcur, mine, false);
    } else {
        if (QONLY) return false;
        // Entries must lie in the order of their positions, so the block is queued sixteen lanes at a time (at most 256
        // starts and the closing entry), the queue emitted down to at most 64 entries before each.
#pragma unroll 1
        for (uint32_t grp = 0; grp < 4u; ++grp) {
            drain<UNIQ, HASH>(A, w, seen, bcur, mine, false);
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
        drain<UNIQ, HASH>(A, w, seen, bcur, mine, false);
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

template <bool UNIQ, int WAVES, int HASH, bool QONLY, bool PAIRED = false>
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
            if (!block16<UNIQ, HASH, QONLY>(A, w, tab, bcur, mine, a, cur.nl, lo, hi, cur.pos)) {       \
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
                drain<UNIQ, HASH>(A, w, tab, bcur, mine, true);                                         \
                if (UNIQ && !(FGFA_SHORT_ABLATE & 16)) {                                                \
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

template <bool UNIQ>
constexpr auto k_walk_short = k_scan_short<UNIQ, kShortWaves, kShortHash, true>;
template <bool UNIQ>
constexpr auto k_walk_medium = k_scan_short<UNIQ, kMediumWaves, kMediumHash, false, kMediumPaired>;

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
constexpr uint32_t kTinyMax = 128;   // steps
#ifndef FGFA_TINY_BITS
#define FGFA_TINY_BITS 8
#endif
constexpr uint32_t kTinyBits = FGFA_TINY_BITS, kTinyTab = 1u << kTinyBits;   // entries of a wave's id set (at most half full)
constexpr uint32_t kTinyQueue = 64 + kTinyMax;  // a wave's record queue: what is left over + one path of all starts

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

template <bool UNIQ>
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
    uint32_t since[3] = {0u, 0u, 0u};                     // memory operations issued behind each slot's loads
    const auto next_path = [&](uint32_t &b, uint32_t &n) {
        n = 0u;
        b = 0u;
        if (gi >= A.n_short) return;
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
        next_path(pb[D], pn[D]);                                                                    \
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
        FGFA_TINY_REQ(D);                                                                                              \
        const bool v0 = (uint32_t)lane < n, v1 = 64u + (uint32_t)lane < n;                                             \
        uint32_t x0 = a0 >> 1, x1 = a1 >> 1;                                                                           \
        bad |= (v0 && x0 >= A.n_segs) || (v1 && x1 >= A.n_segs);                                                       \
        x0 = x0 < A.n_segs ? x0 : 0u;                                                                                  \
        x1 = x1 < A.n_segs ? x1 : 0u;                                                                                  \
        bool f0 = true, f1 = true;  /* first visits */                                                                 \
        if (UNIQ && !(FGFA_TINY_ABLATE & 1)) {                                                                         \
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

// LDS control words of k_scan, behind the two cursor tables: the next block of the current /
// next item nobody has taken yet (two cells, by item parity), how many waves have left the item
// (two cells), and how many items are complete.
// k_scan's builds: plain, diagnostic (FLATGFA_DEBUG_SKIP), ranged (one of several walks of a graph beyond 16 M segments)
constexpr int kModePlain = 0, kModeDbg = 1, kModeRanged = 2, kModeBig = 3, kModeRangedBig = 4, kModePacked = 5, kModePackedRanged = 6;  // (Big: builds of their own for bucket arrays of 2^30 records or more, see put(); Packed: sub-buckets of exactly the size their records need, items dealt in a fixed order)
constexpr bool mode_ranged(int m) { return m == kModeRanged || m == kModeRangedBig || m == kModePackedRanged; }
constexpr bool mode_big(int m) { return m == kModeBig || m == kModeRangedBig; }
constexpr bool mode_packed(int m) { return m == kModePacked || m == kModePackedRanged; }
// A wave's run queue.  A packed call keeps two LDS tables for up to 4096 windows (cursors and the sub-buckets'
// offsets), which leaves its queues 88 entries less: a block of (nearly) all starts is then queued behind a
// drain down to one entry, and one of more than 1005 starts -- ids without any run at all, which such a plan is
// not made for -- is flagged and the call completed through the atomic kernels.
constexpr uint32_t kQPacked = 64 + 944;
template <int MODE>
constexpr uint32_t kQueueOf = mode_packed(MODE) ? kQPacked : kQ2;
#ifndef FGFA_WIDE
#define FGFA_WIDE 4
#endif
constexpr int kWide = FGFA_WIDE;  // chunks of 64 queue entries k_scan emits side by side
// (kCtlRing cells each for the block counters, the arrival counters and the items, indexed by the
// item's ordinal mod kCtlRing in a tagged call -- a wave with records to append may be kTagSlots items
// ahead of the slowest there, one without any kIdleAhead -- and mod 2 otherwise)
constexpr uint32_t kCtlRing = 32, kIdleAhead = 20;
constexpr uint32_t kCtlNext = 0, kCtlArrive = kCtlRing, kCtlEpoch = 2 * kCtlRing, kCtlJobs = 2 * kCtlRing + 8, kCtlDesc = 3 * kCtlRing + 8,
                   kCtlWords = 7 * kCtlRing + 8;  // (kCtlDesc: four words per cell of the item ring -- the items' descriptors, tagged calls)
// A tagged call deals the items out as the workgroups get to them (an item's tag is its ordinal in
// its workgroup, whatever the item): ctl[kCtlJobs + (r mod kCtlRing)] is the workgroup's r-th item, or one of
constexpr size_t kTprofRow = 4 + kWaves;  // FLATGFA_SCAN_TIME: a workgroup's row of ScanArgs::tprof -- start, end, where it ran, its items, when each wave ran out of work
constexpr uint32_t kJobEmpty = 0xFFFFFFFFu, kJobPending = 0xFFFFFFFEu;  // nobody has asked yet / a wave is fetching it

__device__ __forceinline__ uint32_t epoch_now(uint32_t *ctl) {
    return __hip_atomic_load(ctl + kCtlEpoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Emit `n` queue entries starting at `base`, one per lane; entry base + n must exist (it closes
// the last run).  The whole run must lie below n_segs: that is the bounds check of every step in
// it.  A run that crosses into the next window (at most one: runs are shorter than a window) is
// emitted as two records.
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
    uint32_t win[K], rel[K], lenm1[K], pos[K];
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
        ovf |= put<DBG, mode_big(MODE), mode_packed(MODE)>(A, w, mine, valid[k], pos[k], win[k], rel[k] | (l1 << wb) | w.tagc);
    }
    if (__builtin_amdgcn_ballot_w64(any_cross)) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t pos2 = cross[k] ? atomicAdd(&bcur[win[k] + 1u], 1u) : 0u;
            ovf |= put<DBG, mode_big(MODE), mode_packed(MODE)>(A, w, mine, cross[k], pos2, win[k] + 1u, ((lenm1[k] - (wmask - rel[k]) - 1u) << wb) | w.tagc);
        }
    }
    flag_if_any(A, ovf, kStOverflow);
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
    uint32_t shared;   // items[].z >> 1: 0, or 1 + the ordinal of the split path this item is a piece of
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
    it.src = nullptr;
    if (have) {
        it.dir = (d.z & 1u) ? ~0u : 1u;
        it.shared = d.z >> 1;
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
    if (kWide > 1 && !FGFA_SKIP(kDbgNoEmit) && w.fill >= 64u * (DBG ? 1 : kWide) + 1u) {
        if (!w.epoch_ok && epoch_now(ctl) >= need) w.epoch_ok = true;
        tmark<DBG>(A, w, 1);
        if (w.epoch_ok) drain_raw<MODE, (DBG ? 1 : kWide)>(A, w, bcur, mine, false);
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
    if (TAGGED && MODE == kModePlain && A.tprof && threadIdx.x == 0) {
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
#define FGFA_ITEM_TAG() (TAGGED ? __builtin_amdgcn_readfirstlane(it.shared ? kTagCount - it.shared : rr) << kTagShift : 1u << 24)
    w.tagc = FGFA_ITEM_TAG();
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
        if (blk[SET] < it.nblk) FGFA_LOAD_BLOCK(SET, blk[SET]);                               \
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
            drain_raw<MODE, (DBG ? 1 : kWide)>(A, w, bcur, mine, true);
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
    if (TAGGED && MODE == kModePlain && A.tprof && lane == 0) A.tprof[kTprofRow * blockIdx.x + 4 + wave] = __builtin_amdgcn_s_memrealtime();
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
    if (TAGGED && MODE == kModePlain && A.tprof && threadIdx.x == 0) {
        A.tprof[kTprofRow * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        A.tprof[kTprofRow * blockIdx.x + 3] = rr;  // the items it took
    }
}

// ------------------------------------------------------------------ pass 2 ---

struct AccArgs {
    uint32_t n_segs, n_win, n_slots, cap;
    uint32_t *counts;         // [n_win][n_slots] final cursors (zeroed here: self-cleaning scratch)
    uint32_t *counts0;        // [n_win][n_slots] cursors k_scan started from: records before them carry their kind
    uint32_t has_pre;
    const uint32_t *buckets;
    const uint2 *dir;         // [n_win][dstride]
    const uint32_t *islot;    // [dstride]
    uint32_t dstride;
    const uint32_t *elist;    // k_scan's items in the order pass 2 walks them: item | first-of-its-path << 31
    const uint32_t *wave_off; // [parts * kAccWaves + 1] which stretch of elist each wave (of each of a window's workgroups) walks
    uint32_t n_items;         // static items; handed-back ones follow (one path each)
    const uint32_t *work_counter;
    uint32_t max_back;
    uint32_t *depth_out;
    uint32_t *uniq_out;
    uint32_t *status;
    uint32_t dbg;  // FLATGFA_DEBUG_SKIP ablations (results are then wrong by construction)
    // path depth in the same walk (k_accum<false, 12, true>): per path of k_scan's items, the sums of
    // measure_path (depth.rs:116-131) over the steps that fall into this window
    const uint4 *items;                  // item j belongs to path items[j].w
    const uint32_t *seg_len;
    ulonglong2 *psum_part;               // [n_win][dstride] {sum len, sum depth * len} of item j in this window
    // Paths too long for one wave (more than half a wave's even share of the steps) are walked by
    // all sixteen waves of one of the window's workgroups together, one such path after the
    // other, on one shared bitset:
    const uint32_t *fat_off;   // [parts + 1] which of these paths workgroup blockIdx.y walks
    const uint32_t *fat_woff;  // [n_fat][kAccWaves + 1] which stretch of elist each wave walks of the path's items
    uint32_t parts;  // workgroups per window (blockIdx.y): each walks its share of the paths / sub-buckets and ADDS its counts to the (zeroed) outputs
    uint32_t n_shared;  // tagged calls: the split paths, whose bitsets all waves of the workgroup share (tags kTagCount - n_shared .. kTagCount - 1)
    uint32_t *tprof;    // FLATGFA_ACC_TIME (diagnostic): sixteen words per wave, see AccTimer
    uint32_t *pair_part;  // k_accum_pair: [n_win][2][depth | revisits][window] the two workgroups' halves
    uint32_t *pair_flag;  // k_accum_pair: [n_win] how many halves are there (zero between calls)
    uint32_t accumulate;  // the outputs hold the counts of the paths walked before (another group of the same call): add to them
    const uint32_t *taken;  // tagged: [n_slots] items each k_scan workgroup took (ScanArgs::taken)
    uint32_t *fullest;      // this range's fullest sub-bucket beyond half the capacity (read and cleared by fast_plan_grow)
    const uint2 *pk;        // packed buckets: [n_win][n_slots] {where the sub-bucket starts in `buckets`, its room}; else null
};

// FLATGFA_ACC_TIME: charge the time since the last mark to phase `ph` of this wave; the wave's
// sixteen words (eight phase times in 10 ns units, eight event counts) are written once, at the end
struct AccTimer {
    uint32_t *buf;
    unsigned long long last;
    uint32_t acc[16];
    __device__ __forceinline__ void start(uint32_t *b) {
        buf = b;
        for (int k = 0; k < 16; ++k) acc[k] = 0;
        if (buf) last = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void mark(int ph) {
        if (!buf) return;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        acc[ph] += (uint32_t)(now - last);
        last = now;
    }
    __device__ __forceinline__ void count(int what, uint32_t n = 1u) {
        if (buf) acc[8 + what] += n;
    }
    __device__ __forceinline__ void finish() {
        if (!buf || (threadIdx.x & 63)) return;
        uint32_t *o = buf + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kAccWaves + (threadIdx.x >> 6)) * 16;
        for (int k = 0; k < 16; ++k) o[k] = acc[k];
    }
};

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
template <int WB>
__device__ __forceinline__ void claim_step(uint32_t rec, uint32_t sb, unsigned long long vm, uint32_t dbase, uint32_t rbase,
                                           uint32_t one, uint32_t mone) {
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
        : [rec] "v"(rec), [sb] "v"(sb), [vm] "s"(vm), [dbase] "s"(dbase), [rbase] "s"(rbase), [one] "v"(one), [mone] "v"(mone),
          [relmask] "i"(kRelMask), [basemask] "i"(kBaseMask), [wb] "i"(WB), [wb5] "i"(WB - 5)
        : "vcc", "memory");
}

// The same for records of one segment each (k_scan_dense's: ids without runs): two depth updates, one returning OR of
// the segment's bit, and under the lanes that found it set the revisit's two updates.  Eleven vector instructions,
// where the C++ rendering had two predicated regions with their exec bookkeeping.
template <int WB>
__device__ __forceinline__ void claim_point(uint32_t rec, uint32_t sb, unsigned long long vm, uint32_t dbase, uint32_t rbase, uint32_t one, uint32_t mone) {
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
        : [rec] "v"(rec), [sb] "v"(sb), [vm] "s"(vm), [dbase] "s"(dbase), [rbase] "s"(rbase), [one] "v"(one), [mone] "v"(mone),
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
template <int WB, bool POINT, bool SHARED, bool LOW = false, int SLOTS = (int)kTagSlots>
__device__ __forceinline__ void apply_tagged(const AccArgs &A, int *D, int *R, uint32_t *bits, const uint2 *scnt2, const uint32_t *wbase,
                                             uint32_t *grab) {
    constexpr uint32_t kW = 1u << WB, kNW = kW / 32u;
    constexpr uint32_t kSlots = (uint32_t)SLOTS;  // private bitsets per wave: kTagSlots, or twice as many where the LDS allows (k_scan's order of the tags holds for any multiple)
    constexpr uint32_t kPriv = kAccWaves * kSlots;  // slot ids: the waves' private bitsets first, the shared ones behind
    const int lane = threadIdx.x & 63;
    const uint32_t lane4 = 4u * (uint32_t)lane;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t y16 = blockIdx.y * kAccWaves, nw = A.parts * kAccWaves;
    const uint32_t shlo = kTagCount - A.n_shared;  // tags from here up name split paths
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
    const auto clear_slots = [&](int from, int to) {  // the bitsets of the tags from .. to change hands
        for (int t = from; t <= to; ++t) {
            uint32_t *bs = bits + (wv * kSlots + ((uint32_t)t & (kSlots - 1u))) * kNW;
            for (uint32_t i = lane; i < kNW / 2u; i += 64) reinterpret_cast<uint2 *>(bs)[i] = make_uint2(0u, 0u);
        }
    };
    const auto claim = [&](uint32_t rec, uint32_t tag, unsigned long long act, bool any_shared) {
        // each lane's bitset: its wave's slot tag mod kSlots, or its split path's
        uint32_t sb = priv_b + ((tag & (kSlots - 1u)) << (WB - 3));
        if (SHARED && any_shared) sb = tag >= shlo ? bits0 + ((kPriv + kTagCount - 1u - tag) << (WB - 3)) : sb;
        if (POINT) {  // every record is one segment (k_scan_dense)
            claim_point<WB>(rec, sb, act, dbase, rbase, one, mone);
        } else if (!(FGFA_TAG_ABLATE & 1)) {
            claim_step<WB>(rec, sb, act, dbase, rbase, one, mone);
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
#define FGFA_TAG_STEP(K, NV, FR)                                                                                       \
    if (NV == 0u) break;                                                                                               \
    {                                                                                                                  \
        const uint32_t rec = LOW ? rec_take_lo<K>() : rec_take_s<K>();                                                 \
        const unsigned long long vm = NV >= 64u ? ~0ull : (1ull << NV) - 1ull;  /* the lanes that hold a record */     \
        const uint32_t last = NV - 1u;                                                                                 \
        if (FR & 2u) {  /* a sub-bucket whose private tags each have a slot: all of them cleared here, none changes hands */ \
            for (uint32_t i = lane; i < kSlots * kNW / 4u; i += 64) reinterpret_cast<uint4 *>(bits + wv * kSlots * kNW)[i] = make_uint4(0u, 0u, 0u, 0u); \
            hmax = (int)kSlots - 1;                                                                                 \
        } else if (FR) hmax = -1;  /* the private slots start over with this sub-bucket */                             \
        FGFA_TAG_GEN(K, NV, FR);                                                                                       \
        const uint32_t tag = rec >> kTagShift;                                                                         \
        const unsigned long long shm = SHARED ? __builtin_amdgcn_ballot_w64(tag >= shlo) & vm : 0ull;                  \
        const unsigned long long pvm = vm & ~shm;  /* the lanes whose tag names an item of their own */                \
        bool general = false;                                                                                          \
        if (!(FGFA_TAG_ABLATE & 2) && (__builtin_amdgcn_ballot_w64((uint32_t)hmax - tag >= kSlots) & pvm)) {        \
            /* tags beyond those met so far (or, which cannot be, kSlots behind): their bitsets change hands. */     \
            /* The last record's tag is the highest unless waves that ran ahead have interleaved the items */          \
            const int c = (int)__builtin_amdgcn_readlane(tag, (int)last);                                              \
            general = !((pvm >> last) & 1ull) || (__builtin_amdgcn_ballot_w64((int)tag > c || (int)(tag + kSlots) <= c) & pvm); \
            if (!general) {                                                                                            \
                clear_slots(max(hmax + 1, c - (int)(kSlots - 1u)), c);                                              \
                hmax = c;                                                                                              \
            }                                                                                                          \
        }                                                                                                              \
        if (!general) {                                                                                                \
            claim(rec, tag, vm, shm != 0ull);                                                                          \
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
                claim(rec, tag, act, shm != 0ull);                                                                     \
                todo &= ~act;                                                                                          \
            } while (todo);                                                                                            \
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
uint32_t tagged_lds_bytes(uint32_t wb, uint32_t n_shared, uint32_t slots = kTagSlots) { return (kAccWaves * slots + n_shared) * ((1u << wb) / 8u); }

// PAIR (tagged, unique depth, no split paths): TWO workgroups per window, each with half of its
// sub-buckets, and both resident on a CU (64 registers, under 80 KB of LDS): the walk issues about
// one instruction per cycle and CU where two are possible, and eight waves per SIMD hide more of its
// LDS round trips than four.  Both leave their partial vectors in scratch; the second one to
// finish adds the other's to its own and writes the results.
template <bool UNIQ, int WB, bool PSUM, bool POINT, bool BIG, bool TAGGED, bool PAIR, int SLOTS = (int)kTagSlots, bool LOWREG = PAIR>
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
        if (PAIR) apply_tagged<WB, POINT, false, true>(A, D, R, tag_bits, scnt2, wbase, &grab);
        else if (LOWREG && A.n_shared) apply_tagged<WB, POINT, true, true, SLOTS>(A, D, R, tag_bits, scnt2, wbase, &grab);
        else if (LOWREG) apply_tagged<WB, POINT, false, true, SLOTS>(A, D, R, tag_bits, scnt2, wbase, &grab);
        else if (A.n_shared) apply_tagged<WB, POINT, true, false, SLOTS>(A, D, R, tag_bits, scnt2, wbase, &grab);
        else apply_tagged<WB, POINT, false, false, SLOTS>(A, D, R, tag_bits, scnt2, wbase, &grab);
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

template <bool UNIQ, int WB, bool PSUM = false, bool POINT = false, bool BIG = false, bool TAGGED = false, int SLOTS = (int)kTagSlots>
__global__ __launch_bounds__(kAccThreads) void k_accum(const AccArgs A) {
    accum_body<UNIQ, WB, PSUM, POINT, BIG, TAGGED, false, SLOTS>(A);
}
template <int WB, bool POINT>
__global__ __launch_bounds__(kAccThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_accum_pair(const AccArgs A) {
    accum_body<true, WB, false, POINT, false, true, true>(A);
}
// Half-size windows, TWO of them resident on a CU (64 registers): eight waves per SIMD without a second
// workgroup's setup, scan and exchange per window.
template <int WB>
__global__ __launch_bounds__(kAccThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_accum_small(const AccArgs A) {
    accum_body<true, WB, false, false, false, true, false, (int)kTagSlots, true>(A);
}

// Plan time: how many runs (as k_scan_short cuts them: +1 continuations, cut at multiples of 32)
// each path has.  One workgroup per path at a time.
__global__ __launch_bounds__(256) void k_count_runs(const uint32_t *__restrict__ steps, const uint32_t *__restrict__ pb,
                                                     const uint32_t *__restrict__ pe, uint32_t n_paths,
                                                     uint32_t *__restrict__ runs, uint32_t *__restrict__ runs_down) {
    __shared__ uint32_t total, total_down;
    for (uint32_t p = blockIdx.x; p < n_paths; p += gridDim.x) {
        if (threadIdx.x == 0) total = total_down = 0;
        __syncthreads();
        const uint64_t b = pb[p], e = pe[p];
        uint32_t mine = 0, down = 0;  // (down: the runs the path has when it is read backwards)
        for (uint64_t i = b + threadIdx.x; i < e; i += 256) {
            const uint32_t id = steps[i] >> 1, before = i == b ? 0u : steps[i - 1] >> 1;
            mine += (i == b || id != before + 1u || (id & 31u) == 0u) ? 1u : 0u;
            down += (i == b || id + 1u != before || (before & 31u) == 0u) ? 1u : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) {
            mine += __shfl_down(mine, off, 64);
            down += __shfl_down(down, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&total, mine);
            atomicAdd(&total_down, down);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            runs[p] = total;
            runs_down[p] = total_down;
        }
        __syncthreads();
    }
}

// Plan time: the steps of the listed paths (x = first step, y = one past the last, z = where the copy
// starts) in reverse order.  One workgroup per path at a time.
__global__ __launch_bounds__(256) void k_reverse_copy(const uint32_t *__restrict__ steps, const uint4 *__restrict__ list, uint32_t n,
                                                      uint32_t *__restrict__ out) {
    for (uint32_t j = blockIdx.x; j < n; j += gridDim.x) {
        const uint4 d = list[j];
        for (uint32_t i = threadIdx.x; i < d.y - d.x; i += 256) out[d.z + i] = steps[d.y - 1u - i];
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
#ifndef FGFA_DENSE_TILE
#define FGFA_DENSE_TILE 8192
#endif
constexpr uint32_t kDenseTile = FGFA_DENSE_TILE;
constexpr int kDensePer = kDenseTile / kThreads;  // steps per thread and tile

// A barrier for LDS traffic only: __syncthreads() also waits for every global load in flight, and
// the next tile's steps are meant to stay in flight across the barriers of this tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

uint32_t dense_lds_bytes(uint32_t nwp) { return (6u * nwp + 64u + kDenseTile + 64u) * 4u; }  // (behind the stage: a sink, the totals of two tiles)

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
        const uint32_t tagc = A.tagged ? ((d.z >> 1) ? kTagCount - (d.z >> 1) : rr) << kTagShift : 1u << 24;  // (see kTagShift)
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

// Plan time: which way each of k_scan's items runs through the segment ids.  Bit 0 of items[j].z = 1 when
// more of its steps follow their predecessor downwards (id - 1) than upwards (id + 1), else 0.
// One workgroup per item at a time.
// the segment a path starts at (plan creation: paths of equal length are dealt out in this order)
__global__ __launch_bounds__(256) void k_first_ids(const uint32_t *__restrict__ steps, const uint32_t *__restrict__ at, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = steps[at[i]] >> 1;
}

__global__ __launch_bounds__(256) void k_item_dirs(const uint32_t *__restrict__ steps, uint4 *__restrict__ items, uint32_t n_items,
                                                    unsigned long long *__restrict__ n_runs) {
    __shared__ uint32_t up, down;
    for (uint32_t j = blockIdx.x; j < n_items; j += gridDim.x) {
        if (threadIdx.x == 0) up = down = 0;
        __syncthreads();
        const uint64_t b = items[j].x, e = items[j].y;
        uint32_t u = 0, d = 0;
        for (uint64_t i = b + 1 + threadIdx.x; i < e; i += 256) {
            const uint32_t id = steps[i] >> 1, before = steps[i - 1] >> 1;
            u += id == before + 1u ? 1u : 0u;
            d += id + 1u == before ? 1u : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) {
            u += __shfl_down(u, off, 64);
            d += __shfl_down(d, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&up, u);
            atomicAdd(&down, d);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            items[j].z = (items[j].z & ~1u) | (down > up ? 1u : 0u);
            atomicAdd(n_runs, (unsigned long long)(e - b) - max(up, down));  // the records k_scan will make of the item (but for window crossings)
        }
        __syncthreads();
    }
}

// (a plan of at most kMaxWin windows may run either build of k_scan: sized for the untagged one)
uint32_t scan_lds_bytes(uint32_t nwp, bool tagged_only = false, bool packed = false) { return ((tagged_only && !packed ? 1u : 2u) * nwp + kCtlWords + kWaves * (packed ? kQPacked : kQ2) * 2u) * 4u; }

#define FAST_TRY(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            return false;                                                                   \
        }                                                                                   \
    } while (0)

// The bucket array: capacity per (window, sub-bucket), bounded by the 24-bit slot arithmetic of
// put() and by memory: up to 2^30 records (4 GB) without asking; beyond -- a graph of many windows
// whose paths run along it leaves most sub-buckets empty and needs the few others deep -- up to a
// quarter of the device memory that is free, 64 GB at most (FLATGFA_BUCKET_GB; the diagnostic build
// of k_scan keeps the 32-bit offsets).
// Returns 1 when it is allocated, 0 when the capacity would be too small to be useful, -1 on a HIP error.
int alloc_buckets(FastPlan *fp, uint64_t want_cap) {
    const uint64_t slots = (uint64_t)fp->n_win * fp->n_slots;
    uint64_t max_cap = ((1ull << 30) - 1) / ((uint64_t)(fp->n_win + 1) * fp->n_slots);
    if (want_cap > max_cap && !fp->dbg && fp->tagged && !fp->n_short && !fp->n_medium && !fp->n_tiny) {  // (k_scan's tagged builds only)
        size_t free_b = 0, total_b = 0;
        uint64_t budget = 64ull << 30;
        if (const char *e = getenv("FLATGFA_BUCKET_GB")) budget = strtoull(e, nullptr, 10) << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<uint64_t>(budget, (uint64_t)free_b / 4 + (fp->buckets ? (slots + fp->n_slots) * (uint64_t)fp->cap * 4 : 0));
        else (void)hipGetLastError();
        max_cap = std::max<uint64_t>(max_cap, budget / 4 / ((uint64_t)(fp->n_win + 1) * fp->n_slots));
    }
    uint64_t cap = std::min(want_cap, max_cap);
    cap = std::min<uint64_t>(cap, ((1ull << 24) - 1) / fp->n_slots);  // window * (n_slots * cap) + pos is a 24-bit multiply
    cap &= cap >= 64 ? ~31ull : ~3ull;  // sub-buckets start on 128-byte lines: neighbours (other workgroups, other XCDs) never share one
    if (cap < 4) return 0;
    if (fp->buckets && cap <= fp->cap) return 1;  // (at its limit: the array stays as it is)
    uint32_t *fresh = nullptr;
    const hipError_t e = hipMalloc(&fresh, (slots + fp->n_slots) * cap * 4);  // (the new one first: a plan that cannot grow keeps what it has)
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error(std::string("hipMalloc(buckets): ") + hipGetErrorString(e));
        return fp->buckets ? 1 : -1;
    }
    if (fp->buckets) (void)hipFree(fp->buckets);
    fp->buckets = fresh;
    fp->cap = (uint32_t)cap;
    return 1;
}

}  // namespace

static int run_range(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                     uint32_t *status, hipStream_t stream, const PathSums *ps, bool count_only);

// The plan of one range of segments, [seg_base, seg_base + n_range): the whole graph, or one of
// the ranges of a graph beyond 16 M segments.
static bool create_range(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, FastPlan *fp, uint32_t seg_base,
                         uint32_t n_range, uint32_t max_win, uint32_t force_wb, uint32_t siblings = 1) {
    *fp = FastPlan();
    fp->seg_base = seg_base;
    fp->n_range = n_range;
    const bool ranged = seg_base != 0 || n_range != g.n_segs;
    // Windows of 4096 segments up to 4 M segments, of 8192 beyond (pass 2 keeps a window's
    // difference array and per-path bitsets in LDS).
    uint32_t wb = g.n_segs <= 1024u * 4096u ? 12u : 13u;
    if (force_wb) wb = force_wb;  // (fast_plan_create: 4096-segment windows on a larger graph, for the sake of its split paths)
    if (const char *f = getenv("FLATGFA_WB")) wb = (uint32_t)strtoul(f, nullptr, 10);
    const uint32_t n_win = (uint32_t)(((uint64_t)n_range + (1u << wb) - 1) >> wb);
    if (n_win > max_win) return true;
    hipDeviceProp_t prop;
    int dev = 0;
    FAST_TRY(hipGetDevice(&dev));
    FAST_TRY(hipGetDeviceProperties(&prop, dev));
    fp->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    fp->n_slots = fp->n_cus;
    if (fp->n_slots > kMaxSlots) return true;
    fp->n_win = n_win;
    fp->wb = wb;
    fp->nwp = (n_win + 1u + 63u) & ~63u;  // (one entry more than windows: a packed plan's offset table ends with the region's end)
    fp->lds_bytes_scan = scan_lds_bytes(fp->nwp, n_win > kMaxWin);
    if (fp->lds_bytes_scan + 64 > kLdsLimit) return true;
    if (const char *d = getenv("FLATGFA_DEBUG_SKIP")) fp->dbg = (uint32_t)strtoul(d, nullptr, 10);
    if (fp->dbg && ranged) return true;  // the diagnostic build of k_scan has no registers left for ranges
    // Paths of at most `short_max` steps are walked by single waves (k_scan_short), unless their
    // last block would reach beyond the step array.  Those kernels address at most 256 windows
    // of 4096 segments.
    uint64_t short_max = (fp->dbg || ranged || wb != kShortWinBits || n_win > kShortMaxWin || g.n_segs > kShortMaxSegs) ? 0 : kShortMax;  // (the wave-per-path kernels know nothing of ranges)
    if (const char *forced = getenv("FLATGFA_SHORT_MAX")) short_max = std::min<uint64_t>(short_max, strtoull(forced, nullptr, 10));
    // Which kernel walks a path depends on how many runs it has: short paths must fit the run queue,
    // paths with at most kMediumRuns runs are walked wave by wave too, by pairs of waves that share a
    // bigger hash set (k_scan_short's medium variant).  The counts come from a one-off kernel.
    std::vector<uint32_t> runs, runs_down;
    if (short_max) {
        uint32_t *d_runs = nullptr;
        FAST_TRY(hipMalloc(&d_runs, (size_t)g.n_paths * 8));
        hipLaunchKernelGGL(k_count_runs, dim3(std::min<uint32_t>(g.n_paths, fp->n_cus * 8u)), dim3(256), 0, nullptr, g.steps,
                           g.path_begin, g.path_end, g.n_paths, d_runs, d_runs + g.n_paths);
        runs.resize(g.n_paths);
        runs_down.resize(g.n_paths);
        hipError_t e = hipMemcpy(runs.data(), d_runs, (size_t)g.n_paths * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(runs_down.data(), d_runs + g.n_paths, (size_t)g.n_paths * 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_runs);
        FAST_TRY(e);
    }
    const bool short_any = getenv("FLATGFA_SHORT_ANY") != nullptr;  // tests: let k_scan_short find out and hand back
    const bool no_rev = getenv("FLATGFA_NO_REVERSED_COPIES") != nullptr;  // (measurements)
    // A wave-per-path kernel only knows runs that go up.  A path that walks the ids downwards (a
    // contig on the reverse strand) has far fewer runs when it is read backwards, and the order of a
    // path's steps does not matter to the counts: such a path is walked from a reversed copy of its
    // steps, made here once (rev_steps; every copy starts at a multiple of 16).
    std::vector<uint4> items, short_items, medium_items, short_rev, medium_rev, whole, rev_list, tiny_items;
    const bool no_tiny = getenv("FLATGFA_NO_TINY") != nullptr;  // (measurements, tests: tiny paths go to k_scan_short as before)
    uint64_t rev_len = 0;
    for (uint32_t p = 0; p < g.n_paths; ++p) {
        const uint64_t b = hb[p], e = he[p], n = e - b;
        if (n == 0) continue;
        const bool in_reach = ((e + 15) & ~15ull) <= g.n_steps;  // the last block must not read past the step array
        const bool down = short_max && !no_rev && runs_down[p] < runs[p] && rev_len + n + 2048 < 0xFFFFFFFFull;
        const uint32_t rn = short_max ? (down ? runs_down[p] : runs[p]) : 0u;
        const bool is_tiny = n <= std::min<uint64_t>(short_max, kTinyMax) && !no_tiny;  // (k_scan_tiny: a wave holds the whole path)
        const bool is_short = !is_tiny && n <= short_max && (down || in_reach) && (rn + 16 <= kQCap || short_any);
        const bool is_medium = !is_tiny && !is_short && short_max && (down || in_reach) && rn <= kMediumRuns;
        if (is_tiny) {
            tiny_items.push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else if ((is_short || is_medium) && down) {
            const uint32_t at = (uint32_t)rev_len;
            rev_list.push_back(make_uint4((uint32_t)b, (uint32_t)e, at, p));
            (is_short ? short_rev : medium_rev).push_back(make_uint4(at, at + (uint32_t)n, kNoSlot, p));
            rev_len += (n + 15) & ~15ull;
        } else if (is_short) {
            short_items.push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else if (is_medium) {
            medium_items.push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else {
            whole.push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        }
    }
    // Paths of equal length -- the ties of the sort below -- are walked in the order of where they
    // start: k_scan's workgroups take the items one after the other, so neighbours in the list go to
    // different workgroups, and it is the paths that start near each other that meet in a window.  A
    // sub-bucket then holds one path's records of its window, not those of the five or six that
    // chance gave one workgroup (the capacity every sub-bucket gets is the fullest one's, §2).
    if (whole.size() > 1 && !getenv("FLATGFA_KEEP_PATH_ORDER")) {
        std::vector<uint32_t> at(whole.size()), first(whole.size(), 0u);
        for (size_t i = 0; i < whole.size(); ++i) at[i] = whole[i].x;
        uint32_t *d_at = nullptr;
        FAST_TRY(hipMalloc(&d_at, whole.size() * 8));
        hipError_t e = hipMemcpy(d_at, at.data(), whole.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_first_ids, dim3((uint32_t)((whole.size() + 255) / 256)), dim3(256), 0, nullptr, g.steps, d_at, (uint32_t)whole.size(),
                               d_at + whole.size());
            e = hipMemcpy(first.data(), d_at + whole.size(), whole.size() * 4, hipMemcpyDeviceToHost);
        }
        (void)hipFree(d_at);
        FAST_TRY(e);
        std::vector<uint32_t> order(whole.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });
        std::vector<uint4> sorted(whole.size());
        for (size_t i = 0; i < whole.size(); ++i) sorted[i] = whole[order[i]];
        whole.swap(sorted);
    }
    if (!rev_list.empty()) {
        fp->n_rev_steps = (uint32_t)(rev_len + 1024);  // (a block is read whole)
        FAST_TRY(hipMalloc(&fp->rev_steps, (size_t)fp->n_rev_steps * 4));
        FAST_TRY(hipMemset(fp->rev_steps, 0, (size_t)fp->n_rev_steps * 4));
        uint4 *d_list = nullptr;
        FAST_TRY(hipMalloc(&d_list, rev_list.size() * sizeof(uint4)));
        hipError_t e = hipMemcpy(d_list, rev_list.data(), rev_list.size() * sizeof(uint4), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_reverse_copy, dim3(std::min<uint32_t>((uint32_t)rev_list.size(), fp->n_cus * 8u)), dim3(256), 0, nullptr,
                               g.steps, d_list, (uint32_t)rev_list.size(), fp->rev_steps);
            e = hipDeviceSynchronize();
        }
        (void)hipFree(d_list);
        FAST_TRY(e);
    }
    // k_scan's work items: whole paths, except that a path longer than `piece` steps is cut into
    // equal pieces, so that graphs with few long paths still fill the chip.  The workgroups take the
    // items, longest first, in a fixed snake order (item_of): with a hundred paths of a million steps
    // and pieces of N / (2 CUs), three pieces fall to some workgroups and two to most (+28 % on the
    // longest).  So the deal is played through on the host for a few piece sizes, every item charged
    // a few blocks' worth for its turnaround, and the size with the shortest longest hand is taken
    // (the largest such size: for 1000 paths of 100 k steps, no cutting at all).
    // (z: bit 0 = the item walks the ids downwards, set by k_item_dirs; from bit 1 up, 1 + the
    // ordinal of the split path the item is a piece of, or 0 for a whole path)
    const auto cut = [&](uint64_t piece, std::vector<uint4> *out) -> uint32_t {
        uint32_t n_split = 0;
        for (const uint4 &w : whole) {
            const uint64_t b = w.x, n = (uint64_t)w.y - w.x;
            const uint32_t k = (uint32_t)((n + piece - 1) / piece);
            const uint32_t z = k > 1 ? (++n_split) << 1 : 0u;
            for (uint32_t j = 0; j < k; ++j)
                out->push_back(make_uint4((uint32_t)(b + n * j / k), (uint32_t)(b + n * (j + 1) / k), z, w.w));
        }
        return n_split;
    };
    const auto longer = [](const uint4 &a, const uint4 &b) { return a.y - a.x > b.y - b.x; };
    uint64_t piece = 0;
    if (const char *forced = getenv("FLATGFA_PIECE_STEPS")) {
        piece = (std::max<uint64_t>(256, strtoull(forced, nullptr, 10)) + 255) & ~255ull;
    } else {
        constexpr uint64_t kTurnaround = 8192;  // steps a workgroup could have walked while it changes items
        uint64_t long_steps = 0;
        for (const uint4 &w : whole) long_steps += w.y - w.x;
        uint64_t best = ~0ull;
        for (const uint32_t twice_m : {4u, 5u, 6u, 8u, 10u, 12u, 16u, 20u, 24u}) {  // pieces of N / (m CUs), m = 2 .. 12
            uint64_t cand = std::max<uint64_t>(32768, (2 * long_steps + (uint64_t)twice_m * fp->n_slots - 1) / ((uint64_t)twice_m * fp->n_slots));
            cand = (cand + 255) & ~255ull;
            std::vector<uint4> trial;
            cut(cand, &trial);
            std::stable_sort(trial.begin(), trial.end(), longer);
            std::vector<uint64_t> hand(fp->n_slots, 0);
            for (size_t i = 0; i < trial.size(); ++i) {
                const size_t round = i / fp->n_slots, pos = i % fp->n_slots;
                hand[(round & 1) ? fp->n_slots - 1 - pos : pos] += (uint64_t)(trial[i].y - trial[i].x) + kTurnaround;
            }
            const uint64_t longest = *std::max_element(hand.begin(), hand.end());
            if (longest + longest / 64 < best) {  // smaller pieces have to win by more than 1.5 %
                best = longest;
                piece = cand;
            }
            if (cand == 32768) break;
        }
        // Cutting has a price beyond the turnaround where pass 2 cannot give all split paths a bitset
        // of their own kind -- none at all with 8192-segment windows, 128 with smaller ones: the plan
        // then takes smaller windows, more ranges, or walks the paths in groups (fast_plan_create).  A
        // graph of thousands of paths rarely needs its long ones cut: k_scan's workgroups take the
        // items longest first as they get to them, and if whole paths dealt that way (to the least
        // loaded workgroup each) leave the longest hand within a tenth of the best cut's, they stay whole.
        if (piece && !getenv("FLATGFA_KEEP_CUTS")) {
            std::vector<uint4> trial;
            const uint32_t n_split = cut(piece, &trial);
            if (n_split > (wb <= 12 ? kMaxShared : 0u)) {
                std::vector<uint64_t> lens;
                for (const uint4 &w : whole) lens.push_back((uint64_t)w.y - w.x);
                std::sort(lens.begin(), lens.end(), std::greater<uint64_t>());
                std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> hands;
                for (uint32_t i = 0; i < fp->n_slots; ++i) hands.push(0);
                uint64_t longest = 0;
                for (const uint64_t n : lens) {
                    const uint64_t h = hands.top() + n + kTurnaround;
                    hands.pop();
                    hands.push(h);
                    longest = std::max(longest, h);
                }
                if (longest <= best + best / 10) piece = ~0ull >> 1;  // (longer than any path)
            }
        }
    }
    fp->n_shared = cut(piece ? piece : 32768, &items);
    std::stable_sort(items.begin(), items.end(), longer);
    std::stable_sort(short_items.begin(), short_items.end(), longer);
    std::stable_sort(medium_items.begin(), medium_items.end(), longer);
    std::stable_sort(short_rev.begin(), short_rev.end(), longer);
    std::stable_sort(medium_rev.begin(), medium_rev.end(), longer);
    fp->n_short_rev = (uint32_t)short_rev.size();
    fp->n_medium_rev = (uint32_t)medium_rev.size();
    short_items.insert(short_items.end(), short_rev.begin(), short_rev.end());      // (the reversed ones behind the others)
    medium_items.insert(medium_items.end(), medium_rev.begin(), medium_rev.end());
    fp->n_items = (uint32_t)items.size();
    fp->n_short = (uint32_t)short_items.size();
    fp->n_medium = (uint32_t)medium_items.size();
    fp->n_tiny = (uint32_t)tiny_items.size();
    {
        const auto steps_of = [](const std::vector<uint4> &v) {
            uint64_t n = 0;
            for (const uint4 &d : v) n += d.y - d.x;
            return n;
        };
        fp->class_steps[0] = steps_of(items);
        fp->class_steps[1] = steps_of(short_items);
        fp->class_steps[2] = steps_of(medium_items);
        fp->class_steps[3] = steps_of(tiny_items);
    }
    if (items.empty() && short_items.empty() && medium_items.empty() && tiny_items.empty()) return true;
    fp->max_back = std::min<uint32_t>(fp->n_short, kMaxHandBack);
    fp->exact_short = !short_any;  // the run counts the lists were made from are exact: nothing is handed back
    fp->dstride = fp->n_items + fp->max_back + 1;
    // The directory (one cursor pair per item and window) must stay small next to the steps.
    if ((uint64_t)fp->dstride * n_win * 8 > std::max<uint64_t>(64ull << 20, g.n_steps * 2)) {
        *fp = FastPlan();
        return true;
    }
    // Pass 2 runs one workgroup per window -- or, when the graph has fewer windows than half the
    // CUs, several that share the window's paths and sub-buckets and add their counts up (1000
    // paths over 100 k segments: 25 workgroups took 0.48 ms where 250 take 0.06).
    fp->acc_parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({16, fp->n_cus / n_win, (g.n_steps / n_win + (32u << 10) - 1) >> 15}));  // a workgroup per 32 k steps: a wave's walk is a chain of dependent round trips, a microsecond per 64 records
    if (const char *f = getenv("FLATGFA_ACC_PARTS")) fp->acc_parts = std::max(1u, std::min(64u, (uint32_t)strtoul(f, nullptr, 10)));
    const uint32_t acc_waves = fp->acc_parts * kAccWaves;
    // Tagged calls (records say whose they are; see kTagShift): every workgroup's items -- the handed-back
    // ones included -- must have tags of their own next to the split paths', and pass 2 needs LDS for a
    // bitset per split path (none to spare with 8192-segment windows; a window shared by several
    // workgroups cannot share bitsets).  FLATGFA_TAGGED=0 keeps the directory (tests, measurements).
    {
        const uint32_t grid = (fp->n_short || fp->n_medium || fp->n_tiny) ? fp->n_slots : std::min<uint32_t>(fp->n_items, fp->n_slots);
        const uint64_t per_wg = grid ? ((uint64_t)fp->n_items + fp->max_back + grid - 1) / grid : 0;
        const char *t = getenv("FLATGFA_TAGGED");
        const uint32_t shared_cap = wb <= 12 ? kMaxShared : 0u;
        const bool base_ok = !fp->dbg && !(t && t[0] == '0') && (fp->n_shared == 0 || fp->acc_parts == 1);
        const bool taggable = base_ok && fp->n_shared <= shared_cap;
        // A workgroup's private tags: what the split paths leave of the 512 (FLATGFA_TAG_LIMIT: fewer, tests).
        // k_scan deals the items out as its workgroups get to them, so it is not the mean that has to
        // fit but the most any workgroup takes: the deal is played through here (its first two items
        // are fixed, every further one goes to whoever is done first), with some room to spare --
        // a workgroup that does run out of tags stops taking items (k_scan) and the others go on.
        uint32_t limit = fp->n_shared < kTagCount ? kTagCount - fp->n_shared : 0u;
        if (const char *f = getenv("FLATGFA_TAG_LIMIT")) limit = std::min<uint32_t>(limit, std::max(2u, (uint32_t)strtoul(f, nullptr, 10)));
        fp->tag_limit = limit;
        uint64_t most = per_wg;
        if (taggable && grid && per_wg <= limit && fp->n_items + fp->max_back > 2ull * grid && !getenv("FLATGFA_TAG_MEAN_ONLY")) {
            constexpr uint64_t kTurn = 2048;  // steps' worth an item costs beyond its steps
            std::priority_queue<std::pair<uint64_t, uint32_t>, std::vector<std::pair<uint64_t, uint32_t>>, std::greater<std::pair<uint64_t, uint32_t>>> hands;
            std::vector<uint32_t> taken(grid, 0u);
            for (uint32_t i = 0; i < grid; ++i) {
                uint64_t h = 0;
                for (uint32_t k = 0; k < 2; ++k)
                    if (i + k * grid < fp->n_items) h += (uint64_t)(items[i + k * grid].y - items[i + k * grid].x) + kTurn, taken[i] += 1;
                hands.push({h, i});
            }
            for (uint64_t j = 2ull * grid; j < (uint64_t)fp->n_items + fp->max_back; ++j) {
                const uint64_t n = j < fp->n_items ? (uint64_t)(items[j].y - items[j].x) : kShortMax;
                auto [h, i] = hands.top();
                hands.pop();
                taken[i] += 1;
                hands.push({h + n + kTurn, i});
            }
            most = *std::max_element(taken.begin(), taken.end());
            most += most / 4;  // (the workgroups do not run at one speed)
        }
        fp->tagged = taggable && per_wg <= limit && most <= limit;
        // what fast_plan_create may do about a plan that is not: walk the paths in groups (fewer items, fewer
        // split paths per group), or take 4096-segment windows (pass 2 then has LDS for split paths' bitsets)
        fp->too_many_items = base_ok && !fp->tagged && fp->acc_parts == 1 && !fp->n_short && !fp->n_medium && !fp->n_tiny && (taggable || (wb <= 12 && fp->n_shared > shared_cap));
        fp->want_wb12 = base_ok && !fp->tagged && wb == 13 && fp->n_shared > 0 && fp->acc_parts == 1;
        if (!fp->tagged && n_win > kMaxWin) {  // so many windows only without cursor snapshots: the caller cuts smaller ranges
            const bool many = fp->too_many_items, w12 = fp->want_wb12;
            fast_plan_destroy(fp);
            fp->too_many_items = many;
            fp->want_wb12 = w12;
            return true;
        }
    }
    // Pass 2 walks k_scan's items grouped by path (the pieces of a split path share a bitset),
    // each of its waves a contiguous stretch of the list: paths are dealt to the waves longest
    // first, each to the wave with the least steps so far.
    {
        std::vector<std::vector<uint32_t>> by_path;  // item indices per path that has items
        std::vector<uint64_t> path_steps;
        std::vector<int64_t> slot_of(g.n_paths, -1);
        for (uint32_t j = 0; j < fp->n_items; ++j) {
            const uint32_t p = items[j].w;
            if (slot_of[p] < 0) {
                slot_of[p] = (int64_t)by_path.size();
                by_path.emplace_back();
                path_steps.push_back(0);
            }
            by_path[(size_t)slot_of[p]].push_back(j);
            path_steps[(size_t)slot_of[p]] += items[j].y - items[j].x;
        }
        std::vector<uint32_t> order(by_path.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return path_steps[a] > path_steps[b]; });
        // A path with more than half a wave's even share of the steps would hold its wave up (four
        // paths of 25 M steps: four waves busy out of sixteen, pass 2 2.6 times slower).  Such
        // paths go to the window's workgroups whole -- to the one with the least so far -- and
        // their pieces to its sixteen waves in turn; the others are dealt to single waves as before.
        uint64_t total_steps = 0;
        for (uint64_t v : path_steps) total_steps += v;
        const uint64_t fat_min = total_steps / (2ull * acc_waves) + 1;
        std::vector<std::vector<uint32_t>> per_wave(acc_waves), fat_of_part(fp->acc_parts);
        std::vector<uint64_t> load(acc_waves, 0), part_load(fp->acc_parts, 0);
        for (uint32_t gi : order) {
            if (path_steps[gi] < fat_min || by_path[gi].size() < kAccWaves / 2 || getenv("FLATGFA_NO_FAT_PATHS")) continue;  // (fewer pieces than half the waves: better one wave busy all the time than three)
            const uint32_t q = (uint32_t)(std::min_element(part_load.begin(), part_load.end()) - part_load.begin());
            part_load[q] += path_steps[gi];
            fat_of_part[q].push_back(gi);
        }
        for (uint32_t q = 0; q < fp->acc_parts; ++q)
            for (uint32_t wv = 0; wv < kAccWaves; ++wv) load[q * kAccWaves + wv] = part_load[q] / kAccWaves;
        std::vector<bool> is_fat(by_path.size(), false);
        for (const auto &v : fat_of_part)
            for (uint32_t gi : v) is_fat[gi] = true;
        for (uint32_t gi : order) {
            if (is_fat[gi]) continue;
            const uint32_t wv = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
            load[wv] += path_steps[gi] + 64;
            bool first = true;
            for (uint32_t j : by_path[gi]) {
                per_wave[wv].push_back(j | (first ? 0x80000000u : 0u));
                first = false;
            }
        }
        std::vector<uint32_t> elist, wave_off(acc_waves + 1, 0);
        for (uint32_t wv = 0; wv < acc_waves; ++wv) {
            wave_off[wv] = (uint32_t)elist.size();
            elist.insert(elist.end(), per_wave[wv].begin(), per_wave[wv].end());
        }
        wave_off[acc_waves] = (uint32_t)elist.size();
        std::vector<uint32_t> fat_off(fp->acc_parts + 1, 0), fat_woff;
        for (uint32_t q = 0; q < fp->acc_parts; ++q) {
            fat_off[q] = fp->n_fat;
            for (uint32_t gi : fat_of_part[q]) {
                const std::vector<uint32_t> &its = by_path[gi];
                for (uint32_t wv = 0; wv < kAccWaves; ++wv) {
                    fat_woff.push_back((uint32_t)elist.size());
                    bool first = true;
                    for (size_t k = wv; k < its.size(); k += kAccWaves) {
                        elist.push_back(its[k] | (first ? 0x80000000u : 0u));
                        first = false;
                    }
                }
                fat_woff.push_back((uint32_t)elist.size());
                fp->n_fat += 1;
            }
        }
        fat_off[fp->acc_parts] = fp->n_fat;
        FAST_TRY(hipMalloc(&fp->fat_off, fat_off.size() * 4));
        FAST_TRY(hipMemcpy(fp->fat_off, fat_off.data(), fat_off.size() * 4, hipMemcpyHostToDevice));
        FAST_TRY(hipMalloc(&fp->fat_woff, (fat_woff.size() + 1) * 4));
        if (!fat_woff.empty()) FAST_TRY(hipMemcpy(fp->fat_woff, fat_woff.data(), fat_woff.size() * 4, hipMemcpyHostToDevice));
        // k_scan leaves an item's cursors and sub-bucket at the item's place in this order
        std::vector<uint32_t> perm(fp->n_items + 1, 0);
        for (size_t at = 0; at < elist.size(); ++at) perm[elist[at] & 0x7FFFFFFFu] = (uint32_t)at | (elist[at] & 0x80000000u);
        FAST_TRY(hipMalloc(&fp->perm, perm.size() * 4));
        FAST_TRY(hipMemcpy(fp->perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
        FAST_TRY(hipMalloc(&fp->elist, (elist.size() + 1) * 4));
        if (!elist.empty()) FAST_TRY(hipMemcpy(fp->elist, elist.data(), elist.size() * 4, hipMemcpyHostToDevice));
        FAST_TRY(hipMalloc(&fp->wave_off, wave_off.size() * 4));
        FAST_TRY(hipMemcpy(fp->wave_off, wave_off.data(), wave_off.size() * 4, hipMemcpyHostToDevice));
    }
    // Worst case is one record per step (plus one per block and window crossing) for k_scan and
    // one depth plus one uniq record per step for k_scan_short, spread evenly over the
    // sub-buckets; real graphs need a fraction of that (runs), skewed ones more, so the
    // capacity starts at the even share of N records + 25% and grows on demand (fast_plan_grow).
    const uint64_t slots = (uint64_t)n_win * fp->n_slots;
    uint64_t walked = 0;  // (the steps the paths span, not the pool: a plan over a few paths of a large graph needs little)
    for (uint32_t p = 0; p < g.n_paths; ++p) walked += he[p] - hb[p];
    uint64_t cap = (std::min<uint64_t>(g.n_steps, walked) + slots - 1) / slots;
    if (ranged) cap = (uint64_t)((double)cap * n_range / g.n_segs) + 1;  // a range sees its share of the runs
    cap = cap + cap / 4 + 256;
    if (const char *forced = getenv("FLATGFA_BUCKET_CAP")) {  // tests: force the overflow route
        cap = strtoull(forced, nullptr, 10);
        fp->cap_forced = true;
    }
    // Packed buckets (below, once the items are on the device): where an even share for every sub-bucket would take
    // gigabytes -- whole-genome graphs, whose paths leave most sub-buckets of a window empty and a few deep --
    // the plan counts what every sub-bucket gets and lays them out back to back.  For tagged plans whose
    // records all come from k_scan; FLATGFA_PACKED=0|1 never / whenever possible (tests, measurements).
    const bool can_pack = fp->tagged && !fp->n_short && !fp->n_medium && !fp->n_tiny && !fp->dbg && !fp->cap_forced && n_win <= kMaxWinTagged && fp->acc_parts == 1 &&
                          scan_lds_bytes(fp->nwp, true, true) + 64 <= kLdsLimit;
    // (the even layout may take 2 GB for a graph's buckets: this plan's share when segment ranges and path groups make several of it)
    bool want_packed = can_pack && (slots + fp->n_slots) * std::max<uint64_t>(cap, 4) * 4 > std::max<uint64_t>(128ull << 20, (2ull << 30) / std::max(1u, siblings));
    if (const char *f = getenv("FLATGFA_PACKED")) want_packed = can_pack && strtol(f, nullptr, 10) != 0;
    if (!want_packed) {
        const int rc = alloc_buckets(fp, std::max<uint64_t>(cap, 4));
        if (rc < 0) return false;
        if (rc == 0) return true;  // not eligible; the caller's destroy releases what was allocated
        fp->bucket_records = (slots + fp->n_slots) * (uint64_t)fp->cap;
    }
    FAST_TRY(hipMalloc(&fp->counts, slots * 4));
    FAST_TRY(hipMemset(fp->counts, 0, slots * 4));
    FAST_TRY(hipMalloc(&fp->taken, ((size_t)fp->n_slots + 1) * 4));
    FAST_TRY(hipMemset(fp->taken, 0xFF, (size_t)fp->n_slots * 4));  // (nothing known until a tagged k_scan has run)
    FAST_TRY(hipMemset(fp->taken + fp->n_slots, 0, 4));              // (the word behind them: this range's fullest sub-bucket beyond half the capacity)
    FAST_TRY(hipMalloc(&fp->counts0, slots * 4));
    FAST_TRY(hipMemset(fp->counts0, 0, slots * 4));
    FAST_TRY(hipMalloc(&fp->dir, (size_t)fp->dstride * n_win * sizeof(uint2)));
    FAST_TRY(hipMemset(fp->dir, 0, (size_t)fp->dstride * n_win * sizeof(uint2)));
    FAST_TRY(hipMalloc(&fp->islot, (size_t)fp->dstride * 4));
    FAST_TRY(hipMemset(fp->islot, 0, (size_t)fp->dstride * 4));
    FAST_TRY(hipMalloc(&fp->items, (items.size() + fp->max_back + 1) * sizeof(uint4)));
    if (!items.empty()) {
        FAST_TRY(hipMemcpy(fp->items, items.data(), items.size() * sizeof(uint4), hipMemcpyHostToDevice));
        if (!getenv("FLATGFA_NO_ITEM_DIRS")) {  // (knob for measurements: every item taken as running upwards)
            unsigned long long *d_runs64 = nullptr, runs64 = 0, item_steps = 0;
            FAST_TRY(hipMalloc(&d_runs64, 8));
            FAST_TRY(hipMemset(d_runs64, 0, 8));
            hipLaunchKernelGGL(k_item_dirs, dim3(std::min<uint32_t>(fp->n_items, fp->n_cus * 8u)), dim3(256), 0, nullptr, g.steps,
                               reinterpret_cast<uint4 *>(fp->items), fp->n_items, d_runs64);
            const hipError_t e = hipMemcpy(&runs64, d_runs64, 8, hipMemcpyDeviceToHost);
            (void)hipFree(d_runs64);
            FAST_TRY(e);
            for (const uint4 &it : items) item_steps += it.y - it.x;
            fp->est_records = runs64;
            // more than three records for four steps: not worth looking for runs (k_scan_dense)
            const bool can = !fp->dbg && dense_lds_bytes(fp->nwp) + 64 <= kLdsLimit;
            // More than a record for two steps: k_scan_dense may be the better pass 1 -- when the
            // ids jump about; steps that stay in one window make its LDS atomics queue on one
            // address.  Sized for the dense form (one record per step is the most either makes),
            // then timed both ways by the plan's creator.
            fp->dense_maybe = can && runs64 * 2 > item_steps;
            fp->dense = fp->dense_maybe;
        }
        if (const char *f = getenv("FLATGFA_DENSE")) {  // tests, measurements
            fp->dense = !fp->dbg && strtol(f, nullptr, 10) != 0 && dense_lds_bytes(fp->nwp) + 64 <= kLdsLimit;
            fp->dense_maybe = false;
        }
    }
    if (!short_items.empty()) {
        FAST_TRY(hipMalloc(&fp->short_items, short_items.size() * sizeof(uint4)));
        FAST_TRY(hipMemcpy(fp->short_items, short_items.data(), short_items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    if (!medium_items.empty()) {
        FAST_TRY(hipMalloc(&fp->medium_items, medium_items.size() * sizeof(uint4)));
        FAST_TRY(hipMemcpy(fp->medium_items, medium_items.data(), medium_items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    if (!tiny_items.empty()) {
        FAST_TRY(hipMalloc(&fp->tiny_items, tiny_items.size() * sizeof(uint4)));
        FAST_TRY(hipMemcpy(fp->tiny_items, tiny_items.data(), tiny_items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    {
        std::vector<uint32_t> other;
        for (const uint4 &it : short_items) other.push_back(it.w);
        for (const uint4 &it : medium_items) other.push_back(it.w);
        for (const uint4 &it : tiny_items) other.push_back(it.w);
        fp->n_other = (uint32_t)other.size();
        if (!other.empty()) {
            FAST_TRY(hipMalloc(&fp->other_ids, other.size() * 4));
            FAST_TRY(hipMemcpy(fp->other_ids, other.data(), other.size() * 4, hipMemcpyHostToDevice));
        }
    }
    fp->lds_bytes_short = (kShortMaxWin + kShortWaves * (kQCap + 2 * kPCap + (2u << kShortHash)) + 128u + 16u) * 4u;
    fp->lds_bytes_medium = kMediumPaired ? (kShortMaxWin + kMediumWaves * (kQPaired + 2 * kPCap) + (kMediumWaves / 2) * (2u << kMediumHash) + 128u + 16u) * 4u
                                         : (kShortMaxWin + kMediumWaves * (kQCap + 2 * kPCap + (2u << kMediumHash)) + 128u + 16u) * 4u;
    // (the attribute belongs to the kernel, not to the plan: plans of different sizes live side by side)
    FAST_TRY(hipFuncSetAttribute((const void *)k_walk_short<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_walk_short<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_walk_medium<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_walk_medium<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    fp->lds_bytes_tiny = (kShortMaxWin + kWaves * kTinyTab + kWaves * kTinyQueue * 2u + 128u) * 4u;
    FAST_TRY(hipMalloc(&fp->work_counter, 256));
    FAST_TRY(hipMemset(fp->work_counter, 0, 256));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModePlain, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModePlain, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModeDbg, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModeRanged, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModeRanged, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModeBig, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModeRangedBig, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModePacked, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan<kModePackedRanged, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    // pass 2 of a tagged call keeps its bitsets in dynamic shared memory (next to about 60 KB of static arrays, 93 KB with 8192-segment windows)
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum<true, 11, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(11, kMaxShared)));
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum<true, 12, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(12, kMaxShared)));
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum<true, 12, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(12, kMaxShared)));
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum<true, 12, false, false, false, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(12, 64, 8)));
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum<true, 13, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(13, 0)));
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum<true, 13, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(13, 0)));
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum_small<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(11, kMaxShared)));
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum_pair<12, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(12, 0)));
    FAST_TRY(hipFuncSetAttribute((const void *)k_accum_pair<12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tagged_lds_bytes(12, 0)));
    // Two pass-2 workgroups per window (k_accum_pair): tagged plans without split paths whose windows
    // do not fill the chip twice over anyway.  FLATGFA_ACC_PAIR=1 (measurements, tests).
    // In round 3 it paid where a window had 64 k records or more (the chromosome model 4 % faster, ten
    // thousand contigs 13 %); since the one-workgroup walk clears a plain sub-bucket's bitsets at once
    // and takes eight bitsets per wave where tags are many, it no longer does (same box: chromosome model
    // 95 against 79 us, ids without runs 175 against 149, ten thousand contigs 165 against 118) -- and the
    // table of sub-bucket starts (packed buckets) took the 4 KB of LDS that let two of its workgroups share
    // a CU: off by default.
    const bool pair_ok = fp->tagged && fp->n_shared == 0 && wb == 12 && fp->acc_parts == 1 && n_win <= 2 * fp->n_cus;
    fp->acc_pair = false;
    if (const char *f = getenv("FLATGFA_ACC_PAIR")) fp->acc_pair = pair_ok && strtol(f, nullptr, 10) != 0;
    // Private bitsets per wave of the tagged walk: four, or eight where a sub-bucket holds the records of more
    // items than that (a 64-record step then spans more tags than four bitsets serve in one round: 3125 paths
    // of 32 k steps 77 -> 58 us, 16 000 paths of 100 k 0.81 -> 0.56 ms) and the LDS allows it: 4096-segment
    // windows, at most 64 split paths; one workgroup per window then (k_accum_pair has no room for them).
    // With four items or fewer per workgroup four are as good and cheaper to clear (cfg-L: +3 % with eight).
    {
        const uint32_t grid = std::min<uint32_t>(std::max<uint32_t>(fp->n_items, 1u), fp->n_slots);
        const bool can8 = fp->tagged && wb == 12 && fp->n_shared <= 64 && fp->acc_parts == 1;
        fp->acc_slots = can8 && (uint64_t)fp->n_items + fp->max_back > 4ull * grid ? 8u : kTagSlots;
        if (const char *f = getenv("FLATGFA_ACC_SLOTS")) fp->acc_slots = can8 && strtol(f, nullptr, 10) == 8 ? 8u : kTagSlots;
        if (fp->acc_slots == 8) fp->acc_pair = false;
    }
    if (fp->acc_pair) {
        FAST_TRY(hipMalloc(&fp->pair_part, (size_t)n_win * 2 * 2 * (1u << wb) * 4));
        FAST_TRY(hipMalloc(&fp->pair_flag, (size_t)n_win * 4));
        FAST_TRY(hipMemset(fp->pair_flag, 0, (size_t)n_win * 4));
    }
    FAST_TRY(hipFuncSetAttribute((const void *)k_scan_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    if (want_packed && (fp->dense || fp->dense_maybe)) {  // (pass 1 by partition keeps the even layout)
        want_packed = false;
        const int rc = alloc_buckets(fp, std::max<uint64_t>(cap, 4));
        if (rc < 0) return false;
        if (rc == 0) return true;
        fp->bucket_records = (slots + fp->n_slots) * (uint64_t)fp->cap;
    }
    if (want_packed) {
        // The counting call: k_scan alone, every sub-bucket without room (all records go to a sink), the
        // items dealt in the fixed order every later call uses.  Its cursors are the layout.
        const size_t row = (size_t)n_win + 1;
        uint32_t *d_status = nullptr;
        FAST_TRY(hipMalloc(&fp->pk_off, row * fp->n_slots * 4));
        FAST_TRY(hipMemset(fp->pk_off, 0, row * fp->n_slots * 4));
        FAST_TRY(hipMalloc(&fp->pk_base, (size_t)fp->n_slots * 8));
        FAST_TRY(hipMemset(fp->pk_base, 0, (size_t)fp->n_slots * 8));
        FAST_TRY(hipMalloc(&fp->buckets, 4096));
        FAST_TRY(hipMalloc(&d_status, 256));
        FAST_TRY(hipMemset(d_status, 0, 256));
        fp->packed = true;
        fp->cap = 0;
        fp->eligible = true;
        const uint32_t lds_even = fp->lds_bytes_scan;
        fp->lds_bytes_scan = scan_lds_bytes(fp->nwp, true, true);
        const int rc = run_range(*fp, g, nullptr, nullptr, d_status, nullptr, nullptr, true);
        std::vector<uint32_t> cnt(slots);
        hipError_t e = rc == FLATGFA_OK ? hipDeviceSynchronize() : hipErrorUnknown;
        if (e == hipSuccess) e = hipMemcpy(cnt.data(), fp->counts, slots * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemset(fp->counts, 0, slots * 4);
        uint32_t st = 0;
        if (e == hipSuccess) e = hipMemcpy(&st, d_status, 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_status);
        fp->eligible = false;
        FAST_TRY(e);
        if (st & kStBounds) {  // (an id out of range: the atomic kernels report it; no layout to be had)
            fp->packed = false;
            fp->lds_bytes_scan = lds_even;
            return true;
        }
        const bool countable = !(st & kStBackOverflow);  // (blocks without any runs do not fit a packed call's queues: the even layout)
        std::vector<uint32_t> off(row * fp->n_slots);
        std::vector<uint64_t> base(fp->n_slots);
        std::vector<uint2> pk(slots);
        uint64_t total = 0, deepest = 0;
        bool fits = true;
        for (uint32_t gq = 0; gq < fp->n_slots; ++gq) {
            uint64_t o = 0;
            base[gq] = total;
            for (uint32_t wq = 0; wq < n_win; ++wq) {
                const uint64_t room = ((uint64_t)cnt[(size_t)wq * fp->n_slots + gq] + 3) & ~3ull;  // (sub-buckets start on 16 bytes: pass 2 reads four records at a time)
                off[gq * row + wq] = (uint32_t)o;
                pk[(size_t)wq * fp->n_slots + gq] = make_uint2((uint32_t)(total + o), (uint32_t)room);
                deepest = std::max(deepest, room);
                o += room;
            }
            off[gq * row + n_win] = (uint32_t)o;  // the region's sink
            o += 64;
            fits = fits && o < (1ull << 30);      // (a region's byte offsets take 32 bits)
            total += o;
        }
        fits = fits && countable && total < (1ull << 32);
        (void)hipFree(fp->buckets);
        fp->buckets = nullptr;
        if (!fits) {  // (not the case this layout is for: the even one, if it can be had)
            fp->packed = false;
            fp->lds_bytes_scan = lds_even;
            (void)hipFree(fp->pk_off);
            (void)hipFree(fp->pk_base);
            fp->pk_off = nullptr;
            fp->pk_base = nullptr;
            const int rc2 = alloc_buckets(fp, std::max<uint64_t>(cap, 4));
            if (rc2 < 0) return false;
            if (rc2 == 0) return true;
            fp->bucket_records = (slots + fp->n_slots) * (uint64_t)fp->cap;
        } else {
            FAST_TRY(hipMalloc(&fp->buckets, std::max<uint64_t>(total, 64) * 4));
            FAST_TRY(hipMemcpy(fp->pk_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
            FAST_TRY(hipMemcpy(fp->pk_base, base.data(), base.size() * 8, hipMemcpyHostToDevice));
            FAST_TRY(hipMalloc(&fp->pk, slots * sizeof(uint2)));
            FAST_TRY(hipMemcpy(fp->pk, pk.data(), slots * sizeof(uint2), hipMemcpyHostToDevice));
            fp->cap = (uint32_t)std::max<uint64_t>(deepest, 4);  // (what describe() reports: the deepest sub-bucket)
            fp->bucket_records = total;
        }
    }
    fp->eligible = true;
    return true;
}

// The plans of all segment ranges for one set of path spans, appended to `plans`.  *all: every one
// of them is eligible; *many: one of them is kept from tagged calls by its number of items alone.
static bool append_ranges(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, uint32_t max_win, uint32_t force_wb,
                          std::vector<FastPlan> *plans, bool *all, bool *many, bool *want12 = nullptr, uint32_t n_groups = 1) {
    uint64_t max_range = (uint64_t)max_win << (force_wb ? force_wb : 13u);
    if (const char *f = getenv("FLATGFA_RANGE_SEGS")) max_range = std::max<uint64_t>(8192, strtoull(f, nullptr, 10) & ~8191ull);  // tests
    const uint32_t n_ranges = (uint32_t)((g.n_segs + max_range - 1) / max_range);
    if (n_ranges > 64) {
        *all = false;
        return true;
    }
    const uint32_t per = (uint32_t)((((uint64_t)g.n_segs + n_ranges - 1) / n_ranges + 8191) & ~8191ull);
    for (uint32_t r = 0; r < n_ranges; ++r) {
        const uint32_t base = r * per;
        FastPlan q;
        if (!create_range(g, hb, he, &q, base, std::min<uint32_t>(per, g.n_segs - base), max_win, force_wb, n_ranges * n_groups)) {
            fast_plan_destroy(&q);
            return false;
        }
        *many = *many || q.too_many_items;
        if (want12) *want12 = *want12 || q.want_wb12;
        plans->push_back(q);
        if (!q.eligible) {  // all ranges or none
            *all = false;
            break;
        }
    }
    return true;
}

static void destroy_plans(std::vector<FastPlan> *plans) {
    for (FastPlan &q : *plans) fast_plan_destroy(&q);
    plans->clear();
}

static void adopt_plans(std::vector<FastPlan> *plans, FastPlan *fp) {
    *fp = (*plans)[0];
    fp->n_more = (uint32_t)plans->size() - 1;
    fp->more = fp->n_more ? new FastPlan[fp->n_more] : nullptr;
    for (uint32_t r = 0; r < fp->n_more; ++r) fp->more[r] = (*plans)[r + 1];
    plans->clear();
}

bool fast_plan_create(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, FastPlan *fp) {
    *fp = FastPlan();
    if (g.n_segs == 0 || g.n_paths == 0 || g.n_steps == 0) return true;
    if ((reinterpret_cast<uintptr_t>(g.steps) & 15u) != 0) return true;  // 16-byte step loads
    // One range while the graph fits 4096 windows of 8192 segments (32 M); beyond, ranges of equal
    // size, each a walk of the steps per call (64 M segments / 100 M steps: two walks).  So many
    // windows only for plans whose calls are tagged (k_scan then keeps one LDS table per window, not
    // two); when a range cannot be (FLATGFA_TAGGED=0, more split paths than pass 2 has bitsets
    // for), the ranges are cut again at 2048 windows.  FLATGFA_MAX_WINDOWS keeps the old cut-off (tests).
    //
    // A record's tag has nine bits: a k_scan workgroup can name 512 items (less the split paths).
    // When only that keeps a plan from being tagged -- 131 k paths or more on a graph too large for
    // the wave-per-path kernels -- the PATHS are walked in groups, each with plans of its own over
    // the same ranges: the first group's pass 2 stores its counts, the others' add theirs.  Every
    // step is still read once per range.  (FLATGFA_PATH_GROUPS=0: never; n: at least n groups, tests.)
    if (const char *off = getenv("FLATGFA_MAX_WINDOWS")) {
        const uint32_t wb = g.n_segs <= 1024u * 4096u ? 12u : 13u;
        if ((((uint64_t)g.n_segs + (1u << wb) - 1) >> wb) > strtoul(off, nullptr, 10)) return true;
    }
    uint32_t want_groups = 1;
    bool groups_ok = true;
    if (const char *e = getenv("FLATGFA_PATH_GROUPS")) {
        want_groups = (uint32_t)strtoul(e, nullptr, 10);
        groups_ok = want_groups != 0;
        want_groups = std::max(1u, std::min(64u, want_groups));
    }
    std::vector<uint32_t> busy;  // the paths that have steps
    for (uint32_t p = 0; p < g.n_paths; ++p)
        if (he[p] > hb[p]) busy.push_back(p);
    // (max_win, force_wb) by preference: 4096 windows of the graph's own size; the same with 4096-segment
    // windows when the split paths of a larger graph ask for them (their bitsets need the LDS that
    // 8192-segment windows take: a graph of 16 M segments walked by ninety paths of ten million
    // steps would otherwise have no tagged plan, nor -- its buckets beyond 2^30 records -- any);
    // then 2048 windows, the untagged plans' limit.
    struct Try {
        uint32_t max_win, force_wb;
    };
    std::vector<Try> tries{{kMaxWinTagged, 0u}};
    bool tried_wb12 = false;
    for (size_t ti = 0; ti < tries.size(); ++ti) {
        const uint32_t max_win = tries[ti].max_win, force_wb = tries[ti].force_wb;
        std::vector<FastPlan> plans;
        bool all = true, many = false, want12 = false;
        if (want_groups == 1) {
            if (!append_ranges(g, hb, he, max_win, force_wb, &plans, &all, &many, &want12)) {
                destroy_plans(&plans);
                return false;
            }
            bool all_tagged = all;
            for (const FastPlan &q : plans) all_tagged = all_tagged && q.tagged;
            if (want12 && !all_tagged && !force_wb && !tried_wb12 && !getenv("FLATGFA_WB") && !getenv("FLATGFA_RANGE_SEGS")) {
                tried_wb12 = true;
                tries.push_back({kMaxWinTagged, 12u});  // next; if that does not yield a tagged plan either, this try comes again
                tries.push_back({max_win, 0u});
                destroy_plans(&plans);
                continue;
            }
            if (force_wb && !all_tagged && !many) {  // (the smaller windows were for tags' sake only)
                destroy_plans(&plans);
                continue;
            }
            if (all && !(many && groups_ok)) {
                adopt_plans(&plans, fp);
                return true;
            }
        } else {
            many = true;
        }
        if (many && groups_ok && busy.size() >= 2) {
            // an even share of the paths per group; more groups while some group still has too many items (long paths are cut into pieces)
            uint32_t hint_items = (uint32_t)busy.size(), hint_slots = 256;
            if (!plans.empty() && plans[0].n_slots) {
                hint_slots = plans[0].n_slots;
                hint_items = std::max<uint32_t>(hint_items, plans[0].n_items);
            }
            destroy_plans(&plans);
            uint32_t n_groups = std::max<uint32_t>(want_groups, (uint32_t)((hint_items + 384ull * hint_slots - 1) / (384ull * hint_slots)));
            n_groups = std::max(n_groups, 2u);
            for (; n_groups <= 64 && n_groups <= busy.size(); n_groups *= 2) {
                bool g_all = true, g_many = false, hip_ok = true;
                std::vector<uint32_t> hbk(g.n_paths), hek(g.n_paths);
                for (uint32_t k = 0; k < n_groups && g_all && hip_ok; ++k) {
                    const size_t lo = busy.size() * k / n_groups, hi = busy.size() * (k + 1) / n_groups;
                    for (uint32_t p = 0; p < g.n_paths; ++p) hbk[p] = hek[p] = hb[p];  // (a path outside the group: no steps)
                    for (size_t i = lo; i < hi; ++i) hek[busy[i]] = he[busy[i]];
                    const size_t first = plans.size();
                    hip_ok = append_ranges(g, hbk.data(), hek.data(), max_win, force_wb, &plans, &g_all, &g_many, nullptr, n_groups);
                    for (size_t i = first; i < plans.size(); ++i) {
                        plans[i].accumulate = k > 0;
                        g_all = g_all && plans[i].tagged;  // (a group is only worth it tagged)
                    }
                }
                if (!hip_ok) {
                    destroy_plans(&plans);
                    return false;
                }
                if (g_all && !plans.empty()) {
                    adopt_plans(&plans, fp);
                    fp->n_groups = n_groups;
                    return true;
                }
                destroy_plans(&plans);
                if (!g_many) break;  // (something else stands in the way)
            }
            if (force_wb) continue;  // (the smaller windows were for tags' sake only)
            // no luck: the plan the whole path set gets
            all = true;
            many = false;
            if (!append_ranges(g, hb, he, max_win, force_wb, &plans, &all, &many)) {
                destroy_plans(&plans);
                return false;
            }
            if (all) {
                adopt_plans(&plans, fp);
                return true;
            }
        }
        destroy_plans(&plans);
        if (ti + 1 == tries.size() && max_win == kMaxWinTagged && !force_wb) {
            const uint64_t max_range = getenv("FLATGFA_RANGE_SEGS") ? 0 : (uint64_t)max_win << 13;
            const uint64_t n_ranges = max_range ? (g.n_segs + max_range - 1) / max_range : 1;
            if (((uint64_t)g.n_segs + 8191) / 8192 > kMaxWin * n_ranges) tries.push_back({kMaxWin, 0u});  // (else the smaller cut-off would make the same ranges)
        }
    }
    return true;
}

// Scratch for path sums riding on seg_depth: one {sum len, sum depth * len} per (window, item).
// False when that would be out of proportion (then the caller walks the steps a second time).
bool fast_plan_want_path_sums(FastPlan *fp) {
    if (fp->packed) return false;  // (its calls are tagged; the fused sums ride on the directory)
    if (!fp->eligible || fp->wb != 12 || fp->acc_parts > 1 || fp->n_more || fp->n_win > kMaxWin) return false;  // (the fused form needs a window's final depth in one workgroup, and the directory: k_scan's untagged build)
    if (((uint64_t)fp->n_win + 1) * fp->n_slots * fp->cap >= (1ull << 30)) return false;  // (... which knows 32-bit bucket offsets only)
    if (fp->psum_part) return true;
    const uint64_t bytes = (uint64_t)fp->n_win * fp->dstride * 16;
    if (bytes > (256ull << 20)) return false;
    if (hipMalloc(&fp->psum_part, std::max<uint64_t>(bytes, 16)) != hipSuccess) {
        fp->psum_part = nullptr;
        (void)hipGetLastError();
        return false;
    }
    return true;
}

// After a call that ran out of sub-bucket room: four times the capacity, if that is possible (twice, ahead of need).
static bool grow_range(FastPlan *fp, uint32_t factor) {
    const uint32_t before = fp->cap;
    if (alloc_buckets(fp, (uint64_t)before * factor) <= 0) return false;
    return fp->cap > before;  // else the slot arithmetic allows no more
}

bool fast_plan_grow(FastPlan *fp, bool ahead_of_need) {
    if (!fp->eligible || fp->cap_forced) return false;
    if (fp->packed || std::any_of(fp->more, fp->more + fp->n_more, [](const FastPlan &q) { return q.packed; })) {
        // packed buckets have the room the counted call needed and no more: a call that makes other
        // records (the steps changed behind the plan) is completed through the atomic kernels
        if (!ahead_of_need) fp->eligible = false;
        return false;
    }
    const uint32_t factor = ahead_of_need ? 2u : 4u;  // (ahead of need: what was more than half full is then at most half full)
    // Ahead of need only the ranges (or path groups) whose own word says so are given more room: their
    // bucket arrays may be gigabytes each, and one hot window is no reason to double them all.
    const auto wants = [&](FastPlan *q) {
        if (!ahead_of_need || !q->taken) return true;
        uint32_t v = 0;
        if (hipMemcpy(&v, q->taken + q->n_slots, 4, hipMemcpyDeviceToHost) != hipSuccess) return true;
        if (v) (void)hipMemset(q->taken + q->n_slots, 0, 4);
        return v > (q->cap >> 1);
    };
    bool ok = wants(fp) ? grow_range(fp, factor) : true;
    for (uint32_t r = 0; r < fp->n_more && (ok || ahead_of_need); ++r)
        if (wants(&fp->more[r])) ok = grow_range(&fp->more[r], factor) && ok;  // (after an overflow: the status word does not say which range ran out)
    if (!ok && !ahead_of_need) fp->eligible = false;  // the atomic kernels take over
    return ok;
}

void fast_plan_destroy(FastPlan *fp) {
    for (uint32_t r = 0; r < fp->n_more; ++r) fast_plan_destroy(&fp->more[r]);
    delete[] fp->more;
    for (void *p : {(void *)fp->counts, (void *)fp->counts0, (void *)fp->buckets, (void *)fp->dir, (void *)fp->islot, (void *)fp->perm,
                    (void *)fp->elist, (void *)fp->wave_off, (void *)fp->fat_off, (void *)fp->fat_woff, (void *)fp->items, (void *)fp->short_items,
                    (void *)fp->medium_items, (void *)fp->tiny_items, (void *)fp->rev_steps, (void *)fp->work_counter, (void *)fp->other_ids, (void *)fp->psum_part,
                    (void *)fp->pair_part, (void *)fp->pair_flag, (void *)fp->taken, (void *)fp->pk_off, (void *)fp->pk_base, (void *)fp->pk})
        if (p) (void)hipFree(p);
    *fp = FastPlan();
}

// One range of the graph: the outputs are the range's own stretch of the result vectors.
static int run_range(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                     uint32_t *status, hipStream_t stream, const PathSums *ps, bool count_only) {
    if (ps && (uniq_out || fp.wb != 12 || !g.seg_len || !fp.psum_part || fp.n_range != g.n_segs)) {
        set_error("fast_seg_depth: path sums ride on seg_depth with 4096-segment windows only");
        return FLATGFA_ERR_ARG;
    }
    const uint32_t stride = fp.n_slots * fp.cap;
    const bool has_pre = fp.n_short || fp.n_medium || fp.n_tiny;
    // one persistent workgroup per CU; k_scan may be handed short paths back, so it gets a full grid when there are any
    // (and whenever the wave-per-path kernels ran: it saves their cursors for pass 2)
    // -- unless it has no items of its own and nothing can come back (the run counts of the lists are
    // exact): then the launch is left out, every record is one of the wave-per-path kernels', and a
    // path that does not fit after all (steps changed behind the plan) raises kStBackOverflow.
    const bool scan_skip = has_pre && fp.n_items == 0 && fp.exact_short && !fp.dbg && !getenv("FLATGFA_SCAN_ALWAYS");
    const uint32_t grid = scan_skip ? 0u : has_pre ? fp.n_slots : std::min<uint32_t>(fp.n_items, fp.n_slots);
    ScanArgs sa;
    sa.zero_a = sa.zero_b = nullptr;
    sa.zero_c = sa.zero_d = nullptr;
    sa.n_zero64 = 0;
    sa.path_begin = sa.path_end = nullptr;
    sa.rev_steps = nullptr;
    sa.n_fwd = ~0u;
    sa.steps = g.steps;
    sa.n_steps = g.n_steps;
    sa.items = reinterpret_cast<uint4 *>(fp.items);
    sa.short_items = reinterpret_cast<const uint4 *>(fp.short_items);
    sa.n_short = fp.n_short;
    sa.n_items = fp.n_items;
    sa.n_segs = fp.n_range;
    sa.seg_base = fp.seg_base;
    sa.n_total = g.n_segs;
    sa.ranged = (fp.seg_base != 0 || fp.n_range != g.n_segs) ? 1u : 0u;
    sa.n_win = fp.n_win;
    sa.n_slots = fp.n_slots;
    sa.wb = fp.wb;
    sa.nwp = fp.nwp;
    sa.has_pre = has_pre ? 1u : 0u;
    sa.max_back = scan_skip ? 0u : fp.max_back;
    sa.work_counter = fp.work_counter;
    sa.counts = fp.counts;
    sa.counts0 = fp.counts0;
    sa.buckets = fp.buckets;
    sa.dir = reinterpret_cast<uint2 *>(fp.dir);
    sa.islot = fp.islot;
    sa.perm = fp.perm;
    sa.dstride = fp.dstride;
    sa.cap = fp.cap;
    sa.stride = stride;
    sa.sink = fp.n_win * stride;
    sa.big = ((uint64_t)fp.n_win + 1) * stride >= (1ull << 30) ? 1u : 0u;
    sa.status = status;
    sa.dbg = fp.dbg;
    // Tagged: k_scan's records name their items, pass 2 walks whole sub-buckets.  Path sums ride on
    // the directory walk (an item's records have to be found again once the window's depth is final).
    const bool tagged = fp.tagged && !ps;
    sa.tagged = tagged ? 1u : 0u;
    sa.tag_limit = std::max(2u, fp.tag_limit);
    sa.taken = fp.taken;
    sa.mall_steps = fp.mall_steps;
    sa.pk_off = fp.pk_off;
    sa.pk_base = fp.pk_base;
    if (fp.packed && !tagged) { set_error("fast_seg_depth: a plan with packed buckets runs tagged calls only"); return FLATGFA_ERR_ARG; }
    sa.tprof = nullptr;
    if (getenv("FLATGFA_SCAN_TIME") && hipMalloc(&sa.tprof, kTprofRow * 8 * (size_t)fp.n_slots) == hipSuccess) (void)hipMemset(sa.tprof, 0, kTprofRow * 8 * (size_t)fp.n_slots);
    AccArgs aa{fp.n_range, fp.n_win, fp.n_slots, fp.cap, fp.counts, fp.counts0, scan_skip ? 2u : has_pre ? 1u : 0u, fp.buckets,
               reinterpret_cast<const uint2 *>(fp.dir), fp.islot, fp.dstride, fp.elist, fp.wave_off, fp.n_items,
               fp.work_counter, scan_skip ? 0u : fp.max_back, depth_out, uniq_out, status, fp.dbg,
               reinterpret_cast<const uint4 *>(fp.items), g.seg_len, ps ? reinterpret_cast<ulonglong2 *>(fp.psum_part) : nullptr,
               fp.fat_off, fp.fat_woff, fp.acc_parts, tagged ? fp.n_shared : 0u, nullptr, fp.pair_part, fp.pair_flag, fp.accumulate ? 1u : 0u, getenv("FLATGFA_NO_PLAIN") ? nullptr : fp.taken, fp.taken ? fp.taken + fp.n_slots : nullptr,
               fp.packed ? reinterpret_cast<const uint2 *>(fp.pk) : nullptr};  // (FLATGFA_NO_PLAIN: measurements)
    if (const char *sk = getenv("FLATGFA_ACC_SKIP")) aa.dbg = (uint32_t)strtoul(sk, nullptr, 10);  // (diagnostic: pass 2 without its revisit counts 128 / depth 256 / claims 64 / words behind the first 1024)
    const size_t tprof_words = (size_t)fp.n_win * fp.acc_parts * kAccWaves * 16;
    if (getenv("FLATGFA_ACC_TIME") && uniq_out && hipMalloc(&aa.tprof, tprof_words * 4) != hipSuccess) aa.tprof = nullptr;
    // The wave-per-path kernels: the paths read from the graph's steps, then those read from their
    // reversed copies (a handed-back one is walked by k_scan from the graph's own steps).
    if (fp.n_short && hipMemsetAsync(fp.work_counter, 0, 4, stream) != hipSuccess) return FLATGFA_ERR_HIP;
    if (fp.n_tiny) {  // paths a wave holds whole (at most 128 steps)
        ScanArgs sk = sa;
        sk.short_items = reinterpret_cast<const uint4 *>(fp.tiny_items);
        sk.n_short = fp.n_tiny;
        const uint32_t kgrid = std::min<uint32_t>((fp.n_tiny + kWaves - 1) / kWaves, fp.n_slots);
        ProfScope pscope(uniq_out ? "k_scan_tiny<uniq>" : "k_scan_tiny<depth>", stream);
        if (uniq_out) hipLaunchKernelGGL(k_scan_tiny<true>, dim3(kgrid), dim3(kThreads), fp.lds_bytes_tiny, stream, sk);
        else hipLaunchKernelGGL(k_scan_tiny<false>, dim3(kgrid), dim3(kThreads), fp.lds_bytes_tiny, stream, sk);
    }
    for (int medium = 0; medium < 2; ++medium) {
        const uint32_t n_all = medium ? fp.n_medium : fp.n_short, n_rev = medium ? fp.n_medium_rev : fp.n_short_rev;
        const uint4 *list = reinterpret_cast<const uint4 *>(medium ? fp.medium_items : fp.short_items);
        {
            const uint32_t n = n_all;
            if (!n) continue;
            ScanArgs sk = sa;
            sk.short_items = list;
            sk.n_short = n;
            sk.n_fwd = n_all - n_rev;  // (the reversed ones lie behind the others in the list)
            sk.rev_steps = fp.rev_steps;
            sk.path_begin = g.path_begin;
            sk.path_end = g.path_end;
            const uint32_t per_wg = medium ? (kMediumPaired ? kMediumWaves / 2 : kMediumWaves) : kShortWaves;  // paths a workgroup walks at a time
            const uint32_t kgrid = std::min<uint32_t>((n + per_wg - 1) / per_wg, fp.n_slots);
            ProfScope pscope(medium ? (uniq_out ? "k_scan_medium<uniq>" : "k_scan_medium<depth>") : (uniq_out ? "k_scan_short<uniq>" : "k_scan_short<depth>"), stream);
            if (medium) {
                if (uniq_out) hipLaunchKernelGGL(k_walk_medium<true>, dim3(kgrid), dim3(kMediumWaves * 64), fp.lds_bytes_medium, stream, sk);
                else hipLaunchKernelGGL(k_walk_medium<false>, dim3(kgrid), dim3(kMediumWaves * 64), fp.lds_bytes_medium, stream, sk);
            } else {
                if (uniq_out) hipLaunchKernelGGL(k_walk_short<true>, dim3(kgrid), dim3(kShortWaves * 64), fp.lds_bytes_short, stream, sk);
                else hipLaunchKernelGGL(k_walk_short<false>, dim3(kgrid), dim3(kShortWaves * 64), fp.lds_bytes_short, stream, sk);
            }
        }
    }
    if (ps && ps->clear) {  // the sums k_path_reduce adds to start at zero: k_scan's first act, or two memsets when it does not run
        if (grid) {
            sa.zero_c = (unsigned long long *)ps->len_out;
            sa.zero_d = (unsigned long long *)ps->weighted_out;
            sa.n_zero64 = g.n_paths;
        } else {
            ProfScope pscope("memset_path_sums", stream);
            if (hipMemsetAsync(ps->len_out, 0, (size_t)g.n_paths * 8, stream) != hipSuccess) return FLATGFA_ERR_HIP;
            if (hipMemsetAsync(ps->weighted_out, 0, (size_t)g.n_paths * 8, stream) != hipSuccess) return FLATGFA_ERR_HIP;
        }
    }
    if (grid) {
        if (fp.acc_parts > 1 && !fp.accumulate && !count_only) {  // (a later group of paths adds to what is there)
            sa.zero_a = depth_out;
            sa.zero_b = uniq_out;
        }
        ProfScope pscope(fp.dense ? "k_scan_dense" : "k_scan", stream);
        if (fp.dense) hipLaunchKernelGGL(k_scan_dense, dim3(grid), dim3(kThreads), dense_lds_bytes(fp.nwp), stream, sa);
        else if (fp.dbg) hipLaunchKernelGGL((k_scan<kModeDbg, false>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
        else if (fp.packed && sa.ranged) hipLaunchKernelGGL((k_scan<kModePackedRanged, true>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
        else if (fp.packed) hipLaunchKernelGGL((k_scan<kModePacked, true>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
        else if (sa.big && !tagged) { set_error("fast_seg_depth: a bucket array this large needs a tagged call"); return FLATGFA_ERR_ARG; }
        else if (sa.big && sa.ranged) hipLaunchKernelGGL((k_scan<kModeRangedBig, true>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
        else if (sa.big) hipLaunchKernelGGL((k_scan<kModeBig, true>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
        else if (sa.ranged && tagged) hipLaunchKernelGGL((k_scan<kModeRanged, true>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
        else if (sa.ranged) hipLaunchKernelGGL((k_scan<kModeRanged, false>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
        else if (tagged) hipLaunchKernelGGL((k_scan<kModePlain, true>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
        else hipLaunchKernelGGL((k_scan<kModePlain, false>), dim3(grid), dim3(kThreads), fp.lds_bytes_scan, stream, sa);
    }
    if (count_only) return hipGetLastError() == hipSuccess ? FLATGFA_OK : FLATGFA_ERR_HIP;  // (the counting call of a packed plan: pass 1 alone)
    if (fp.acc_parts > 1 && !grid && !fp.accumulate) {  // the window's workgroups add to the outputs: cleared by k_scan, or here when it does not run
        ProfScope pscope("memset_outputs", stream);
        if (hipMemsetAsync(depth_out, 0, (size_t)fp.n_range * 4, stream) != hipSuccess) return FLATGFA_ERR_HIP;
        if (uniq_out && hipMemsetAsync(uniq_out, 0, (size_t)fp.n_range * 4, stream) != hipSuccess) return FLATGFA_ERR_HIP;
    }
    {
        ProfScope pscope(uniq_out ? "k_accum<uniq>" : (ps ? "k_accum<depth+paths>" : "k_accum<depth>"), stream);
        const dim3 agrid(fp.n_win, fp.acc_parts);
        const bool pair = uniq_out && tagged && fp.acc_pair;
        if (pair) {
            aa.parts = 2;
            const uint32_t tl = tagged_lds_bytes(fp.wb, 0);
            const dim3 pgrid(fp.n_win, 2);
            if (fp.dense) hipLaunchKernelGGL((k_accum_pair<12, true>), pgrid, dim3(kAccThreads), tl, stream, aa);
            else hipLaunchKernelGGL((k_accum_pair<12, false>), pgrid, dim3(kAccThreads), tl, stream, aa);
        } else if (uniq_out && tagged) {
            const uint32_t tl = tagged_lds_bytes(fp.wb, fp.n_shared);
            if (fp.dense && fp.wb == 12) hipLaunchKernelGGL((k_accum<true, 12, false, true, false, true>), agrid, dim3(kAccThreads), tl, stream, aa);
            else if (fp.dense && fp.wb == 13) hipLaunchKernelGGL((k_accum<true, 13, false, true, false, true>), agrid, dim3(kAccThreads), tl, stream, aa);
            else if (fp.wb == 11 && getenv("FLATGFA_ACC_SMALL")) hipLaunchKernelGGL((k_accum_small<11>), agrid, dim3(kAccThreads), tl, stream, aa);
            else if (fp.wb == 11) hipLaunchKernelGGL((k_accum<true, 11, false, false, false, true>), agrid, dim3(kAccThreads), tl, stream, aa);
            else if (fp.wb == 12 && fp.acc_slots == 8 && fp.n_shared <= 64) hipLaunchKernelGGL((k_accum<true, 12, false, false, false, true, 8>), agrid, dim3(kAccThreads), tagged_lds_bytes(12, fp.n_shared, 8), stream, aa);
            else if (fp.wb == 12) hipLaunchKernelGGL((k_accum<true, 12, false, false, false, true>), agrid, dim3(kAccThreads), tl, stream, aa);
            else hipLaunchKernelGGL((k_accum<true, 13, false, false, false, true>), agrid, dim3(kAccThreads), tl, stream, aa);
        } else if (uniq_out) {
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
            else if (fp.wb == 12 && ps) hipLaunchKernelGGL((k_accum<false, 12, true>), agrid, dim3(kAccThreads), 0, stream, aa);
            else if (fp.wb == 12) hipLaunchKernelGGL((k_accum<false, 12>), agrid, dim3(kAccThreads), 0, stream, aa);
            else hipLaunchKernelGGL((k_accum<false, 13>), agrid, dim3(kAccThreads), 0, stream, aa);
        }
    }
    if (ps && fp.n_items) {
        ProfScope pscope("k_path_reduce", stream);
        hipLaunchKernelGGL(k_path_reduce, dim3((fp.n_items + 3) / 4), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(fp.items),
                           fp.elist, fp.n_items, fp.n_win, fp.dstride, reinterpret_cast<const ulonglong2 *>(fp.psum_part),
                           (unsigned long long *)ps->len_out, (unsigned long long *)ps->weighted_out);
    }
    if (hipGetLastError() != hipSuccess) {
        set_error("fast_seg_depth: kernel launch failed");
        return FLATGFA_ERR_HIP;
    }
    if (sa.tprof) {  // diagnostic: when the workgroups of k_scan start, when their first wave runs out of work, when they end
        constexpr size_t kRow = kTprofRow;
        std::vector<unsigned long long> raw(kRow * (size_t)fp.n_slots);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(raw.data(), sa.tprof, raw.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(sa.tprof);
        unsigned long long t0 = ~0ull;
        for (uint32_t i = 0; i < grid; ++i) t0 = std::min(t0, raw[kRow * i]);
        std::vector<double> st, fw, lw, en;
        for (uint32_t i = 0; i < grid; ++i) {
            st.push_back((raw[kRow * i] - t0) / 100.0);
            en.push_back((raw[kRow * i + 1] - t0) / 100.0);
            unsigned long long a = ~0ull, b = 0;
            for (size_t k = 0; k < kWaves; ++k) {
                a = std::min(a, raw[kRow * i + 4 + k]);
                b = std::max(b, raw[kRow * i + 4 + k]);
            }
            fw.push_back((a - t0) / 100.0);
            lw.push_back((b - t0) / 100.0);
        }
        const auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        fprintf(stderr, "k_scan%s workgroups (us since the first one started): start p50 %.1f max %.1f | first wave out of work p5 %.1f p50 %.1f p95 %.1f | last wave p5 %.1f p50 %.1f p95 %.1f max %.1f | end p50 %.1f max %.1f\n",
                tagged ? " [tagged]" : "", pct(st, 0.5), pct(st, 1.0), pct(fw, 0.05), pct(fw, 0.5), pct(fw, 0.95), pct(lw, 0.05), pct(lw, 0.5), pct(lw, 0.95), pct(lw, 1.0), pct(en, 0.5), pct(en, 1.0));
        // FLATGFA_SCAN_TIME=<file>: one line per workgroup and call -- call, workgroup, XCC, HW_ID, items, start, first / last wave out of work, end (us)
        static int call_no = 0;
        const char *where = getenv("FLATGFA_SCAN_TIME");
        if (where && strchr(where, '/')) {
            if (FILE *f = fopen(where, "a")) {
                for (uint32_t i = 0; i < grid; ++i)
                    fprintf(f, "%d,%u,%u,0x%08x,%u,%.2f,%.2f,%.2f,%.2f\n", call_no, i, (unsigned)(raw[kRow * i + 2] >> 32) & 15u, (unsigned)raw[kRow * i + 2],
                            (unsigned)raw[kRow * i + 3], st[i], fw[i], lw[i], en[i]);
                fclose(f);
            }
        }
        call_no += 1;
    }
    if (aa.tprof) {  // diagnostic: where the waves of pass 2 spend their time
        std::vector<uint32_t> raw(tprof_words);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(raw.data(), aa.tprof, tprof_words * 4, hipMemcpyDeviceToHost);
        (void)hipFree(aa.tprof);
        const size_t waves = tprof_words / 16;
        double sum[16] = {}, mx[16] = {};
        for (size_t w = 0; w < waves; ++w)
            for (int k = 0; k < 16; ++k) {
                sum[k] += raw[w * 16 + k];
                mx[k] = std::max<double>(mx[k], raw[w * 16 + k]);
            }
        static const char *names[5] = {"setup", "flat", "walk", "barrier", "scan+store"};
        fprintf(stderr, "k_accum%s per wave, us avg (max):", tagged ? " [tagged]" : "");
        for (int k = 0; k < 5; ++k) fprintf(stderr, "  %s %.2f (%.2f)", names[k], sum[k] / waves / 100.0, mx[k] / 100.0);
        fprintf(stderr, "  | per wave avg (max): steps %.1f (%.0f) rounds %.1f flushes %.1f (%.0f)\n", sum[8] / waves, mx[8], sum[9] / waves, sum[10] / waves, mx[10]);
    }
    if (fp.dbg & kDbgTime) {  // diagnostic: where the waves of k_scan spend their cycles
        unsigned long long acc[8] = {};
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(acc, status + 8, sizeof acc, hipMemcpyDeviceToHost);
        (void)hipMemset(status + 8, 0, sizeof acc);
        const double waves = (double)grid * kWaves;
        fprintf(stderr, "k_scan cycles per wave: wait_block %.0f  epoch_wait %.0f  passA+B %.0f  drain %.0f  other %.0f  item switch %.0f\n",
                acc[0] / waves, acc[1] / waves, acc[2] / waves, acc[3] / waves, acc[4] / waves, acc[5] / waves);
    }
    return FLATGFA_OK;
}

int fast_seg_depth(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                   uint32_t *status, hipStream_t stream, const PathSums *ps) {
    int rc = run_range(fp, g, depth_out, uniq_out, status, stream, ps, false);
    for (uint32_t r = 0; r < fp.n_more && rc == FLATGFA_OK; ++r) {
        const FastPlan &q = fp.more[r];
        rc = run_range(q, g, depth_out + q.seg_base, uniq_out ? uniq_out + q.seg_base : nullptr, status, stream, nullptr, false);
    }
    return rc;
}

}  // namespace fgfa_dev
