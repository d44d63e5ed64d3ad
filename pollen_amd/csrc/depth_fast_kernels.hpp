// Internal to the bucketed node-depth path: what its translation units share -- the constants, the kernels'
// argument blocks, the wave-level helpers and the streaming loads into pinned landing registers.
//   depth_scan.hip        pass 1: k_scan (runs of long items), k_scan_dense (ids without runs)
//   depth_scan_paths.hip  pass 1 for short paths: k_scan_short / k_scan_medium / k_scan_tiny (a wave per path)
//   depth_accum.hip       pass 2: k_accum, k_path_reduce
//   depth_fast.hip        the plan (which kernel walks which path, the scratch) and the launches
// DESIGN.md section 3 is the long version.  Constants, inline device functions and plain structs only (the same in every
// translation unit that includes it).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "depth_fast.hpp"
#include "device_common.hpp"

namespace fgfa_dev {


constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kMaxSlots = 1024;  // sub-buckets per window (= workgroups of pass 1) k_accum can stage
constexpr int kAccThreads = 1024;
constexpr uint32_t kAccWaves = kAccThreads / 64;
constexpr uint32_t kLdsLimit = 160 * 1024;
// status word bits (flatgfa_dev_status)
constexpr uint32_t kStBounds = 1u, kStDebug = 2u, kStOverflow = 4u, kStInternal = 8u, kStBackOverflow = 16u, kStStale = 32u;  // (32: FLATGFA_CHECK_NO_CLAIM=1 found the step values changed behind the plan)  // (16: more short paths handed back than the list holds: larger buckets would not help, the atomic kernels complete the call)  // (8: an invariant between the two passes did not hold -- a bug, reported as an error rather than as counts)

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
//   shared tag   = kTagCount - 2 - (ordinal of a path that is cut into pieces): the pieces are walked
//                  by different workgroups, their records lie in many sub-buckets, and all waves of
//                  pass 2 claim in ONE bitset per such path (LDS ORs are atomic across waves).
//   no-claim tag = kTagNoClaim, the highest: an item whose path walks the segment ids strictly upwards (or strictly
//                  downwards) from its first step to its last -- a haplotype walk without inversions or
//                  repeats -- never meets a segment twice, so the `seen` test of depth.rs:30-34 is always
//                  true for it: its records count for depth and for unique depth alike, pass 2 applies them
//                  without a bitset, and the item needs neither a private slot nor a shared one.  The plan
//                  finds these items when it is made (k_item_dirs; items[].z bit 31).
constexpr uint32_t kTagShift = 23, kTagCount = 1u << (32 - kTagShift);
constexpr uint32_t kTagNoClaim = kTagCount - 1u;
constexpr uint32_t kItemNoClaim = 0x80000000u;  // items[].z: the item's path is strictly monotone in the segment ids
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
    uint32_t mono_lo, mono_n;   // the wave-per-path kernels: the paths [mono_lo, mono_lo + mono_n) of the list never meet a segment twice (the plan looked): no claims
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
    const uint32_t *cflags;  // tagged: bit (first step / 16) of a block -- its records need no claim (FastPlan::cflags); nullptr: the items' own flags only
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
// The same with a uniform bit F on top: every id comes out as (F << 31) | id, one v_alignbit_b32 where take_block has a v_lshrrev_b32
// (k_scan: the plan says of every block whether its records need pass 2's claim, and the bit rides in the ids to the records' tags).
#define FGFA_TAKE16F(F, R0, R1, R2, R3, R4, R5, R6, R7, R8, R9, R10, R11, R12, R13, R14, R15)                           \
    asm volatile("v_alignbit_b32 %0, %16, " R0 ", 1\n\tv_alignbit_b32 %1, %16, " R1 ", 1\n\tv_alignbit_b32 %2, %16, " R2 ", 1"      \
                 "\n\tv_alignbit_b32 %3, %16, " R3 ", 1\n\tv_alignbit_b32 %4, %16, " R4 ", 1\n\tv_alignbit_b32 %5, %16, " R5 ", 1"  \
                 "\n\tv_alignbit_b32 %6, %16, " R6 ", 1\n\tv_alignbit_b32 %7, %16, " R7 ", 1\n\tv_alignbit_b32 %8, %16, " R8 ", 1"  \
                 "\n\tv_alignbit_b32 %9, %16, " R9 ", 1\n\tv_alignbit_b32 %10, %16, " R10 ", 1\n\tv_alignbit_b32 %11, %16, " R11 ", 1" \
                 "\n\tv_alignbit_b32 %12, %16, " R12 ", 1\n\tv_alignbit_b32 %13, %16, " R13 ", 1\n\tv_alignbit_b32 %14, %16, " R14 ", 1" \
                 "\n\tv_alignbit_b32 %15, %16, " R15 ", 1"                                                              \
                 : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]),       \
                   "=v"(a[8]), "=v"(a[9]), "=v"(a[10]), "=v"(a[11]), "=v"(a[12]), "=v"(a[13]), "=v"(a[14]), "=v"(a[15])  \
                 : "s"(F)                                                                                               \
                 : "memory")
template <int SET>
__device__ __forceinline__ void take_block_flagged(uint32_t (&a)[16], uint32_t f) {
    if (SET == 0) FGFA_TAKE16F(f, "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
    else if (SET == 2) FGFA_TAKE16F(f, "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
    else FGFA_TAKE16F(f, "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}
template <int SET>
__device__ __forceinline__ void take_block(uint32_t (&a)[16]) {
    if (SET == 0) FGFA_TAKE16("v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
    else if (SET == 2) FGFA_TAKE16("v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
    else FGFA_TAKE16("v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
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

// LDS control words of k_scan, behind the two cursor tables: the next block of the current /
// next item nobody has taken yet (two cells, by item parity), how many waves have left the item
// (two cells), and how many items are complete.
// k_scan's builds: plain, diagnostic (FLATGFA_DEBUG_SKIP), ranged (one of several walks of a graph beyond 16 M segments)
constexpr int kModePlain = 0, kModeDbg = 1, kModeRanged = 2, kModeBig = 3, kModeRangedBig = 4, kModePacked = 5, kModePackedRanged = 6, kModePlainFlags = 7, kModePackedFlags = 8, kModePlainNarrow = 9;  // (...Flags: the plain and the packed build for plans with per-block no-claim flags, ScanArgs::cflags)  // (Big: builds of their own for bucket arrays of 2^30 records or more, see put(); Packed: sub-buckets of exactly the size their records need, items dealt in a fixed order)
constexpr bool mode_ranged(int m) { return m == kModeRanged || m == kModeRangedBig || m == kModePackedRanged; }
constexpr bool mode_big(int m) { return m == kModeBig || m == kModeRangedBig; }
constexpr bool mode_packed(int m) { return m == kModePacked || m == kModePackedRanged || m == kModePackedFlags; }
constexpr bool mode_flags(int m) { return m == kModePlainFlags || m == kModePackedFlags; }
constexpr bool mode_plain(int m) { return m == kModePlain || m == kModePlainNarrow; }
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
// ... but two where runs are long (kModePlainNarrow: the plain build for plans that count fewer than a record for eight steps -- the
// benchmark's walks make one for ten): a block then rarely leaves four chunks' worth behind, and what it leaves goes out sooner
// (same box, cfg-L: k_scan 90.3 / 90.0 / 89.5 -> 88.1 / 88.7 / 88.7 us warm, 102.5 -> 101.7 and 93.8 -> 91.3 with every step from
// HBM; paths along the graph, three records for ten steps: 118.8 -> 120.7 the other way, so they keep four: NOTES R6.5)
constexpr int mode_wide(int m) { return m == kModePlainNarrow ? 2 : kWide; }
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
    uint32_t n_shared;  // tagged calls: the split paths, whose bitsets all waves of the workgroup share (tags kTagCount - 1 - n_shared .. kTagCount - 2)
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

// ---- the wave-per-path kernels' queues and tables (the plan sorts the paths by what fits them) ----
#ifndef FGFA_QCAP
#define FGFA_QCAP 416
#endif
constexpr uint32_t kQCap = FGFA_QCAP;  // at least 64 left over + up to 257 from sixteen lanes of a block; a short path has at most kQCap - 16 runs, hence bitset words: its 512-entry hash set must not fill up
constexpr uint32_t kPCap = 96;   // parked claims (two words each): 31 left over + up to 64 from one chunk
constexpr uint32_t kTinyMax = 128;   // steps
#ifndef FGFA_TINY_BITS
#define FGFA_TINY_BITS 8
#endif
constexpr uint32_t kTinyBits = FGFA_TINY_BITS, kTinyTab = 1u << kTinyBits;   // entries of a wave's id set (at most half full)
constexpr uint32_t kTinyQueue = 64 + kTinyMax;  // a wave's record queue: what is left over + one path of all starts
// ---- dynamic LDS of the kernels (host side) ----
// the "seen" bitsets of a tagged call: (kAccWaves * slots + n_shared) * window / 8 bytes
#ifndef FGFA_DENSE_TILE
#define FGFA_DENSE_TILE 8192
#endif
constexpr uint32_t kDenseTile = FGFA_DENSE_TILE;
constexpr int kDensePer = kDenseTile / kThreads;  // steps per thread and tile
// (a plan of at most kMaxWin windows may run either build of k_scan: sized for the untagged one)
inline uint32_t scan_lds_bytes(uint32_t nwp, bool tagged_only = false, bool packed = false) { return ((tagged_only && !packed ? 1u : 2u) * nwp + kCtlWords + kWaves * (packed ? kQPacked : kQ2) * 2u) * 4u; }
inline uint32_t tagged_lds_bytes(uint32_t wb, uint32_t n_shared, uint32_t slots = kTagSlots) { return (kAccWaves * slots + n_shared) * ((1u << wb) / 8u); }
inline uint32_t dense_lds_bytes(uint32_t nwp) { return (6u * nwp + 64u + kDenseTile + 64u) * 4u; }  // (behind the stage: a sink, the totals of two tiles)

// ---- the kernels' launchers: one per translation unit that holds kernels ----
// (each sets its kernels' dynamic-LDS attribute once per DEVICE -- OncePerDevice above: the attribute belongs to the kernel, not to a plan)
bool scan_kernels_setup();        // depth_scan.hip
bool path_kernels_setup();        // depth_scan_paths.hip
bool accum_kernels_setup();       // depth_accum.hip
// pass 1 over the plan's long items: k_scan in the build the plan needs (tagged / ranged / packed / big), or k_scan_dense
int launch_scan(const FastPlan &fp, const ScanArgs &sa, bool tagged, uint32_t grid, hipStream_t stream);
// the wave-per-path kernels: paths a wave holds whole (k_scan_tiny), short and medium ones (k_scan_short's two builds)
void launch_scan_tiny(const FastPlan &fp, const ScanArgs &sk, bool uniq, uint32_t grid, hipStream_t stream);
void launch_scan_short(const FastPlan &fp, const ScanArgs &sk, bool medium, bool uniq, uint32_t grid, hipStream_t stream);
// pass 2 in the build the call needs (unique depth or not, tagged walk or directory walk, path sums, window size)
void launch_accum(const FastPlan &fp, AccArgs &aa, bool uniq, bool tagged, bool psum, hipStream_t stream);
void launch_path_reduce(const FastPlan &fp, unsigned long long *len_out, unsigned long long *weighted_out, hipStream_t stream);

}  // namespace fgfa_dev
