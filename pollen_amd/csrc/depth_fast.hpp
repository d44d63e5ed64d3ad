// The bucketed node-depth path (depth_fast.hip); see DESIGN.md "Kernels".
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <mutex>

#include "../../include/flatgfa.h"

namespace fgfa_dev {

// The runtime keeps a kernel's attributes (hipFuncSetAttribute: more than 64 KB of dynamic LDS) per DEVICE, and one process
// may drive several (flatgfa_sharded_create: a plan per shard on its own device and host thread; Python: graphs on cuda:1 and
// up).  `fn` runs once per device that is current when a plan is made there; devices beyond the mask's 64 run it every time.
struct OncePerDevice {
    std::mutex mu;
    uint64_t done = 0;
    template <class F>
    bool operator()(F fn) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        std::lock_guard<std::mutex> lk(mu);
        if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return true;
        if (!fn()) return false;
        if (dev >= 0 && dev < 64) done |= 1ull << dev;
        return true;
    }
};

// multiProcessorCount of a device, asked for once per device (depth_device.hip)
int device_cu_count(int device);
// FLATGFA_TIMING=1: prints what the calling thread spent since its last tick (the device drained first), to stderr; `what` == nullptr restarts the clock
void plan_tick(const char *what);

struct FastPlan {
    bool eligible = false;
    bool cap_forced = false;   // FLATGFA_BUCKET_CAP (tests): the capacity must not grow
    // A plan covers the segments [seg_base, seg_base + n_range).  A graph beyond 16 M segments gets
    // several (`more`: a heap array of n_more further plans); each call then walks the steps once
    // per range, every walk keeping only the part of each run that falls into its range.
    uint32_t seg_base = 0, n_range = 0;
    FastPlan *more = nullptr;
    uint32_t n_more = 0;
    uint32_t n_cus = 256;
    bool tagged = false;       // k_scan's records can carry their item's tag: pass 2 then walks whole sub-buckets and needs no directory (depth_fast.hip: kTagShift)
    uint32_t tag_limit = 0;    // tagged: items a k_scan workgroup may take (its private tags; at most 512 less the split paths; FLATGFA_TAG_LIMIT shrinks it for tests)
    uint32_t n_shared = 0;     // paths cut into pieces (their bitsets are shared by all waves of a pass-2 workgroup in a tagged call)
    bool big_groups = false;   // pass 2 looks for steps that lie inside one item (worth it when a path has hundreds of records per window; the plan's creator times both)
    bool dense_maybe = false;  // between one and nine records for ten steps: the plan's creator times k_scan_dense against k_scan
    bool dense = false;        // nearly every step starts a run: pass 1 partitions the steps themselves (k_scan_dense)
    bool narrow_emit = false;  // fewer than a record for eight steps: the plain tagged build of k_scan that emits two chunks side by side, not four (kModePlainNarrow)
    uint32_t acc_parts = 1;  // workgroups per window in pass 2 (small graphs: fewer windows than CUs)
    uint64_t est_records = 0;  // records k_scan will make of its items (counted when the plan is made)
    bool acc_pair = false;     // tagged calls with unique depth run two workgroups per window, both resident on a CU (k_accum_pair)
    uint32_t acc_slots = 4;    // private bitsets per wave of the tagged walk
    bool acc_own = false;      // ... which keeps track of their owners (k_accum<..., OWN>): sub-buckets of sparse tags
    uint32_t *pair_part = nullptr, *pair_flag = nullptr;  // their halves of the result vectors, and how many are there
    // Packed buckets: every (window, workgroup) sub-bucket has exactly the room its records need (counted once when the
    // plan is made, k_scan dealing its items in a fixed order from then on), a workgroup's sub-buckets back to back.
    uint64_t mall_steps = 0;       // the steps below this index are read without the nt hint: they stay in the Infinity Cache from call to call (plan_create decides)
    bool packed = false;
    bool can_pack = false;         // create_range: the plan could have packed buckets (tagged, all records k_scan's, ...): whether making it again for them can lead anywhere
    uint32_t *pk_off = nullptr;    // u32[n_slots][n_win + 1] sub-bucket starts within the workgroup's region; the last entry is its sink
    uint64_t *pk_base = nullptr;   // u64[n_slots] where each workgroup's region starts
    void *pk = nullptr;            // uint2[n_win][n_slots] {start in the array, room} for pass 2
    uint64_t bucket_records = 0;   // records the bucket array has room for (what flatgfa_dev_plan_describe reports as scratch)
    uint32_t n_slots = 0;      // sub-buckets per window = persistent workgroups of pass 1
    uint32_t n_win = 0;        // accumulation windows
    uint32_t wb = 12;          // log2 of the window size (4096 segments, 8192 beyond 4 M segments)
    uint32_t nwp = 0;          // n_win rounded up to a multiple of 64
    uint32_t cap = 0;          // records per (window, sub-bucket)
    uint32_t lds_bytes_scan = 0;
    uint32_t dbg = 0;          // FLATGFA_DEBUG_SKIP ablation mask (diagnostics only)
    uint32_t *counts = nullptr;    // u32[n_win * n_slots] cursors, zero between calls (pass 2 resets)
    uint32_t *counts0 = nullptr;   // u32[n_win * n_slots] the cursors k_scan started from
    uint32_t *buckets = nullptr;   // u32[(n_win + 1) * n_slots * cap]
    void *dir = nullptr;           // uint2[n_win * dstride] {cursor before, after} each item, per window, in pass 2's walk order
    uint32_t *islot = nullptr;     // u32[dstride] per position of pass 2's walk order: the sub-bucket that holds the item's records | first of its path << 31
    uint32_t *perm = nullptr;      // u32[n_items] where item j stands in pass 2's walk order | first of its path << 31
    uint32_t dstride = 0;          // n_items + max_back + 1
    uint32_t *lists_slab = nullptr;  // one allocation behind perm, elist, wave_off, fat_off and fat_woff
    uint32_t *elist = nullptr;     // k_scan's items in pass 2's order (grouped by path, split among its waves)
    uint32_t *wave_off = nullptr;  // u32[16 * acc_parts + 1] the stretch of elist each wave of pass 2 walks
    uint32_t *fat_off = nullptr;   // u32[acc_parts + 1] the long paths each of a window's workgroups walks with all its waves
    uint32_t *fat_woff = nullptr;  // u32[n_fat][17] the stretch of elist each wave walks of such a path's items
    uint32_t n_fat = 0;
    void *items = nullptr;         // uint4[n_items + max_back] whole paths and pieces of long paths, longest first,
                                   // with room for the short paths k_scan_short hands back
    uint32_t n_items = 0;
    uint32_t n_noclaim = 0;        // items whose path walks the segment ids strictly one way (their records skip pass 2's claim: depth_fast.hip kTagNoClaim)
    uint32_t *cflags = nullptr;    // a bit per 16 steps of the step array, of which the one of a k_scan block's FIRST chunk counts: every chunk of the block
                                   // lies in windows its path enters once and walks one way, and the block makes no-claim records too (k_visit_bits,
                                   // k_chunk_flags, k_block_flags); nullptr: none
    uint64_t n_flag_chunks = 0;    // how many chunks say so
    bool marks_wanted = false;     // create_range: this plan may gain from such marks (tagged, one range, items that claim) -- they are made off the
                                   // critical path (fast_marks_start on a side stream, fast_marks_finish installs them between two calls)
    uint64_t item_steps = 0;       // steps of k_scan's items (what the marks' share is measured against)
    uint32_t max_back = 0;
    bool accumulate = false;       // a group of paths behind the first (see fast_plan_create): pass 2 adds to the outputs
    bool too_many_items = false;   // create_range's verdict: only the number of items (or of split paths) per k_scan workgroup stands between this range and a tagged plan
    bool want_wb12 = false;        // ... or only the split paths' bitsets, for which pass 2 has LDS with 4096-segment windows
    uint32_t n_groups = 1;         // (of the first plan) path groups the ranges of `more` belong to
    bool exact_short = false;      // the wave-per-path lists hold only paths that fit: nothing is handed back to k_scan
    void *short_items = nullptr;   // uint4[n_short] paths every wave walks on its own (k_scan_short): first those read from the graph's
                                   // steps, then n_short_rev paths that walk the ids downwards, read from rev_steps (their
                                   // steps in reverse order: a wave-per-path kernel only knows runs that go up)
    uint32_t n_short = 0, n_short_rev = 0;
    // the paths of a wave-per-path list that walk the segment ids strictly one way -- no segment twice, so no claim --
    // lie at [mono_lo, mono_lo + mono_n) of their list, on both sides of the boundary to the reversed ones
    uint32_t short_mono_lo = 0, short_mono_n = 0, medium_mono_lo = 0, medium_mono_n = 0, tiny_mono_lo = 0, tiny_mono_n = 0;
    uint32_t *rev_steps = nullptr; // the reversed copies, every path at a multiple of 16
    uint32_t n_rev_steps = 0;
    uint32_t lds_bytes_short = 0;
    void *medium_items = nullptr;  // uint4[n_medium] longer paths with few enough runs for a 2048-entry hash set
    uint32_t n_medium = 0, n_medium_rev = 0;  // (laid out like short_items)
    uint32_t lds_bytes_medium = 0;
    void *tiny_items = nullptr;    // uint4[n_tiny] paths of at most 128 steps: a wave holds one whole (k_scan_tiny)
    uint32_t n_tiny = 0;
    uint64_t class_steps[4] = {0, 0, 0, 0};  // steps walked by k_scan / k_scan_short / k_scan_medium / k_scan_tiny (flatgfa_dev_plan_describe)
    uint32_t lds_bytes_tiny = 0;
    uint32_t *taken = nullptr;         // u32[n_slots] items each workgroup of the last tagged k_scan took
    uint32_t *work_counter = nullptr;  // how many short paths were handed back in this call
    void *psum_part = nullptr;         // ulonglong2[n_win * dstride] per-window path sums of k_scan's items (on first use)
    uint32_t *other_ids = nullptr;     // u32[n_other] the paths k_scan_short walks (their sums need k_path_sums)
    uint32_t n_other = 0;
};

// The per-block no-claim marks of a plan (k_visit_bits, k_chunk_flags, k_block_flags: three more reads of the steps) are
// not on the way to the first answer: a job enqueues them on a stream of its own; once they are there they are looked
// at on the host (do enough blocks qualify?) and installed between two calls.
struct MarksJob {
    bool active = false;
    uint32_t *vis = nullptr, *pbeg = nullptr, *chunks = nullptr, *flags = nullptr;
    unsigned long long *cnt = nullptr;
    hipEvent_t done = nullptr;
};
// Enqueues the job's kernels on `side` (which must not be a stream a call of the plan runs on).  `host_path_begin`: the
// spans the plan was made with (P entries, alive until the job is finished).  False on a HIP error (the plan then
// simply has no marks).  A plan that does not want marks (fp.marks_wanted) gets no job: job->active stays false.
bool fast_marks_start(const FastPlan &fp, const flatgfa_dev_graph_t &g, const uint32_t *host_path_begin, hipStream_t side, MarksJob *job);
// Has the job finished?  (never blocks)
bool fast_marks_ready(const MarksJob &job);
// Waits for the job, releases its scratch and installs the marks in `fp` if enough blocks qualify (no call of the plan
// may be enqueued concurrently from another thread; calls already running are not affected: their arguments were
// copied when they were launched).
void fast_marks_finish(FastPlan *fp, MarksJob *job);
// Debug (FLATGFA_CHECK_NO_CLAIM=1): re-derives, on `stream` and before a call's kernels, the facts about the step values
// that the plan relies on without re-checking them per call -- items and short paths found strictly monotone, the
// per-block marks, the reversed copies -- and raises status bit kStStale where one no longer holds.
int fast_check_plan_facts(const FastPlan &fp, const flatgfa_dev_graph_t &g, const uint32_t *host_path_begin, const uint32_t *host_path_end, uint32_t *status,
                          hipStream_t stream);

// Decides eligibility (16-byte aligned steps, directories that stay small next to the steps) and
// allocates the scratch: one range of at most 2048 windows, or several.  Returns false only on a HIP error.
// `scan_workgroups`: how many persistent workgroups pass 1 runs (= sub-buckets per window); 0 = one per CU.  A plan that
// is one lane of a pipeline (calls in flight) takes fewer, so that another call's kernels share the chip with its pass 1.
// `prefer_packed`: record buckets laid out to the count wherever the plan allows it (by default only where the even layout
// would take gigabytes from the start).
bool fast_plan_create(const flatgfa_dev_graph_t &g, const uint32_t *host_path_begin, const uint32_t *host_path_end,
                      FastPlan *fp, uint32_t scan_workgroups = 0, bool prefer_packed = false);
void fast_plan_destroy(FastPlan *fp);
// After a call whose records did not fit their sub-buckets (status bit 4): quadruple the capacity.
// Returns false -- and marks the plan ineligible -- when that is not possible.
// gives back the bucket arrays kept for the next plan (one per device)
void fast_release_scratch();
bool fast_plan_grow(FastPlan *fp, bool ahead_of_need = false);  // ahead_of_need: a plan that cannot grow stays eligible
// Per-path sums of measure_path (depth.rs:116-131), accumulated by pass 2 of a seg_depth call for
// the paths k_scan walks (plan.other_ids lists the rest): u64[n_paths] each, zeroed by the caller.
struct PathSums {
    uint64_t *len_out, *weighted_out;  // one per path of the graph
    bool clear = false;                // the call zeroes them first (inside k_scan where it runs)
};
// Allocates (once) the scratch `ps` needs; false = not possible for this plan.
bool fast_plan_want_path_sums(FastPlan *fp);
// Enqueues the kernels.  uniq_out may be NULL (seg_depth); `ps` only with uniq_out == NULL and
// after fast_plan_want_path_sums.
int fast_seg_depth(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                   uint32_t *status, hipStream_t stream, const PathSums *ps = nullptr);

}  // namespace fgfa_dev
