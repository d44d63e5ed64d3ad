// The bucketed two-kernel node-depth path (depth_fast.hip); see DESIGN.md "Kernels".
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "../../include/flatgfa.h"

namespace fgfa_dev {

struct FastPlan {
    bool eligible = false;
    uint32_t n_cus = 256;
    uint32_t n_slots = 0;      // sub-buckets per window = persistent workgroups of pass 1
    uint32_t n_win = 0;        // 4096-segment accumulation windows (of one segment-range pass)
    uint32_t n_pass = 1;       // segment-range passes: 1 when the whole bitset fits LDS
    uint32_t seg_range = 0;    // segments per pass
    uint32_t n_words = 0;      // 32-bit words of the per-path "seen" bitset (padded)
    uint32_t cap = 0;          // records per (window, sub-bucket)
    uint32_t lds_bytes_uniq = 0, lds_bytes_depth = 0;
    uint32_t dbg = 0;          // FLATGFA_DEBUG_SKIP ablation mask (diagnostics only)
    uint32_t *counts = nullptr;    // u32[n_win * n_slots], zero between calls (pass 2 resets)
    uint32_t *buckets = nullptr;   // u32[n_win * n_slots * cap]
    int *ovf_d = nullptr;          // int[n_segs + 1], zero between calls
    int *ovf_u = nullptr;
    uint32_t *ovf_flag = nullptr;  // u32[n_win]
    void *items = nullptr;         // uint4[n_items + n_short] whole paths and pieces of long paths, longest first,
                                   // with room for the short paths k_scan_short hands back
    uint32_t n_items = 0;
    void *short_items = nullptr;   // uint4[n_short] paths every wave walks on its own (k_scan_short)
    uint32_t n_short = 0;
    uint32_t lds_bytes_short = 0;
    void *medium_items = nullptr;  // uint4[n_medium] longer paths with few enough runs for a 4096-entry hash set
    uint32_t n_medium = 0;
    uint32_t lds_bytes_medium = 0;
    uint32_t *piece_bits = nullptr;  // bitsets left behind by the pieces of split paths
    void *split = nullptr;         // uint2[n_split] {first piece slot, pieces} per split path
    uint32_t n_split = 0;
    uint32_t *work_counter = nullptr;  // how many short paths were handed back in this call (pass 2 resets)
};

// Decides eligibility (the per-path bitset must fit one CU's 160 KiB LDS, steps must be
// 16-byte aligned) and allocates the scratch.  Returns false only on a HIP error.
bool fast_plan_create(const flatgfa_dev_graph_t &g, const uint32_t *host_path_begin, const uint32_t *host_path_end,
                      FastPlan *fp);
void fast_plan_destroy(FastPlan *fp);
// Enqueues the two kernels.  uniq_out may be NULL (seg_depth).
int fast_seg_depth(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                   uint32_t *status, hipStream_t stream);

}  // namespace fgfa_dev
