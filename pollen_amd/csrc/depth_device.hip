// HIP kernels + device-level C ABI for the FlatGFA depth queries on gfx950.
//
// Reference semantics (cucapra/pollen flatgfa/src/ops/depth.rs):
//   depth[s] = number of steps, over all paths, whose handle's segment is s   (:25-29, :48-53)
//   uniq[s]  = number of path entries that touch s at least once              (:30-34)
//   path sums: length = sum seg_len, weighted = sum depth*seg_len per path    (:116-131)
// All integer; results are order-independent sums, so they are bit-exact by construction.
//
// Data layout in HBM (structure of arrays): steps u32[N] (Handle bits), path_begin/path_end
// u32[P], seg_len u32[S].  See DESIGN.md for the kernel-by-kernel roofline accounting.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/flatgfa.h"
#include "depth_fast.hpp"
#include "temp_arena.hpp"
#include "device_common.hpp"
#include "prof.hpp"

namespace fgfa_dev {

thread_local std::string g_last_error;
void set_error(const std::string &s) { g_last_error = s; }
const char *last_error() { return g_last_error.c_str(); }

// ------------------------------------------------------------ profiling ---

static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

// events belong to the device that was current when they were created: a pool per device
static std::vector<std::vector<hipEvent_t>> g_event_pool;
static std::vector<hipEvent_t> &pool_of_current_device() {  // (g_prof_mu held)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if ((size_t)dev >= g_event_pool.size()) g_event_pool.resize((size_t)dev + 1);
    return g_event_pool[(size_t)dev];
}

bool prof_enabled() { return g_prof_on; }
hipEvent_t prof_event_get() {
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        std::vector<hipEvent_t> &pool = pool_of_current_device();
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
    }
    // (timing only: without the system-scope fence a default event adds to what it is recorded behind --
    // between k_scan and k_accum that is a write-back of the records pass 2 is about to read)
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventCreate(&e);
    }
    return e;
}
void prof_event_put(hipEvent_t e) {
    if (!e) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::vector<hipEvent_t> &pool = pool_of_current_device();
    if (pool.size() < 256) pool.push_back(e);
    else (void)hipEventDestroy(e);
}
void prof_push(const ProfRec &r) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
}

// -------------------------------------------------------------- kernels ---

// One work item = a contiguous piece [begin, end) of one path's step span.
struct WorkItem {
    uint32_t begin, end, path, flags;
};

constexpr int kScanThreads = 256;
constexpr uint32_t kScanPiece = 16384;  // steps per work item of the flat depth scan

// depth only (seg_depth, depth.rs:45-56): coalesced 4-byte step loads, one global atomic per step.
__global__ __launch_bounds__(kScanThreads) void k_depth_scan(const uint32_t *__restrict__ steps,
                                                              const WorkItem *__restrict__ items, uint32_t n_items,
                                                              uint32_t n_segs, uint32_t *__restrict__ depth,
                                                              uint32_t *__restrict__ status) {
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const WorkItem w = items[it];
        for (uint64_t i = (uint64_t)w.begin + threadIdx.x; i < w.end; i += kScanThreads) {
            uint32_t seg = steps[i] >> 1;
            if (seg < n_segs) atomicAdd(&depth[seg], 1u);
            else *status = 1u;
        }
    }
}

// depth + uniq, one workgroup per (path, segment window): the path's "seen" bitset of
// depth.rs:23-34 lives in LDS (one bit per segment of the window); a step whose bit was clear
// bumps uniq.  Window 0 also accumulates depth.
constexpr int kUniqThreads = 1024;
constexpr uint32_t kWinWords = 32768;           // 128 KiB of LDS = 1,048,576 segments per window
constexpr uint32_t kWinBits = kWinWords * 32u;

__global__ __launch_bounds__(kUniqThreads) void k_depth_uniq_path(const uint32_t *__restrict__ steps,
                                                                   const uint32_t *__restrict__ path_begin,
                                                                   const uint32_t *__restrict__ path_end,
                                                                   uint32_t n_paths, uint32_t n_segs,
                                                                   uint32_t n_windows, uint32_t *__restrict__ depth,
                                                                   uint32_t *__restrict__ uniq,
                                                                   uint32_t *__restrict__ status) {
    extern __shared__ uint32_t seen[];
    const uint64_t total = (uint64_t)n_paths * n_windows;
    for (uint64_t job = blockIdx.x; job < total; job += gridDim.x) {
        const uint32_t p = (uint32_t)(job / n_windows);
        const uint32_t win = (uint32_t)(job % n_windows);
        const uint32_t lo = win * kWinBits;
        const uint32_t nbits = min(kWinBits, n_segs - lo);
        const uint32_t nwords = (nbits + 31u) >> 5;
        for (uint32_t w = threadIdx.x; w < nwords; w += kUniqThreads) seen[w] = 0u;
        __syncthreads();
        const uint32_t b = path_begin[p], e = path_end[p];
        for (uint64_t i = (uint64_t)b + threadIdx.x; i < e; i += kUniqThreads) {
            const uint32_t seg = steps[i] >> 1;
            if (seg >= n_segs) {
                *status = 1u;
                continue;
            }
            if (win == 0) atomicAdd(&depth[seg], 1u);
            const uint32_t rel = seg - lo;  // wraps below the window; the compare rejects it
            if (rel < nbits) {
                const uint32_t bit = 1u << (rel & 31u);
                const uint32_t old = atomicOr(&seen[rel >> 5], bit);
                if (!(old & bit)) atomicAdd(&uniq[seg], 1u);
            }
        }
        __syncthreads();
    }
}

// measure_path (depth.rs:116-131): each block reduces one slice of one requested path.
constexpr int kSumThreads = 256;

// (seg_len, depth) side by side, so that a step costs one 8-byte gather instead of two 4-byte ones
// Clears one or two result vectors in one launch (two memsets cost 8 us, which counts when the
// whole query takes 25).
__global__ __launch_bounds__(256) void k_zero_outputs(uint32_t *__restrict__ a, uint32_t *__restrict__ b, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        a[i] = 0u;
        if (b) b[i] = 0u;
    }
}

__global__ __launch_bounds__(256) void k_pack_len_depth(const uint32_t *__restrict__ seg_len,
                                                        const uint32_t *__restrict__ depth, uint32_t n_segs,
                                                        uint2 *__restrict__ tab) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_segs; s += gridDim.x * blockDim.x)
        tab[s] = make_uint2(seg_len[s], depth[s]);
}

__global__ __launch_bounds__(kSumThreads) void k_path_sums(const uint32_t *__restrict__ steps,
                                                            const uint32_t *__restrict__ path_begin,
                                                            const uint32_t *__restrict__ path_end, uint32_t n_paths,
                                                            uint32_t n_segs, const uint2 *__restrict__ tab,
                                                            const uint32_t *__restrict__ path_ids, uint32_t n_ids,
                                                            uint32_t split, uint32_t by_path,
                                                            unsigned long long *__restrict__ length_out,
                                                            unsigned long long *__restrict__ weighted_out,
                                                            uint32_t *__restrict__ status) {
    __shared__ unsigned long long red[2][kSumThreads / 64];
    const uint64_t total = (uint64_t)n_ids * split;
    for (uint64_t job = blockIdx.x; job < total; job += gridDim.x) {
        const uint32_t k = (uint32_t)(job / split), part = (uint32_t)(job % split);
        const uint32_t p = path_ids[k];
        unsigned long long len = 0, wsum = 0;
        if (p < n_paths) {
            const uint64_t b = path_begin[p], e = path_end[p];
            const uint64_t n = e - b;
            const uint64_t lo = b + n * part / split, hi = b + n * (part + 1) / split;
            // Consecutive lanes take consecutive steps: along a run their table entries share lines.
            // Eight steps per thread are requested before any is used, then their eight table
            // entries: the loop is bound by memory latency, not by bandwidth.
            constexpr int kBatch = 8;
            bool bad = false;
            uint64_t i = lo + threadIdx.x;
            for (; i + (uint64_t)(kBatch - 1) * kSumThreads < hi; i += (uint64_t)kBatch * kSumThreads) {
                uint32_t seg[kBatch];
                uint2 ld[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k) seg[k] = steps[i + (uint64_t)k * kSumThreads] >> 1;
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    bad |= seg[k] >= n_segs;
                    ld[k] = tab[seg[k] < n_segs ? seg[k] : 0u];
                }
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    const bool ok = seg[k] < n_segs;
                    len += ok ? (unsigned long long)ld[k].x : 0ull;
                    wsum += ok ? (unsigned long long)ld[k].y * ld[k].x : 0ull;
                }
            }
            for (; i < hi; i += kSumThreads) {
                const uint32_t seg = steps[i] >> 1;
                const bool ok = seg < n_segs;
                const uint2 ld = tab[ok ? seg : 0u];
                bad |= !ok;
                len += ok ? (unsigned long long)ld.x : 0ull;
                wsum += ok ? (unsigned long long)ld.y * ld.x : 0ull;
            }
            if (bad) *status = 1u;
        } else if (threadIdx.x == 0 && part == 0) {
            *status = 1u;
        }
        // wave64 shuffle reduction, then one LDS hop across the block's 4 waves
        for (int off = 32; off > 0; off >>= 1) {
            len += __shfl_down(len, off, 64);
            wsum += __shfl_down(wsum, off, 64);
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) {
            red[0][wave] = len;
            red[1][wave] = wsum;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long a = 0, c = 0;
            for (int w = 0; w < kSumThreads / 64; ++w) {
                a += red[0][w];
                c += red[1][w];
            }
            const uint32_t at = by_path ? p : k;  // results in the order of the request, or indexed by path
            if (a) atomicAdd(&length_out[at], a);
            if (c) atomicAdd(&weighted_out[at], c);
        }
        __syncthreads();
    }
}

}  // namespace fgfa_dev

// ------------------------------------------------------------- the plan ---

using namespace fgfa_dev;

struct flatgfa_dev_plan {
    flatgfa_dev_graph_t g;
    int device = 0;
    int n_cus = 256;
    WorkItem *items = nullptr;  // flat-scan pieces
    uint32_t n_items = 0;
    uint32_t *status = nullptr;
    uint32_t n_windows = 1;
    FastPlan fast;  // the bucketed two-kernel path, used whenever the graph is eligible
    uint32_t *overlap_bits = nullptr;  // per-path coarse handle bitmaps (built on first overlap query)
    uint32_t *overlap_qbits = nullptr; // exact handle bitsets of the queries of the last overlap call (scratch)
    size_t overlap_qbytes = 0;
    bool overlap_qall = false;         // the exact bitsets are those of all paths, in path order (else: of the last call's queries)
    uint2 *len_depth = nullptr;        // (seg_len, depth) table of the last path_sums call (built on first use)
    // the outputs of the last node-depth call through the bucketed path: flatgfa_dev_status
    // completes that call if its records did not fit the sub-buckets
    uint32_t *last_depth = nullptr, *last_uniq = nullptr;
    uint64_t *last_len = nullptr, *last_weighted = nullptr;  // (a path_depth_all call)
    bool last_fast = false;
    uint32_t calls_since_status = 0;   // node-depth calls enqueued since the last flatgfa_dev_status: only the last can be completed there
    uint32_t *all_ids = nullptr;       // 0..n_paths-1 (path_depth_all without the bucketed path)
    int64_t cache_claim = 0;           // bytes of the device's Infinity Cache this plan's resident steps lay claim to (g_cache_claimed)
    bool cache_claim_shared = false;   // ... a claim another plan over the same step array made first (g_cache_shares): counted once
    std::vector<uint32_t> hb, he;      // the spans the plan was made with (flatgfa_dev_plan_steps_changed makes it again from them; the marks' job reads hb)
    uint32_t scan_workgroups = 0;      // (plan_create_impl's argument, likewise)
    hipStream_t side = nullptr;        // the stream the per-block no-claim marks are made on, off the way to the first answer
    std::vector<MarksJob> marks;       // one job per range / path group of `fast` that wants marks (fast itself, then more[0 ..]); empty once all are installed
    bool marks_to_start = false;       // ... whose jobs are enqueued by the first call (or status, or describe) behind the plan's creation: a caller that only wants the first answer never pays for them
    bool check_facts = false;          // FLATGFA_CHECK_NO_CLAIM=1: every call first looks again at what the plan took for granted about the step values
};

// The Infinity Cache (256 MiB on MI355X) is one per device: what the plans of a process keep resident in it is
// budgeted per device, first come first served, and given back when a plan is destroyed.
static std::mutex g_cache_mu;
static int64_t g_cache_claimed[64] = {};
// Plans over the SAME resident step array (two plans of one graph on two streams, so that one call's pass 2 runs
// beside the next call's pass 1; a plan per subset of paths) keep the same first megabytes of it resident: one
// claim, counted once, held until the last of them is destroyed.
struct CacheShare {
    int device;
    const uint32_t *steps;
    int64_t bytes;
    int users;
};
static std::vector<CacheShare> g_cache_shares;

extern "C" int flatgfa_dev_path_overlaps_impl(const flatgfa_dev_graph_t *g, int n_cus, uint32_t **coarse_cache,
                                              uint32_t **qbits_cache, size_t *qbits_bytes, bool *qbits_all, const uint32_t *query_ids, uint32_t n_q, uint8_t *touch_out,
                                              uint32_t *status, hipStream_t stream);

#define HIP_TRY(expr, fail_stmt)                                                            \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            fail_stmt;                                                                      \
        }                                                                                   \
    } while (0)

static int atomic_seg_depth(flatgfa_dev_plan_t *pl, uint32_t *depth_out, uint32_t *uniq_out, hipStream_t stream);

static flatgfa_dev_plan_t *plan_create_impl(const flatgfa_dev_graph_t *g, const uint32_t *hb, const uint32_t *he, uint32_t scan_workgroups,
                                            uint32_t *first_depth = nullptr, uint32_t *first_uniq = nullptr, int *first_status = nullptr);
extern "C" flatgfa_dev_plan_t *flatgfa_dev_plan_create(const flatgfa_dev_graph_t *g, const uint32_t *hb,
                                                        const uint32_t *he) {
    return plan_create_impl(g, hb, he, 0u);
}
extern "C" flatgfa_dev_plan_t *flatgfa_dev_plan_create_first(const flatgfa_dev_graph_t *g, const uint32_t *hb, const uint32_t *he,
                                                              uint32_t *depth_out, uint32_t *uniq_out, int *first_status) {
    if (g && g->n_segs && !depth_out) { set_error("dev_plan_create_first: NULL depth_out"); return nullptr; }
    return plan_create_impl(g, hb, he, 0u, depth_out, uniq_out, first_status);
}

// the properties of a device asked for once (hipGetDeviceProperties fills a kilobyte-sized struct through the driver: a tenth of a millisecond each time)
static int device_cus(int device) {
    static std::mutex mu;
    static int cus[64] = {};
    std::lock_guard<std::mutex> lk(mu);
    if (device >= 0 && device < 64 && cus[device]) return cus[device];
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    else (void)hipGetLastError();
    if (device >= 0 && device < 64) cus[device] = n;
    return n;
}
namespace fgfa_dev {
int device_cu_count(int device) { return device_cus(device); }
void plan_tick(const char *what) {
    static const char *const mode = getenv("FLATGFA_TIMING");  // (=host: the host's clock alone, the device not drained behind a stage)
    if (!mode) return;
    static const bool drain = std::string(mode) != "host";
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    if (what) {
        const auto before = std::chrono::steady_clock::now();
        if (drain) (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "plan: %-52s %8.3f ms (+ %.3f ms for the device to drain)\n", what, std::chrono::duration<double, std::milli>(before - last).count(),
                std::chrono::duration<double, std::milli>(now - before).count());
    }
    last = std::chrono::steady_clock::now();
}
}

namespace fgfa_dev {
hipError_t plan_memcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    constexpr size_t kChunk = (size_t)4 << 20;
    if (bytes < ((size_t)4 << 20) || (kind != hipMemcpyHostToDevice && kind != hipMemcpyDeviceToHost)) return hipMemcpy(dst, src, bytes, kind);
    // two pinned halves per process (portable: any device), two events per device; a plan's creation is not a place where threads
    // race for them, so one lock around the whole copy
    static std::mutex mu;
    static char *pinned = nullptr;
    static hipEvent_t ev[64][2] = {};
    std::lock_guard<std::mutex> lk(mu);
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if (device < 0 || device >= 64) return hipMemcpy(dst, src, bytes, kind);
    if (!pinned && hipHostMalloc((void **)&pinned, 2 * kChunk, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        pinned = nullptr;
        return hipMemcpy(dst, src, bytes, kind);
    }
    for (int k = 0; k < 2; ++k)
        if (!ev[device][k] && (e = hipEventCreateWithFlags(&ev[device][k], hipEventDisableTiming)) != hipSuccess) return e;
    const bool up = kind == hipMemcpyHostToDevice;
    // (a chunk travels for a tenth of a millisecond: polled, not slept on)
    const auto wait = [](hipEvent_t x) -> hipError_t {
        hipError_t q;
        while ((q = hipEventQuery(x)) == hipErrorNotReady) {}
        (void)hipGetLastError();  // ("not ready" must not be what a later check of the launches finds)
        return q;
    };
    size_t off = 0, prev_off = 0, prev_n = 0;
    int k = 0;
    bool busy[2] = {false, false};
    while (off < bytes || prev_n) {
        const size_t n = std::min(kChunk, bytes - off);
        char *half = pinned + (size_t)k * kChunk;
        if (n) {
            if (busy[k]) {  // (its last transfer must be over before the half is written again)
                if ((e = wait(ev[device][k])) != hipSuccess) return e;
                busy[k] = false;
            }
            if (up) {
                memcpy(half, (const char *)src + off, n);
                if ((e = hipMemcpyAsync((char *)dst + off, half, n, hipMemcpyHostToDevice, nullptr)) != hipSuccess) return e;
            } else {
                if ((e = hipMemcpyAsync(half, (const char *)src + off, n, hipMemcpyDeviceToHost, nullptr)) != hipSuccess) return e;
            }
            if ((e = hipEventRecord(ev[device][k], nullptr)) != hipSuccess) return e;
            busy[k] = true;
        }
        if (!up && prev_n) {  // the chunk before this one has arrived in the other half: hand it on while this one travels
            if ((e = wait(ev[device][k ^ 1])) != hipSuccess) return e;
            busy[k ^ 1] = false;
            memcpy((char *)dst + prev_off, pinned + (size_t)(k ^ 1) * kChunk, prev_n);
        }
        prev_off = off;
        prev_n = up ? 0 : n;
        off += n;
        k ^= 1;
    }
    for (int h = 0; h < 2; ++h)
        if (busy[h] && (e = wait(ev[device][h])) != hipSuccess) return e;
    return hipSuccess;
}
}  // namespace fgfa_dev

static void release_cache_claim(flatgfa_dev_plan_t *pl) {
    if (!pl->cache_claim) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    bool last = true;
    if (pl->cache_claim_shared) {
        for (size_t i = 0; i < g_cache_shares.size(); ++i) {
            CacheShare &c = g_cache_shares[i];
            if (c.device != pl->device || c.steps != pl->g.steps) continue;
            last = --c.users == 0;
            if (last) g_cache_shares.erase(g_cache_shares.begin() + (long)i);
            break;
        }
    }
    if (last && pl->device >= 0 && pl->device < 64) g_cache_claimed[pl->device] -= pl->cache_claim;
    pl->cache_claim = 0;
    pl->cache_claim_shared = false;
}

// The per-block no-claim marks are made on a stream of the plan's own, behind its creation:
// (wait) for the jobs that are there, and install what they found.  Called between two calls of the plan.
static void marks_start(flatgfa_dev_plan_t *pl);
static void marks_poll(flatgfa_dev_plan_t *pl, bool wait) {
    if (pl->marks_to_start) {
        pl->marks_to_start = false;
        marks_start(pl);
    }
    if (pl->marks.empty()) return;
    bool pending = false;
    for (size_t k = 0; k < pl->marks.size(); ++k) {
        MarksJob &job = pl->marks[k];
        if (!job.active) continue;
        if (!wait && !fast_marks_ready(job)) { pending = true; continue; }
        FastPlan *fp = k == 0 ? &pl->fast : (k - 1 < pl->fast.n_more ? &pl->fast.more[k - 1] : nullptr);
        if (fp) fast_marks_finish(fp, &job);
    }
    if (!pending) pl->marks.clear();
}
static void marks_start(flatgfa_dev_plan_t *pl) {
    if (!pl->fast.eligible) return;
    bool any = pl->fast.marks_wanted;
    for (uint32_t r = 0; r < pl->fast.n_more; ++r) any = any || pl->fast.more[r].marks_wanted;
    if (!any) return;
    // (the job's scratch and stream belong to the plan's device, whatever device the calling thread has current)
    int cur = pl->device;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
    struct Restore {
        int dev;
        bool on;
        ~Restore() { if (on) (void)hipSetDevice(dev); }
    } restore{cur, cur != pl->device};
    if (restore.on && hipSetDevice(pl->device) != hipSuccess) { (void)hipGetLastError(); restore.on = false; return; }
    if (!pl->side) {  // (one such stream per device for the whole process: creating a stream costs as much as the first answer)
        static std::mutex mu;
        static hipStream_t pool[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        hipStream_t &slot = pool[pl->device >= 0 && pl->device < 64 ? pl->device : 0];
        if (!slot) {
            // the lowest priority the device has: the marks' kernels read the steps three more times, and a query that runs beside them should not wait for that
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
            if (hipStreamCreateWithPriority(&slot, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); slot = nullptr; return; }
        }
        pl->side = slot;
    }
    pl->marks.assign(1 + pl->fast.n_more, MarksJob());
    for (size_t k = 0; k < pl->marks.size(); ++k) {
        const FastPlan &fp = k == 0 ? pl->fast : pl->fast.more[k - 1];
        if (fp.eligible && fp.marks_wanted) (void)fast_marks_start(fp, pl->g, pl->hb.data(), pl->side, &pl->marks[k]);  // (a job that cannot be had: no marks)
    }
}

// What a plan is beyond its handle: the bucketed path's plan (which kernel walks which path, the scratch), sized and
// timed on the graph.  Everything runs on the null stream.  With `first_depth` the query that sizes the record buckets
// leaves its result in the caller's buffers (and every timing run behind it writes the same counts there): the plan's
// creation IS the first query.  *first_st: the status bits that query raised (1 = an id out of range).
static bool plan_build_fast(flatgfa_dev_plan_t *pl, uint32_t *first_depth, uint32_t *first_uniq, uint32_t *first_st) {
    const flatgfa_dev_graph_t *g = &pl->g;
    const uint32_t *hb = pl->hb.data(), *he = pl->he.data();
    const uint32_t scan_workgroups = pl->scan_workgroups;
    const auto tick = [](const char *what) { plan_tick(what); };
    *first_st = 0;
    // FLATGFA_DEPTH_PATH=atomic forces the simple global-atomic kernels (used by the tests to
    // cover both device paths); anything else lets eligibility decide.
    const char *force = getenv("FLATGFA_DEPTH_PATH");
    if (!(force && std::string(force) == "atomic")) {
        if (!fast_plan_create(pl->g, hb, he, &pl->fast, scan_workgroups)) return false;
    }
    tick("fast_plan_create (lists, items, scratch)");
    // Steps kept in the Infinity Cache.  k_scan streams the steps past the caches (nt: whole lines read once),
    // which is right for what does not fit them -- but a resident graph is queried again and again, and the
    // first so-many megabytes of its steps, read WITHOUT the hint, are still in the 256 MiB cache when the next
    // call comes (the nt reads of the rest hit there but do not allocate, so they do not push it out): cfg-L's
    // k_scan 102 -> 88 us with 160 MB.  Only as much as the call's other traffic leaves room for -- its
    // records are written and read back through the same cache (8 bytes each), its results written (8
    // bytes per segment): a graph of 64 M segments has none to spare and would pay 12 % for the lines the
    // plain reads push out of the L2 -- and only what the device's other plans have not claimed.
    // FLATGFA_MALL_MB=n pins the amount (0: none).
    if (pl->fast.eligible && g->n_steps && pl->fast.n_items) {  // (k_scan's reads: the wave-per-path kernels and the partition read plainly anyway)
        uint64_t records = pl->fast.est_records;
        for (uint32_t r = 0; r < pl->fast.n_more; ++r) records += pl->fast.more[r].est_records;
        int64_t budget = (244ll << 20) - 8ll * (int64_t)g->n_segs - 8ll * (int64_t)records;
        budget = std::min<int64_t>(budget, 160ll << 20);
        if (const char *f = getenv("FLATGFA_MALL_MB")) budget = (int64_t)strtoull(f, nullptr, 10) << 20;
        budget = std::min<int64_t>(budget, (int64_t)g->n_steps * 4);
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            const bool pinned = getenv("FLATGFA_MALL_MB") != nullptr;  // (the thresholds below are the plan's own rule: a pinned amount is taken as given)
            CacheShare *share = nullptr;
            for (CacheShare &c : g_cache_shares)
                if (c.device == pl->device && c.steps == g->steps) share = &c;
            if (pl->device < 0 || pl->device >= 64) {
                budget = 0;
            } else if (share && !pinned) {  // another plan keeps this array's first bytes resident already: the same stretch, no second claim
                budget = std::min<int64_t>(share->bytes, (int64_t)g->n_steps * 4);
                share->users += 1;
                pl->cache_claim = share->bytes;
                pl->cache_claim_shared = true;
            } else {
                if (!pinned) {
                    budget = std::min<int64_t>(budget, (160ll << 20) - g_cache_claimed[pl->device]);
                    if (budget < (32ll << 20) || (int64_t)g->n_steps * 4 < (64ll << 20)) budget = 0;  // (not worth the L2 lines; graphs of a few million steps are launch-bound anyway)
                }
                budget = std::max<int64_t>(budget, 0);
                g_cache_claimed[pl->device] += budget;
                pl->cache_claim = budget;
                if (budget && !pinned) {
                    g_cache_shares.push_back(CacheShare{pl->device, g->steps, budget, 1});
                    pl->cache_claim_shared = true;
                }
            }
        }
        pl->fast.mall_steps = (uint64_t)budget / 4;
        for (uint32_t r = 0; r < pl->fast.n_more; ++r) pl->fast.more[r].mall_steps = pl->fast.mall_steps;
    }
    // A caller that wants its first answer from the atomic kernels' plan gets it here.
    const auto atomic_first = [&]() -> bool {
        if (!first_depth || !g->n_segs) return true;
        uint32_t st = 0;
        if (atomic_seg_depth(pl, first_depth, first_uniq, nullptr) != FLATGFA_OK || hipMemcpy(&st, pl->status, 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemset(pl->status, 0, 4) != hipSuccess)
            return false;
        *first_st |= st;
        return true;
    };
    // Size the sub-buckets for this graph now, with one query into scratch outputs (or the caller's: the first
    // answer), so that no later call runs out of room (the record counts per sub-bucket depend on the steps only):
    // a caller that consumes results on the stream -- an all-reduce right behind the kernels --
    // never sees an incomplete vector.  (FLATGFA_BUCKET_CAP keeps its forced capacity: the
    // tests want the overflow route.)
    if (pl->fast.eligible && !pl->fast.cap_forced && g->n_segs) {
        uint32_t *tmp = nullptr;
        const size_t need = (first_depth ? 0 : (size_t)g->n_segs) + (first_uniq ? 0 : (size_t)g->n_segs);
        if (need) HIP_TRY(hipMalloc(&tmp, need * 4), return false);
        uint32_t *const out_d = first_depth ? first_depth : tmp;
        uint32_t *const out_u = first_uniq ? first_uniq : (first_depth ? tmp : tmp + g->n_segs);
        bool complete = false;  // the outputs hold a whole query's counts
        for (int layout = 0; layout < 2; ++layout) {
        for (int attempt = 0; attempt < 10 && pl->fast.eligible; ++attempt) {
            uint32_t st = 0;
            if (fast_seg_depth(pl->fast, pl->g, out_d, out_u, pl->status, nullptr) != FLATGFA_OK ||
                hipMemcpy(&st, pl->status, 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemset(pl->status, 0, 4) != hipSuccess) {
                if (tmp) (void)hipFree(tmp);
                return false;
            }
            *first_st |= st & 1u;
            complete = !(st & (4u | 16u));
            if (!(st & 4u)) break;  // (an out-of-range id is reported by the query that meets it)
            (void)fast_plan_grow(&pl->fast);
        }
        tick("sizing query");
        {   // ... and with headroom: k_scan deals its items to the workgroups as they come, so another call may
            // fill a sub-bucket that was half full this time to the brim (see flatgfa_dev_status)
            uint32_t fullest = 0;
            if (hipMemcpy(&fullest, pl->status + 2, 4, hipMemcpyDeviceToHost) == hipSuccess && fullest && pl->fast.eligible) (void)fast_plan_grow(&pl->fast, true);
            (void)hipMemset(pl->status, 0, 12);
        }
        tick("headroom");
        // An even layout -- every sub-bucket as deep as the fullest -- that had to grow to gigabytes (paths that run along
        // the graph fill a few sub-buckets of a window and leave the others empty: 2000 contigs of 100 k steps on 4 M
        // segments, 0.8 GB of steps, 4.8 GB of buckets): the plan is made again with its buckets laid out to the count.
        if (layout == 0 && pl->fast.eligible && !getenv("FLATGFA_PACKED")) {
            uint64_t even_bytes = 0;
            const auto add = [&](const FastPlan &q) { if (!q.packed) even_bytes += ((uint64_t)q.n_win + 1) * q.n_slots * q.cap * 4; };
            add(pl->fast);
            for (uint32_t r = 0; r < pl->fast.n_more; ++r) add(pl->fast.more[r]);
            if (even_bytes <= (2ull << 30)) break;
            // (a plan that cannot have packed buckets -- paths that single waves walk, pass 1 by partition -- is not made again to find
            // that out: a million tiny paths paid 12 ms of their 25 for it)
            bool could = pl->fast.can_pack;
            for (uint32_t r = 0; r < pl->fast.n_more; ++r) could = could && pl->fast.more[r].can_pack;
            if (!could) break;
            FastPlan again;
            if (!fast_plan_create(pl->g, hb, he, &again, scan_workgroups, true)) { if (tmp) (void)hipFree(tmp); return false; }
            bool all_packed = again.eligible && again.packed;
            for (uint32_t r = 0; r < again.n_more; ++r) all_packed = all_packed && again.more[r].packed;
            if (!all_packed) {
                fast_plan_destroy(&again);
                break;
            }
            again.mall_steps = pl->fast.mall_steps;
            for (uint32_t r = 0; r < again.n_more; ++r) again.more[r].mall_steps = again.mall_steps;
            tick("the plan made again with packed buckets");
            fast_plan_destroy(&pl->fast);
            pl->fast = again;
            tick("the even plan's scratch freed");
        } else {
            break;
        }
        }
        // More than a record for two steps: pass 1 by partition (k_scan_dense) may beat pass 1 by runs.
        // The plan was sized for it (it makes the most records); now both are timed.
        {
            bool maybe = pl->fast.eligible && pl->fast.dense_maybe, any_maybe = maybe;  // all ranges / some range
            for (uint32_t r = 0; r < pl->fast.n_more; ++r) {
                maybe = maybe && pl->fast.more[r].dense_maybe;
                any_maybe = any_maybe || (pl->fast.eligible && pl->fast.more[r].dense_maybe);
            }
            if (maybe) {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                float best[2] = {1e30f, 1e30f};
                bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
                const auto set_dense = [&](bool on) {
                    pl->fast.dense = on;
                    for (uint32_t r = 0; r < pl->fast.n_more; ++r) pl->fast.more[r].dense = on;
                };
                for (int rep = 0; rep < 3 && ok; ++rep) {
                    for (int which = 0; which < 2 && ok; ++which) {
                        set_dense(which != 0);
                        ok = hipEventRecord(e0, nullptr) == hipSuccess;
                        const int rc = fast_seg_depth(pl->fast, pl->g, out_d, out_u, pl->status, nullptr);
                        float ms = 0;
                        ok = ok && rc == FLATGFA_OK && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                             hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
                        if (ok && rep) best[which] = std::min(best[which], ms);
                    }
                }
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                (void)hipMemset(pl->status, 0, 4);
                set_dense(ok && best[1] < best[0]);
                if (getenv("FLATGFA_TIMING")) fprintf(stderr, "plan: pass 1 by runs %.1f us, by partition %.1f us\n", best[0] * 1e3, best[1] * 1e3);
            } else if (any_maybe) {  // (ranges that disagree: all of them by runs, nothing left at an untimed default)
                pl->fast.dense = pl->fast.dense && !pl->fast.dense_maybe;
                for (uint32_t r = 0; r < pl->fast.n_more; ++r) pl->fast.more[r].dense = pl->fast.more[r].dense && !pl->fast.more[r].dense_maybe;
            }
        }
        // Pass 2 can look, before it maps a step's 64 records to their items, whether the step lies
        // inside the item of the step before: a win where a path has hundreds of records per window
        // (paths along the graph: -10 %; ids without runs: -35 %), a few instructions lost where it
        // has a dozen (+4 %).  Timed on this graph, both ways.
        if (pl->fast.eligible && !pl->fast.tagged && !test_hook("FLATGFA_BIG_GROUPS")) {  // (a tagged plan walks sub-buckets, not items)
            hipEvent_t e0 = nullptr, e1 = nullptr;
            float best[2] = {1e30f, 1e30f};
            bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
            for (int rep = 0; rep < 3 && ok; ++rep) {
                for (int which = 0; which < 2 && ok; ++which) {
                    pl->fast.big_groups = which != 0;
                    for (uint32_t r = 0; r < pl->fast.n_more; ++r) pl->fast.more[r].big_groups = which != 0;
                    ok = hipEventRecord(e0, nullptr) == hipSuccess;
                    const int rc = fast_seg_depth(pl->fast, pl->g, out_d, out_u, pl->status, nullptr);
                    float ms = 0;
                    ok = ok && rc == FLATGFA_OK && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                         hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
                    if (ok && rep) best[which] = std::min(best[which], ms);
                }
            }
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            (void)hipMemset(pl->status, 0, 4);
            const bool big = ok && best[1] * 1.02f < best[0];  // (it has to win by more than the noise of two runs)
            pl->fast.big_groups = big;
            for (uint32_t r = 0; r < pl->fast.n_more; ++r) pl->fast.more[r].big_groups = big;
            if (getenv("FLATGFA_TIMING")) fprintf(stderr, "plan: pass 2 item by item %.1f us, with the one-item shortcut %.1f us\n", best[0] * 1e3, best[1] * 1e3);
        } else if (const char *f = test_hook("FLATGFA_BIG_GROUPS")) {
            pl->fast.big_groups = strtol(f, nullptr, 10) != 0;
            for (uint32_t r = 0; r < pl->fast.n_more; ++r) pl->fast.more[r].big_groups = pl->fast.big_groups;
        }
        // Pass 2 of a tagged call can keep track of which tag owns each of a wave's bitsets (k_accum<..., OWN>): where a pass-1
        // workgroup takes hundreds of items of which a window sees a few -- tags twenty apart in a sub-bucket -- every step of
        // 64 records holds several, and the plain walk claims them one stretch after the other (160 000 contigs on 16 M
        // segments: pass 2 2.48 -> 1.62 ms); where the tags are dense it costs 4-9 %.  Timed on this graph, both ways, when
        // the workgroups take more items than a wave has bitsets.
        {
            const auto many_items = [](const FastPlan &q) {
                return q.tagged && !q.dense && (q.wb == 12 || q.wb == 13) && (uint64_t)q.n_items + q.max_back > 2ull * q.acc_slots * std::min<uint32_t>(std::max(q.n_items, 1u), q.n_slots);
            };
            bool any = pl->fast.eligible && many_items(pl->fast);
            for (uint32_t r = 0; r < pl->fast.n_more; ++r) any = any || (pl->fast.eligible && many_items(pl->fast.more[r]));
            const auto set_own = [&](bool on) {
                pl->fast.acc_own = on;
                for (uint32_t r = 0; r < pl->fast.n_more; ++r) pl->fast.more[r].acc_own = on;
            };
            if (any && !test_hook("FLATGFA_ACC_OWN")) {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                float best[2] = {1e30f, 1e30f};
                bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
                for (int rep = 0; rep < 3 && ok; ++rep) {
                    for (int which = 0; which < 2 && ok; ++which) {
                        set_own(which != 0);
                        ok = hipEventRecord(e0, nullptr) == hipSuccess;
                        const int rc = fast_seg_depth(pl->fast, pl->g, out_d, out_u, pl->status, nullptr);
                        float ms = 0;
                        ok = ok && rc == FLATGFA_OK && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                             hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
                        if (ok && rep) best[which] = std::min(best[which], ms);
                    }
                }
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                (void)hipMemset(pl->status, 0, 4);
                set_own(ok && best[1] * 1.02f < best[0]);  // (it has to win by more than the noise of two runs)
                if (getenv("FLATGFA_TIMING")) fprintf(stderr, "plan: pass 2 with bitsets by tag %.1f us, by owner %.1f us\n", best[0] * 1e3, best[1] * 1e3);
            }
        }
        // Small graphs are launch-bound: three kernels of the bucketed path against one of the
        // atomic path (10 k segments / 1 M steps: 76 us against 26).  Up to 8 M steps both are
        // timed here, on this graph, and the plan keeps the faster one.  FLATGFA_DEPTH_PATH=bucketed
        // (or any of the knobs that shape the bucketed path) skips the comparison.
        bool shaped = force != nullptr;
        for (const char *k : {"FLATGFA_PIECE_STEPS", "FLATGFA_SHORT_MAX", "FLATGFA_SHORT_ANY", "FLATGFA_ACC_PARTS", "FLATGFA_RANGE_SEGS",
#ifdef FGFA_MEASURE
                              "FLATGFA_DEBUG_SKIP",
#endif
                              "FLATGFA_WB", "FLATGFA_DENSE", "FLATGFA_TAGGED", "FLATGFA_PATH_GROUPS"})
            shaped = shaped || test_hook(k) != nullptr;
        if (pl->fast.eligible && !shaped && g->n_steps <= (8u << 20)) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            float best[2] = {1e30f, 1e30f};
            bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
            for (int which = 0; which < 2 && ok; ++which) {
                for (int rep = 0; rep < 4 && ok; ++rep) {
                    ok = hipEventRecord(e0, nullptr) == hipSuccess;
                    const int rc = which ? atomic_seg_depth(pl, out_d, out_u, nullptr)
                                         : fast_seg_depth(pl->fast, pl->g, out_d, out_u, pl->status, nullptr);
                    float ms = 0;
                    ok = ok && rc == FLATGFA_OK && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                         hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
                    if (ok && rep) best[which] = std::min(best[which], ms);
                }
            }
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            (void)hipMemset(pl->status, 0, 4);  // (an out-of-range id shows up again in the caller's own first query)
            if (ok && best[1] < best[0]) fast_plan_destroy(&pl->fast);
            if (getenv("FLATGFA_TIMING")) fprintf(stderr, "plan: bucketed %.1f us, atomic %.1f us\n", best[0] * 1e3, best[1] * 1e3);
        }
        tick("timed choices");
        if (tmp) (void)hipFree(tmp);
        // (a plan that could not be given room -- it then runs the atomic kernels -- left an incomplete vector behind)
        if (!complete && !atomic_first()) return false;
    } else if (!atomic_first()) {
        return false;
    }
    // Everything above went through the null stream, which a caller's non-blocking stream does not
    // wait for -- and a memset of device memory need not have happened when hipMemset returns: a
    // plan without a trial call (no paths, say) could have its status words read, on the caller's
    // stream, before they were cleared (seen once: a freed block's contents taken for status bits).
    (void)hipStreamSynchronize(nullptr);
    tick("null stream drained");
    // ... and what can wait until the first answer is out -- the per-block no-claim marks -- is enqueued, on a stream of the
    // plan's own, by whatever the caller does with the plan next (marks_poll)
    pl->marks_to_start = true;
    return true;
}

// (scan_workgroups: pass 1's persistent workgroups, 0 = one per CU; a pipeline's lanes take fewer)
static flatgfa_dev_plan_t *plan_create_impl(const flatgfa_dev_graph_t *g, const uint32_t *hb, const uint32_t *he, uint32_t scan_workgroups,
                                            uint32_t *first_depth, uint32_t *first_uniq, int *first_status) {
    if (first_status) *first_status = FLATGFA_OK;
    if (!g) { set_error("plan_create: NULL graph"); return nullptr; }
    fgfa_dev::TempScope temporaries;  // (the large vectors of a plan's creation come from blocks the thread keeps: temp_arena.hpp)
    fgfa_dev::Vec<uint32_t> cb, ce;
    if (g->n_paths && (!hb || !he)) {
        cb.resize(g->n_paths);
        ce.resize(g->n_paths);
        HIP_TRY(fgfa_dev::plan_memcpy(cb.data(), g->path_begin, (size_t)g->n_paths * 4, hipMemcpyDeviceToHost), return nullptr);
        HIP_TRY(fgfa_dev::plan_memcpy(ce.data(), g->path_end, (size_t)g->n_paths * 4, hipMemcpyDeviceToHost), return nullptr);
        hb = cb.data();
        he = ce.data();
    }
    fgfa_dev::Vec<WorkItem> items;
    for (uint32_t p = 0; p < g->n_paths; ++p) {
        if (hb[p] > he[p] || (uint64_t)he[p] > g->n_steps) {
            set_error("plan_create: path " + std::to_string(p) + " has a step span outside the steps pool");
            return nullptr;
        }
        for (uint64_t b = hb[p]; b < he[p]; b += kScanPiece)
            items.push_back(WorkItem{(uint32_t)b, (uint32_t)std::min<uint64_t>(b + kScanPiece, he[p]), p, 0u});
    }
    plan_tick(nullptr);
    auto *pl = new flatgfa_dev_plan();
    pl->g = *g;
    pl->hb.assign(hb, hb + g->n_paths);
    pl->he.assign(he, he + g->n_paths);
    pl->scan_workgroups = scan_workgroups;
    if (const char *c = getenv("FLATGFA_CHECK_NO_CLAIM")) pl->check_facts = c[0] != '0' && c[0] != 0;
    HIP_TRY(hipGetDevice(&pl->device), { delete pl; return nullptr; });
    pl->n_cus = device_cus(pl->device);
    pl->n_items = (uint32_t)items.size();
    pl->n_windows = g->n_segs ? (uint32_t)(((uint64_t)g->n_segs + kWinBits - 1) / kWinBits) : 1;
    HIP_TRY(hipMalloc(&pl->status, 256), { delete pl; return nullptr; });
    HIP_TRY(hipMemset(pl->status, 0, 256), { flatgfa_dev_plan_destroy(pl); return nullptr; });
    if (!items.empty()) {
        HIP_TRY(hipMalloc(&pl->items, items.size() * sizeof(WorkItem)), { flatgfa_dev_plan_destroy(pl); return nullptr; });
        HIP_TRY(fgfa_dev::plan_memcpy(pl->items, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice),
                { flatgfa_dev_plan_destroy(pl); return nullptr; });
    }
    {   // (per device, like the other kernels' attributes)
        static OncePerDevice once;
        if (!once([] { return hipFuncSetAttribute((const void *)k_depth_uniq_path, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kWinWords * 4)) == hipSuccess; })) {
            set_error("hipFuncSetAttribute(k_depth_uniq_path): dynamic shared memory");
            flatgfa_dev_plan_destroy(pl);
            return nullptr;
        }
    }
    plan_tick("handle: status words, atomic path's items, attributes");
    uint32_t first_st = 0;
    if (!plan_build_fast(pl, first_depth, first_uniq, &first_st)) { flatgfa_dev_plan_destroy(pl); return nullptr; }
    if (first_status && (first_st & 1u)) {
        set_error("a step refers to a segment id (or a query to a path id) that is out of range");
        *first_status = FLATGFA_ERR_BOUNDS;
    }
    return pl;
}

// What plan_build_fast made, given back: the marks' jobs (waited for, nothing installed), the claim on the Infinity Cache, the plan.
static void plan_release_fast(flatgfa_dev_plan_t *pl) {
    pl->marks_to_start = false;
    if (pl->side) (void)hipStreamSynchronize(pl->side);
    for (MarksJob &job : pl->marks) {
        FastPlan nothing;  // (finish releases the job's scratch; what it would install goes with this)
        nothing.cflags = reinterpret_cast<uint32_t *>(1);  // (non-null: nothing is installed)
        fast_marks_finish(&nothing, &job);
    }
    pl->marks.clear();
    release_cache_claim(pl);
    fast_plan_destroy(&pl->fast);
}

extern "C" int flatgfa_dev_plan_steps_changed(flatgfa_dev_plan_t *pl, void *stream_) {
    if (!pl) { set_error("dev_plan_steps_changed: NULL plan"); return FLATGFA_ERR_ARG; }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_), return FLATGFA_ERR_HIP);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev), return FLATGFA_ERR_HIP);
    if (dev != pl->device) HIP_TRY(hipSetDevice(pl->device), return FLATGFA_ERR_HIP);
    plan_release_fast(pl);
    // everything else that was derived from the step values: the overlap query's bitmaps, the (seg_len, depth) table
    for (void **p : {(void **)&pl->overlap_bits, (void **)&pl->overlap_qbits}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    pl->overlap_qbytes = 0;
    pl->overlap_qall = false;
    HIP_TRY(hipMemset(pl->status, 0, 256), return FLATGFA_ERR_HIP);
    pl->calls_since_status = 0;
    pl->last_fast = false;
    pl->last_depth = pl->last_uniq = nullptr;
    pl->last_len = pl->last_weighted = nullptr;
    uint32_t st = 0;
    const bool ok = plan_build_fast(pl, nullptr, nullptr, &st);
    if (dev != pl->device) (void)hipSetDevice(dev);
    return ok ? FLATGFA_OK : FLATGFA_ERR_HIP;
}

extern "C" void flatgfa_dev_release_scratch(void) { fgfa_dev::fast_release_scratch(); }

extern "C" void flatgfa_dev_plan_destroy(flatgfa_dev_plan_t *pl) {
    if (!pl) return;
    plan_release_fast(pl);  // (pl->side is the device's, not the plan's: it stays)
    if (pl->overlap_bits) (void)hipFree(pl->overlap_bits);
    if (pl->overlap_qbits) (void)hipFree(pl->overlap_qbits);
    if (pl->len_depth) (void)hipFree(pl->len_depth);
    if (pl->all_ids) (void)hipFree(pl->all_ids);
    if (pl->items) (void)hipFree(pl->items);
    if (pl->status) (void)hipFree(pl->status);
    delete pl;
}

// The simple global-atomic kernels: the general fallback, and the correctness anchor of the tests.
static int atomic_seg_depth(flatgfa_dev_plan_t *pl, uint32_t *depth_out, uint32_t *uniq_out, hipStream_t stream) {
    const flatgfa_dev_graph_t &g = pl->g;
    {
        ProfScope ps("k_zero_outputs", stream);
        const uint32_t zgrid = std::max<uint32_t>(1u, std::min<uint32_t>((g.n_segs + 1023u) / 1024u, (uint32_t)pl->n_cus * 8u));
        hipLaunchKernelGGL(k_zero_outputs, dim3(zgrid), dim3(256), 0, stream, depth_out, uniq_out, g.n_segs);
    }
    if (g.n_paths == 0 || pl->n_items == 0) return FLATGFA_OK;
    if (!uniq_out) {
        ProfScope ps("k_depth_scan", stream);
        uint32_t grid = std::min<uint32_t>(pl->n_items, (uint32_t)pl->n_cus * 8u);
        hipLaunchKernelGGL(k_depth_scan, dim3(grid), dim3(kScanThreads), 0, stream, g.steps, pl->items, pl->n_items,
                           g.n_segs, depth_out, pl->status);
    } else {
        ProfScope ps("k_depth_uniq_path", stream);
        uint64_t jobs = (uint64_t)g.n_paths * pl->n_windows;
        uint32_t grid = (uint32_t)std::min<uint64_t>(jobs, (uint64_t)pl->n_cus * 64u);
        hipLaunchKernelGGL(k_depth_uniq_path, dim3(grid), dim3(kUniqThreads), kWinWords * 4, stream, g.steps,
                           g.path_begin, g.path_end, g.n_paths, g.n_segs, pl->n_windows, depth_out, uniq_out,
                           pl->status);
    }
    HIP_TRY(hipGetLastError(), return FLATGFA_ERR_HIP);
    return FLATGFA_OK;
}

extern "C" int flatgfa_dev_seg_depth(flatgfa_dev_plan_t *pl, uint32_t *depth_out, uint32_t *uniq_out, void *stream_) {
    if (!pl || (!depth_out && pl->g.n_segs)) { set_error("dev_seg_depth: NULL argument"); return FLATGFA_ERR_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    const flatgfa_dev_graph_t &g = pl->g;
    if (g.n_segs == 0) return FLATGFA_OK;
    marks_poll(pl, false);  // (the per-block no-claim marks, once they are there: installed between two calls)
    pl->last_fast = pl->fast.eligible;
    pl->last_depth = depth_out;
    pl->last_uniq = uniq_out;
    pl->last_len = pl->last_weighted = nullptr;
    pl->calls_since_status += 1;
    if (pl->fast.eligible) {
        if (pl->check_facts) {
            const int rc = fast_check_plan_facts(pl->fast, g, pl->hb.data(), pl->he.data(), pl->status, stream);
            if (rc) return rc;
        }
        return fast_seg_depth(pl->fast, g, depth_out, uniq_out, pl->status, stream);
    }
    return atomic_seg_depth(pl, depth_out, uniq_out, stream);
}

// the sums of `n_ids` paths (device array) through k_path_sums, results indexed by path when by_path
static int path_sums_launch(flatgfa_dev_plan_t *pl, const uint32_t *path_ids, uint32_t n_ids, const uint32_t *depth,
                            uint64_t *length_out, uint64_t *weighted_out, uint32_t by_path, hipStream_t stream) {
    const flatgfa_dev_graph_t &g = pl->g;
    uint32_t split = std::max<uint32_t>(1u, std::min<uint32_t>(64u, (uint32_t)(pl->n_cus * 16) / n_ids));
    uint64_t jobs = (uint64_t)n_ids * split;
    uint32_t grid = (uint32_t)std::min<uint64_t>(jobs, (uint64_t)pl->n_cus * 32u);
    if (!pl->len_depth) HIP_TRY(hipMalloc(&pl->len_depth, (size_t)std::max<uint32_t>(g.n_segs, 1u) * sizeof(uint2)), return FLATGFA_ERR_HIP);
    {
        ProfScope ps("k_pack_len_depth", stream);
        const uint32_t pgrid = std::max<uint32_t>(1u, std::min<uint32_t>((g.n_segs + 255) / 256, (uint32_t)pl->n_cus * 8u));
        hipLaunchKernelGGL(k_pack_len_depth, dim3(pgrid), dim3(256), 0, stream, g.seg_len, depth, g.n_segs, pl->len_depth);
    }
    {
        ProfScope ps("k_path_sums", stream);
        hipLaunchKernelGGL(k_path_sums, dim3(grid), dim3(kSumThreads), 0, stream, g.steps, g.path_begin, g.path_end,
                           g.n_paths, g.n_segs, pl->len_depth, path_ids, n_ids, split, by_path,
                           (unsigned long long *)length_out, (unsigned long long *)weighted_out, pl->status);
    }
    HIP_TRY(hipGetLastError(), return FLATGFA_ERR_HIP);
    return FLATGFA_OK;
}

extern "C" int flatgfa_dev_path_sums(flatgfa_dev_plan_t *pl, const uint32_t *path_ids, uint32_t n_ids,
                                     const uint32_t *depth, uint64_t *length_out, uint64_t *weighted_out,
                                     void *stream_) {
    if (!pl || (n_ids && (!path_ids || !depth || !length_out || !weighted_out))) {
        set_error("dev_path_sums: NULL argument");
        return FLATGFA_ERR_ARG;
    }
    if (n_ids == 0) return FLATGFA_OK;
    const flatgfa_dev_graph_t &g = pl->g;
    if (!g.seg_len && g.n_segs) { set_error("dev_path_sums: graph image has no seg_len array"); return FLATGFA_ERR_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    {
        ProfScope ps("memset_path_sums", stream);
        HIP_TRY(hipMemsetAsync(length_out, 0, (size_t)n_ids * 8, stream), return FLATGFA_ERR_HIP);
        HIP_TRY(hipMemsetAsync(weighted_out, 0, (size_t)n_ids * 8, stream), return FLATGFA_ERR_HIP);
    }
    return path_sums_launch(pl, path_ids, n_ids, depth, length_out, weighted_out, 0u, stream);
}

static int path_depth_all_enqueue(flatgfa_dev_plan_t *pl, uint32_t *depth_out, uint64_t *length_out, uint64_t *weighted_out,
                                  hipStream_t stream, bool use_fast) {
    const flatgfa_dev_graph_t &g = pl->g;
    use_fast = use_fast && pl->fast.eligible;
    if (use_fast && fast_plan_want_path_sums(&pl->fast)) {
        // pass 2 adds up the paths k_scan walked; the wave-per-path kernels' paths take the gather kernel
        const PathSums ps{length_out, weighted_out, true};
        int rc = fast_seg_depth(pl->fast, g, depth_out, nullptr, pl->status, stream, &ps);
        if (rc) return rc;
        if (pl->fast.n_other) rc = path_sums_launch(pl, pl->fast.other_ids, pl->fast.n_other, depth_out, length_out, weighted_out, 1u, stream);
        return rc;
    }
    {
        ProfScope ps("memset_path_sums", stream);
        HIP_TRY(hipMemsetAsync(length_out, 0, (size_t)g.n_paths * 8, stream), return FLATGFA_ERR_HIP);
        HIP_TRY(hipMemsetAsync(weighted_out, 0, (size_t)g.n_paths * 8, stream), return FLATGFA_ERR_HIP);
    }
    int rc = use_fast ? fast_seg_depth(pl->fast, g, depth_out, nullptr, pl->status, stream)
                      : atomic_seg_depth(pl, depth_out, nullptr, stream);
    if (rc) return rc;
    if (!pl->all_ids) {
        std::vector<uint32_t> ids(g.n_paths);
        for (uint32_t i = 0; i < g.n_paths; ++i) ids[i] = i;
        HIP_TRY(hipMalloc(&pl->all_ids, (size_t)g.n_paths * 4), return FLATGFA_ERR_HIP);
        HIP_TRY(fgfa_dev::plan_memcpy(pl->all_ids, ids.data(), (size_t)g.n_paths * 4, hipMemcpyHostToDevice), return FLATGFA_ERR_HIP);
    }
    return path_sums_launch(pl, pl->all_ids, g.n_paths, depth_out, length_out, weighted_out, 1u, stream);
}

extern "C" int flatgfa_dev_path_depth_all(flatgfa_dev_plan_t *pl, uint32_t *depth_out, uint64_t *length_out,
                                          uint64_t *weighted_out, void *stream_) {
    if (!pl || (pl->g.n_segs && !depth_out) || (pl->g.n_paths && (!length_out || !weighted_out))) {
        set_error("dev_path_depth_all: NULL argument");
        return FLATGFA_ERR_ARG;
    }
    const flatgfa_dev_graph_t &g = pl->g;
    if (g.n_paths == 0) return g.n_segs ? flatgfa_dev_seg_depth(pl, depth_out, nullptr, stream_) : FLATGFA_OK;
    if (!g.seg_len && g.n_segs) { set_error("dev_path_depth_all: graph image has no seg_len array"); return FLATGFA_ERR_ARG; }
    if (g.n_segs == 0) {
        HIP_TRY(hipMemsetAsync(length_out, 0, (size_t)g.n_paths * 8, (hipStream_t)stream_), return FLATGFA_ERR_HIP);
        HIP_TRY(hipMemsetAsync(weighted_out, 0, (size_t)g.n_paths * 8, (hipStream_t)stream_), return FLATGFA_ERR_HIP);
        return FLATGFA_OK;
    }
    marks_poll(pl, false);
    pl->last_fast = pl->fast.eligible;
    pl->last_depth = depth_out;
    pl->last_uniq = nullptr;
    pl->last_len = length_out;
    pl->last_weighted = weighted_out;
    pl->calls_since_status += 1;
    if (pl->check_facts && pl->fast.eligible) {
        const int rc = fast_check_plan_facts(pl->fast, g, pl->hb.data(), pl->he.data(), pl->status, (hipStream_t)stream_);
        if (rc) return rc;
    }
    return path_depth_all_enqueue(pl, depth_out, length_out, weighted_out, (hipStream_t)stream_, true);
}

extern "C" int flatgfa_dev_path_overlaps(flatgfa_dev_plan_t *pl, const uint32_t *query_ids, uint32_t n_q,
                                         uint8_t *touch_out, void *stream_) {
    if (!pl || (n_q && (!query_ids || !touch_out))) { set_error("dev_path_overlaps: NULL argument"); return FLATGFA_ERR_ARG; }
    return flatgfa_dev_path_overlaps_impl(&pl->g, pl->n_cus, &pl->overlap_bits, &pl->overlap_qbits, &pl->overlap_qbytes, &pl->overlap_qall,
                                          query_ids, n_q, touch_out, pl->status,
                                          (hipStream_t)stream_);
}

// Status word bits set by the kernels: 1 = an id was out of range, 2 = a diagnostic mode's
// sentinel, 4 = a record did not fit its sub-bucket (depth_fast.hip).
extern "C" int flatgfa_dev_status(flatgfa_dev_plan_t *pl, void *stream_) {
    if (!pl) return FLATGFA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    for (int attempt = 0;; ++attempt) {
        uint32_t st3[3] = {0, 0, 0};  // the flags; -; the fullest sub-bucket beyond half of the capacity
        HIP_TRY(hipMemcpyAsync(st3, pl->status, 12, hipMemcpyDeviceToHost, stream), return FLATGFA_ERR_HIP);
        HIP_TRY(hipStreamSynchronize(stream), return FLATGFA_ERR_HIP);
        const uint32_t st = st3[0];
        const uint32_t n_calls = pl->calls_since_status;
        if (attempt == 0) {
            pl->calls_since_status = 0;
            marks_poll(pl, false);  // (no call is in flight: if the marks' job is done, what it found is installed -- it is never waited for here)
        }
        if (st3[2]) {
            // A call filled a sub-bucket more than half: k_scan deals its items to the workgroups as they
            // come, so a later call may fill it differently -- on a graph whose paths run along it,
            // twice as much when two of a workgroup's items meet in a window.  Make room now (no call
            // is in flight), not when a call enqueued among others has already run out.
            HIP_TRY(hipMemsetAsync(pl->status + 2, 0, 4, stream), return FLATGFA_ERR_HIP);
            if (!(st & (4u | 16u)) && pl->fast.eligible) (void)fast_plan_grow(&pl->fast, true);  // (a plan at its limit stays as it is)
        }
        if (!st) return FLATGFA_OK;
        HIP_TRY(hipMemsetAsync(pl->status, 0, 4, stream), return FLATGFA_ERR_HIP);
        if (st & 1u) {
            set_error("a step refers to a segment id (or a query to a path id) that is out of range");
            return FLATGFA_ERR_BOUNDS;
        }
        if (st & 32u) {  // (kStStale: FLATGFA_CHECK_NO_CLAIM=1)
            set_error("the step values changed behind the plan (a path it found strictly monotone no longer is, a no-claim mark or a reversed copy no longer holds): "
                      "call flatgfa_dev_plan_steps_changed, or make the plan again");
            return FLATGFA_ERR_STALE_PLAN;
        }
        if (st & 8u) {  // (depth_fast.hip: kStInternal)
            char what[512] = "";
            (void)flatgfa_dev_plan_describe(pl, what, (int)sizeof what);
            set_error("node depth: internal error (the records of pass 1 were not in the order pass 2 relies on); status words " + std::to_string(st3[0]) + " " +
                      std::to_string(st3[1]) + " " + std::to_string(st3[2]) + ", " + std::to_string(n_calls) + " call(s) since the last status; plan: " + what);
            return FLATGFA_ERR_HIP;
        }
        if (!(st & (4u | 16u))) return FLATGFA_OK;
        if (attempt == 0 && n_calls > 1) {
            // Several calls were enqueued since the last status and (at least) one of them ran out of
            // scratch room: which one is not recorded, and only the last one's outputs are known here.
            set_error("node depth: a call ran out of scratch room (the steps changed behind the plan?) and " + std::to_string(n_calls) +
                      " calls were enqueued since the last flatgfa_dev_status: call it after every call to have such a call completed");
            return FLATGFA_ERR_HIP;
        }
        // The last node-depth call ran out of sub-bucket room, so its outputs are incomplete: run
        // it again with four times the room, or -- when the plan cannot grow -- through the
        // atomic kernels.  The scratch cleans itself in pass 2, whatever was dropped.
        if (!pl->last_fast || !pl->last_depth || attempt > 8) {
            set_error("node depth: scratch overflow could not be resolved");
            return FLATGFA_ERR_HIP;
        }
        int rc;
        const bool grown = !(st & 16u) && fast_plan_grow(&pl->fast);  // (a full hand-back list is not a matter of bucket room)
        if (st & 16u) pl->fast.eligible = false;  // (later calls take the atomic kernels right away)
        if (!grown) pl->last_fast = false;
        if (pl->last_len) rc = path_depth_all_enqueue(pl, pl->last_depth, pl->last_len, pl->last_weighted, stream, grown);
        else if (grown) rc = fast_seg_depth(pl->fast, pl->g, pl->last_depth, pl->last_uniq, pl->status, stream);
        else rc = atomic_seg_depth(pl, pl->last_depth, pl->last_uniq, stream);
        if (rc) return rc;
    }
}

// Which kernels a plan's calls run: the choices made (some of them by timing) when it was created.
extern "C" int flatgfa_dev_plan_describe(flatgfa_dev_plan_t *pl, char *out, int cap) {
    if (!pl || !out || cap <= 0) return 0;
    marks_poll(pl, true);  // (what the description says about no-claim marks is what later calls run with)
    const FastPlan &f = pl->fast;
    std::string s;
    if (!f.eligible) {
        s = "path=atomic (k_depth_scan / k_depth_uniq_path)";
    } else {
        const bool alone = (f.n_short || f.n_medium || f.n_tiny) && f.n_items == 0 && f.exact_short;  // (k_scan's launch is left out)
        s = "path=bucketed pass1=" + std::string(alone ? "" : f.dense ? "k_scan_dense+" : "k_scan+") + (f.n_short ? "k_scan_short+" : "") + (f.n_medium ? "k_scan_medium+" : "") + (f.n_tiny ? "k_scan_tiny+" : "");
        s.pop_back();
        s += std::string("") +
            " pass2=" + (f.tagged ? (f.n_shared ? "tagged(shared bitsets)" : f.acc_pair ? "tagged(two workgroups per window)" : "tagged") : (f.big_groups ? "directory(one-item shortcut)" : "directory")) +
            " windows=" + std::to_string(f.n_win) + "x" + std::to_string(1u << f.wb) + " ranges=" + std::to_string((f.n_more + 1) / f.n_groups) +
            (f.n_groups > 1 ? " path_groups=" + std::to_string(f.n_groups) : std::string()) +
            (f.narrow_emit && f.tagged && !f.packed && !f.dense && !f.cflags && !f.n_more ? " emit=two_chunks" : "") +
            " scan_workgroups=" + std::to_string(f.n_slots) + " workgroups_per_window=" + std::to_string(f.acc_parts) + " items=" + std::to_string(f.n_items) + (f.acc_own ? " bitset_owners=tracked" : "") + " no_claim_items=" + std::to_string(f.n_noclaim) + (f.n_flag_chunks ? " no_claim_chunks=" + std::to_string(f.n_flag_chunks) : std::string()) + " no_claim_paths=" + std::to_string(f.short_mono_n + f.medium_mono_n + f.tiny_mono_n) + " split_paths=" + std::to_string(f.n_shared) +
            " short_paths=" + std::to_string(f.n_short) + " medium_paths=" + std::to_string(f.n_medium) + " tiny_paths=" + std::to_string(f.n_tiny) +
            " steps=" + std::to_string(f.class_steps[0]) + "/" + std::to_string(f.class_steps[1]) + "/" + std::to_string(f.class_steps[2]) + "/" + std::to_string(f.class_steps[3]) +  // (by k_scan / short / medium / tiny)
            " bucket_cap=" + std::to_string(f.cap);
        // the record buckets of all ranges and path groups: how they are laid out, what they take
        const auto records_of = [](const FastPlan &q) { return q.packed ? q.bucket_records : ((uint64_t)q.n_win + 1) * q.n_slots * q.cap; };
        uint64_t recs = records_of(f);
        bool packed = f.packed;
        for (uint32_t r = 0; r < f.n_more; ++r) {
            recs += records_of(f.more[r]);
            packed = packed || f.more[r].packed;
        }
        s += std::string(" buckets=") + (packed ? "packed" : "even") + " scratch_mb=" + std::to_string((recs * 4 + (1u << 20) - 1) >> 20) +
             " cache_resident_mb=" + std::to_string((f.mall_steps * 4) >> 20);
    }
    const int n = (int)std::min<size_t>(s.size(), (size_t)cap - 1);
    memcpy(out, s.data(), (size_t)n);
    out[n] = 0;
    return n;
}

extern "C" void flatgfa_dev_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

extern "C" int flatgfa_dev_profile_read(const char **names, float *ms, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    std::vector<std::pair<int, hipEvent_t>> done;
    for (auto &r : g_prof) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        if (n < cap) {
            names[n] = r.name;
            ms[n] = t;
            ++n;
        }
        done.push_back({r.device, r.a});
        done.push_back({r.device, r.b});
    }
    g_prof.clear();
    for (const auto &de : done) {  // (back to their device's pool: events are reused, not created per kernel)
        if ((size_t)de.first >= g_event_pool.size()) g_event_pool.resize((size_t)de.first + 1);
        std::vector<hipEvent_t> &pool = g_event_pool[(size_t)de.first];
        if (pool.size() < 256) pool.push_back(de.second);
        else (void)hipEventDestroy(de.second);
    }
    return n;
}

// What the event pair around a launch measures when the kernel does nothing: a launch of k_scan's
// shape (one workgroup of 1024 threads per CU, `lds_bytes` of dynamic LDS), bracketed like every
// profiled kernel.  bench.py reports it beside the kernel times (rocprofv3's dispatch durations do
// not contain it).
__global__ __launch_bounds__(1024) void k_nothing(uint32_t *sink) {
    extern __shared__ uint32_t lds_nothing[];
    if (sink && threadIdx.x == 4096) sink[0] = lds_nothing[0];
}

namespace fgfa_dev {
void warm_launch(void *stream) { hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t *)nullptr); }
}  // namespace fgfa_dev

extern "C" float flatgfa_dev_profile_overhead_ms(int n_workgroups, int lds_bytes, int reps, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_workgroups <= 0 || reps <= 0 || lds_bytes < 0 || lds_bytes > 160 * 1024) return -1.f;
    if (hipFuncSetAttribute((const void *)k_nothing, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1.f;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreateWithFlags(&a, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableSystemFence) != hipSuccess) return -1.f;  // (as the profiling events)
    std::vector<float> ts;
    for (int r = 0; r < reps + 2; ++r) {
        (void)hipEventRecord(a, stream);
        hipLaunchKernelGGL(k_nothing, dim3(n_workgroups), dim3(1024), (size_t)lds_bytes, stream, (uint32_t *)nullptr);
        (void)hipEventRecord(b, stream);
        float t = 0.f;
        if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&t, a, b) != hipSuccess) { ts.clear(); break; }
        if (r >= 2) ts.push_back(t);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (ts.empty()) return -1.f;
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

// ---- calls in flight: K plans of one resident graph on K internal streams ----
struct flatgfa_dev_pipeline {
    int device = 0;
    std::vector<flatgfa_dev_plan_t *> plans;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> joined;   // one per lane: recorded when a caller's stream joins (not per call: an event behind every call costs tens of microseconds)
    hipEvent_t after = nullptr;       // recorded on a caller's stream whose work a call has to wait for
    uint64_t n_calls = 0;
};

extern "C" void flatgfa_dev_pipeline_destroy(flatgfa_dev_pipeline_t *p) {
    if (!p) return;
    for (size_t k = 0; k < p->streams.size(); ++k)
        if (p->streams[k]) (void)hipStreamSynchronize(p->streams[k]);
    for (flatgfa_dev_plan_t *pl : p->plans) flatgfa_dev_plan_destroy(pl);
    for (hipEvent_t e : p->joined)
        if (e) (void)hipEventDestroy(e);
    if (p->after) (void)hipEventDestroy(p->after);
    for (hipStream_t st : p->streams)
        if (st) (void)hipStreamDestroy(st);
    delete p;
}

extern "C" flatgfa_dev_pipeline_t *flatgfa_dev_pipeline_create(const flatgfa_dev_graph_t *g, const uint32_t *hb, const uint32_t *he, int calls_in_flight) {
    if (!g || calls_in_flight < 1 || calls_in_flight > 8) { set_error("dev_pipeline_create: bad argument (1 .. 8 calls in flight)"); return nullptr; }
    auto *p = new flatgfa_dev_pipeline();
    HIP_TRY(hipGetDevice(&p->device), { delete p; return nullptr; });
    for (int k = 0; k < calls_in_flight; ++k) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            set_error("dev_pipeline_create: cannot create a stream");
            if (st) (void)hipStreamDestroy(st);
            flatgfa_dev_pipeline_destroy(p);
            return nullptr;
        }
        p->streams.push_back(st);
        p->joined.push_back(ev);
        // Plans of one step array share the graph image and one claim on the Infinity Cache.  A lane's pass 1 runs fewer
        // persistent workgroups than there are CUs -- half of them with three calls in flight or more, eleven sixteenths
        // with two -- so that the kernels of the other calls share the chip with it all the time, not only at its tail
        // (cfg-L, three in flight: 0.116 -> 0.107 ms per call; 4 M segments 0.165 -> 0.141; a call alone would be slower:
        // 0.133 -> 0.168).  Graphs beyond 2^28 steps, whose pass 1 is bound by instruction issue as much as by memory, take
        // eleven sixteenths whatever is in flight (ninety paths of ten million steps 2.17 -> 1.89 ms; sixteen thousand
        // haplotype walks even; half the chip loses there: profiles/NOTES.md R5.10).
        const int n_cus = device_cus(p->device);
        const uint32_t wgs = calls_in_flight < 2 ? 0u : (uint32_t)(calls_in_flight >= 3 && g->n_steps <= (1ull << 28) ? n_cus / 2 : n_cus * 11 / 16);
        flatgfa_dev_plan_t *pl = plan_create_impl(g, hb, he, wgs);
        if (!pl) { flatgfa_dev_pipeline_destroy(p); return nullptr; }
        p->plans.push_back(pl);
    }
    if (hipEventCreateWithFlags(&p->after, hipEventDisableTiming) != hipSuccess) { set_error("dev_pipeline_create: cannot create an event"); flatgfa_dev_pipeline_destroy(p); return nullptr; }
    return p;
}

extern "C" int flatgfa_dev_pipeline_seg_depth(flatgfa_dev_pipeline_t *p, uint32_t *depth_out, uint32_t *uniq_out, void *after_stream) {
    if (!p || p->plans.empty()) { set_error("dev_pipeline_seg_depth: NULL pipeline"); return FLATGFA_ERR_ARG; }
    const size_t lane = (size_t)(p->n_calls % p->plans.size());
    if (after_stream != (void *)-1) {  // (-1: nothing to wait for)
        HIP_TRY(hipEventRecord(p->after, (hipStream_t)after_stream), return FLATGFA_ERR_HIP);
        HIP_TRY(hipStreamWaitEvent(p->streams[lane], p->after, 0), return FLATGFA_ERR_HIP);
    }
    p->n_calls += 1;
    return flatgfa_dev_seg_depth(p->plans[lane], depth_out, uniq_out, p->streams[lane]);
}

extern "C" int flatgfa_dev_pipeline_path_depth_all(flatgfa_dev_pipeline_t *p, uint32_t *depth_out, uint64_t *length_out, uint64_t *weighted_out,
                                                    void *after_stream) {
    if (!p || p->plans.empty()) { set_error("dev_pipeline_path_depth_all: NULL pipeline"); return FLATGFA_ERR_ARG; }
    const size_t lane = (size_t)(p->n_calls % p->plans.size());
    if (after_stream != (void *)-1) {
        HIP_TRY(hipEventRecord(p->after, (hipStream_t)after_stream), return FLATGFA_ERR_HIP);
        HIP_TRY(hipStreamWaitEvent(p->streams[lane], p->after, 0), return FLATGFA_ERR_HIP);
    }
    p->n_calls += 1;
    return flatgfa_dev_path_depth_all(p->plans[lane], depth_out, length_out, weighted_out, p->streams[lane]);
}

extern "C" int flatgfa_dev_pipeline_join(flatgfa_dev_pipeline_t *p, void *stream) {
    if (!p) { set_error("dev_pipeline_join: NULL pipeline"); return FLATGFA_ERR_ARG; }
    for (size_t k = 0; k < p->streams.size(); ++k) {
        HIP_TRY(hipEventRecord(p->joined[k], p->streams[k]), return FLATGFA_ERR_HIP);
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, p->joined[k], 0), return FLATGFA_ERR_HIP);
    }
    return FLATGFA_OK;
}

extern "C" int flatgfa_dev_pipeline_status(flatgfa_dev_pipeline_t *p) {
    if (!p) { set_error("dev_pipeline_status: NULL pipeline"); return FLATGFA_ERR_ARG; }
    int rc = FLATGFA_OK;
    for (size_t k = 0; k < p->plans.size(); ++k) {
        const int r = flatgfa_dev_status(p->plans[k], p->streams[k]);  // (synchronizes the lane; completes its last call if that ran out of room)
        if (r != FLATGFA_OK && rc == FLATGFA_OK) rc = r;
    }
    return rc;
}

extern "C" int flatgfa_dev_pipeline_steps_changed(flatgfa_dev_pipeline_t *p) {
    if (!p) { set_error("dev_pipeline_steps_changed: NULL pipeline"); return FLATGFA_ERR_ARG; }
    int rc = FLATGFA_OK;
    for (size_t k = 0; k < p->plans.size(); ++k) {
        const int r = flatgfa_dev_plan_steps_changed(p->plans[k], p->streams[k]);
        if (r != FLATGFA_OK && rc == FLATGFA_OK) rc = r;
    }
    return rc;
}

extern "C" int flatgfa_dev_pipeline_describe(flatgfa_dev_pipeline_t *p, char *out, int cap) {
    if (!p || p->plans.empty() || !out || cap <= 0) return 0;
    std::string s = "calls_in_flight=" + std::to_string(p->plans.size()) + " ";
    char buf[768] = "";
    for (size_t k = p->plans.size(); k-- > 0;) (void)flatgfa_dev_plan_describe(p->plans[k], buf, (int)sizeof buf);  // (every lane's marks are waited for; the first lane's description is given)
    s += buf;
    const int n = (int)std::min<size_t>(s.size(), (size_t)cap - 1);
    memcpy(out, s.data(), (size_t)n);
    out[n] = 0;
    return n;
}

extern "C" int flatgfa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
