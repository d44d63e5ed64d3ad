// Path-pair overlap on gfx950 (BASELINE.json configs[4]).
//
// Semantics (cucapra/pollen slow_odgi/slow_odgi/overlap.py:6-14): path q touches path p when
// they are different paths and their sets of ORIENTED handles intersect.  The reference has no
// Rust implementation; slow_odgi is the reference.
//
// Memory follows the queries, not the paths: an exact bitset over all handles costs S/4 bytes per
// path (25 GB for 100 k paths over 1 M segments), so it is built only for the QUERY paths of a
// call, a batch at a time.  Every path additionally gets a coarse bitmap -- one bit per block of
// 2048 handles, 128 bytes at 1 M segments -- built once and kept with the plan.
//
//   k_coarse_bits   one workgroup per path: which blocks of 2048 handles the path has a step in
//   k_handle_bits   one workgroup per (query, orientation): the handles of that orientation the
//                   query uses, as a bitset over segment ids built in LDS (1 bit per segment) and
//                   written to HBM once -> qbits[query][orientation][words]
//   k_pair_touch    one wave per (query, candidate path).  The two coarse bitmaps are ANDed first:
//                   paths that share no block cannot touch (two loads per lane; this settles
//                   nearly every pair of paths that live in different regions of the graph).
//                   Otherwise, when every path has an exact bitset (they fit 1 GB), the two are ANDed
//                   block by block over the common blocks only; when only the queries have one,
//                   the candidate's steps are walked 64 at a time, each probing the query's exact
//                   bitset -- steps in blocks the query never enters are skipped without a probe.
//                   Either way the wave stops at the first common handle.
//
// Integer/bit work only; results are exact by construction (the coarse test has no false
// negatives: a common handle lies in a common block).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "../../include/flatgfa.h"
#include "device_common.hpp"
#include "prof.hpp"

namespace fgfa_dev {
namespace {

// The steps [b, e) of a path, every one exactly once, four per 16-byte load where the array allows it (an aligned
// base; the few steps before the first and behind the last 16-byte boundary one by one): these kernels read every
// step of the graph once per plan, and one 4-byte load per thread and step ran at an eighth of the memory's rate.
template <int THREADS, typename F>
__device__ __forceinline__ void for_each_step(const uint32_t *__restrict__ steps, uint64_t b, uint64_t e, F &&f) {
    if ((reinterpret_cast<uintptr_t>(steps) & 15u) != 0) {
        for (uint64_t i = b + threadIdx.x; i < e; i += THREADS) f(steps[i]);
        return;
    }
    const uint64_t v0 = min((b + 3) & ~(uint64_t)3, e), v1 = max(v0, e & ~(uint64_t)3);
    for (uint64_t i = b + threadIdx.x; i < v0; i += THREADS) f(steps[i]);
    const uint4 *p = reinterpret_cast<const uint4 *>(steps);
    for (uint64_t i = v0 / 4 + threadIdx.x; i < v1 / 4; i += THREADS) {
        const uint4 v = p[i];
        f(v.x);
        f(v.y);
        f(v.z);
        f(v.w);
    }
    for (uint64_t i = v1 + threadIdx.x; i < e; i += THREADS) f(steps[i]);
}

constexpr int kBitsThreads = 1024;
constexpr uint32_t kBitsWinWords = 36864;  // 144 KiB of LDS: 1,179,648 segments per pass
constexpr uint32_t kBlockBits = 11;        // a coarse block = 2048 handles

__global__ __launch_bounds__(kBitsThreads) void k_handle_bits(const uint32_t *__restrict__ steps,
                                                               const uint32_t *__restrict__ path_begin,
                                                               const uint32_t *__restrict__ path_end, uint32_t n_paths,
                                                               const uint32_t *__restrict__ query_ids, uint32_t n_q,
                                                               uint32_t n_segs, uint32_t words, uint32_t *__restrict__ bits,
                                                               uint32_t *__restrict__ status) {
    extern __shared__ uint32_t seen[];
    const uint32_t n_win = (words + kBitsWinWords - 1) / kBitsWinWords;
    const uint64_t jobs = (uint64_t)n_q * 2u * n_win;
    for (uint64_t job = blockIdx.x; job < jobs; job += gridDim.x) {
        const uint32_t win = (uint32_t)(job % n_win);
        const uint32_t orient = (uint32_t)((job / n_win) & 1u);
        const uint32_t k = (uint32_t)(job / (2ull * n_win));
        const uint32_t p = query_ids ? query_ids[k] : k;  // no list: bitsets for all paths, in order
        const uint32_t w0 = win * kBitsWinWords, nw = min(kBitsWinWords, words - w0);
        for (uint32_t i = threadIdx.x; i < nw; i += kBitsThreads) seen[i] = 0u;
        __syncthreads();
        if (p < n_paths) {
            for_each_step<kBitsThreads>(steps, path_begin[p], path_end[p], [&](uint32_t h) {
                const uint32_t seg = h >> 1;
                if (seg >= n_segs) {
                    *status = 1u;
                    return;
                }
                const uint32_t w = (seg >> 5) - w0;  // wraps below the window; the compare rejects it
                if ((h & 1u) == orient && w < nw) atomicOr(&seen[w], 1u << (seg & 31u));
            });
        } else if (threadIdx.x == 0) {
            *status = 1u;
        }
        __syncthreads();
        uint32_t *dst = bits + ((size_t)k * 2u + orient) * words + w0;
        for (uint32_t i = threadIdx.x; i < nw; i += kBitsThreads) dst[i] = seen[i];
        __syncthreads();
    }
}

constexpr int kCoarseThreads = 256;
constexpr uint32_t kCoarseMaxWords = 8192;  // LDS staging: 2^18 blocks = 2^29 handles

__global__ __launch_bounds__(kCoarseThreads) void k_coarse_bits(const uint32_t *__restrict__ steps,
                                                                 const uint32_t *__restrict__ path_begin,
                                                                 const uint32_t *__restrict__ path_end, uint32_t n_paths,
                                                                 uint32_t n_segs, uint32_t cwords, uint32_t *__restrict__ coarse,
                                                                 uint32_t *__restrict__ status) {
    __shared__ uint32_t blk[kCoarseMaxWords];
    for (uint32_t p = blockIdx.x; p < n_paths; p += gridDim.x) {
        for (uint32_t i = threadIdx.x; i < cwords; i += kCoarseThreads) blk[i] = 0u;
        __syncthreads();
        for_each_step<kCoarseThreads>(steps, path_begin[p], path_end[p], [&](uint32_t h) {
            if ((h >> 1) >= n_segs) {
                *status = 1u;
                return;
            }
            const uint32_t c = h >> kBlockBits;
            atomicOr(&blk[c >> 5], 1u << (c & 31u));
        });
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cwords; i += kCoarseThreads) coarse[(size_t)p * cwords + i] = blk[i];
        __syncthreads();
    }
}

constexpr int kPairThreads = 256;

__global__ __launch_bounds__(kPairThreads) void k_pair_touch(const uint32_t *__restrict__ steps,
                                                              const uint32_t *__restrict__ path_begin,
                                                              const uint32_t *__restrict__ path_end,
                                                              const uint32_t *__restrict__ qbits, uint32_t words,
                                                              const uint32_t *__restrict__ coarse, uint32_t cwords,
                                                              const uint32_t *__restrict__ query_ids, uint32_t n_q,
                                                              uint32_t n_paths, uint32_t n_segs, uint32_t by_path,
                                                              uint8_t *__restrict__ out, uint32_t *__restrict__ status) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * (kPairThreads / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (kPairThreads / 64);
    const uint64_t pairs = (uint64_t)n_q * n_paths;
    for (uint64_t pair = wave0; pair < pairs; pair += n_waves) {
        const uint32_t k = (uint32_t)(pair / n_paths), j = (uint32_t)(pair % n_paths);
        const uint32_t ip = query_ids[k];
        bool touch = false;
        if (ip >= n_paths) {
            if (lane == 0) *status = 1u;
        } else if (ip != j) {  // overlap.py:10-11: a path does not touch itself
            // 1. no block of handles in common: cannot touch
            const uint32_t *ca = coarse + (size_t)ip * cwords, *cb = coarse + (size_t)j * cwords;
            bool common = false;
            for (uint32_t i = lane; i < ((cwords + 63u) & ~63u) && !common; i += 64)
                common = __builtin_amdgcn_ballot_w64(i < cwords && (ca[i] & cb[i]) != 0u) != 0ull;
            if (common && by_path) {
                // 2a. both paths have exact bitsets: AND them block by block, common blocks only.  A
                //     block of 2048 handles is 32 words of each orientation's bitset: one word per lane.
                const uint32_t *qa = qbits + (size_t)ip * 2u * words, *qc = qbits + (size_t)j * 2u * words;
                const uint32_t wofs = (uint32_t)(lane >> 5) * words + ((uint32_t)lane & 31u);
                for (uint32_t cw = 0; cw < cwords && !touch; ++cw) {
                    uint32_t m = __builtin_amdgcn_readfirstlane(ca[cw] & cb[cw]);
                    while (m) {
                        const uint32_t c = cw * 32u + (uint32_t)__builtin_ctz(m);
                        m &= m - 1u;
                        const uint32_t w = c * 32u + ((uint32_t)lane & 31u);  // word of either orientation's bitset
                        const bool hit = w < words && (qa[wofs + c * 32u] & qc[wofs + c * 32u]) != 0u;
                        if (__builtin_amdgcn_ballot_w64(hit)) {
                            touch = true;
                            break;
                        }
                    }
                }
            } else if (common) {
                // 2b. walk the candidate's steps against the query's exact bitset
                const uint32_t *qb = qbits + (size_t)k * 2u * words;
                const uint64_t b = path_begin[j], e = path_end[j];
                for (uint64_t i = b + lane; i < ((e - b + 63u) & ~(uint64_t)63u) + b; i += 64) {
                    bool hit = false;
                    if (i < e) {
                        const uint32_t h = steps[i], seg = h >> 1, c = h >> kBlockBits;
                        if (seg < n_segs && ((ca[c >> 5] >> (c & 31u)) & 1u))
                            hit = ((qb[(size_t)(h & 1u) * words + (seg >> 5)] >> (seg & 31u)) & 1u) != 0u;
                    }
                    if (__builtin_amdgcn_ballot_w64(hit)) {
                        touch = true;
                        break;
                    }
                }
            }
        }
        if (lane == 0) out[pair] = touch ? 1 : 0;
    }
}

}  // namespace

}  // namespace fgfa_dev

using namespace fgfa_dev;

// Declared in depth_device.hip's plan; kept here to keep the overlap code in one place.
// `coarse_cache`: the per-path coarse bitmaps, built on the first call and kept with the plan.
// `qbits_cache` / `qbits_bytes`: scratch for the queries' exact bitsets, grown on demand, kept too.
extern "C" int flatgfa_dev_path_overlaps_impl(const flatgfa_dev_graph_t *g, int n_cus, uint32_t **coarse_cache,
                                              uint32_t **qbits_cache, size_t *qbits_bytes, bool *qbits_all, const uint32_t *query_ids, uint32_t n_q, uint8_t *touch_out,
                                              uint32_t *status, hipStream_t stream) {
    if (n_q == 0 || g->n_paths == 0) return FLATGFA_OK;
    const uint32_t words = (((g->n_segs + 31u) / 32u) + 3u) & ~3u;
    if (words == 0) {
        if (hipMemsetAsync(touch_out, 0, (size_t)n_q * g->n_paths, stream) != hipSuccess) return FLATGFA_ERR_HIP;
        return FLATGFA_OK;
    }
    const uint64_t n_blocks = (2ull * g->n_segs + (1u << kBlockBits) - 1) >> kBlockBits;
    const uint32_t cwords = (uint32_t)((n_blocks + 31) / 32);
    if (cwords > kCoarseMaxWords) {
        set_error("path overlaps: more than 2^28 segments");
        return FLATGFA_ERR_TOO_LARGE;
    }
    if (!*coarse_cache) {
        if (hipMalloc(coarse_cache, (size_t)g->n_paths * cwords * 4u) != hipSuccess) {
            set_error("path overlaps: cannot allocate the per-path coarse bitmaps");
            return FLATGFA_ERR_HIP;
        }
        ProfScope ps("k_coarse_bits", stream);
        hipLaunchKernelGGL(k_coarse_bits, dim3(std::min<uint32_t>(g->n_paths, (uint32_t)n_cus * 16u)), dim3(kCoarseThreads), 0,
                           stream, g->steps, g->path_begin, g->path_end, g->n_paths, g->n_segs, cwords, *coarse_cache, status);
        if (hipGetLastError() != hipSuccess) {  // never keep bitmaps that were not built
            (void)hipFree(*coarse_cache);
            *coarse_cache = nullptr;
            set_error("path overlaps: kernel launch failed");
            return FLATGFA_ERR_HIP;
        }
    }
    // Exact bitsets: for every path, built once and kept, when that takes at most 1 GB; otherwise
    // for the queries of this call only, a batch (of at most 1 GB) at a time.
    const uint64_t per_query = 2ull * words * 4u;
    uint64_t dense_max = 1ull << 30;
    if (const char *f = test_hook("FLATGFA_OVERLAP_DENSE_MAX")) dense_max = strtoull(f, nullptr, 10);  // tests: 0 = query bitsets only
    const bool all_paths = (uint64_t)g->n_paths * per_query <= dense_max;
    const uint32_t batch = all_paths ? g->n_paths : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_q, (1ull << 30) / per_query));
    // (what the cache holds is recorded, not inferred from its size: bitsets of every path in path
    // order, or those of the last call's queries in query order)
    bool build_all = false;
    if (*qbits_bytes >= (size_t)batch * per_query && all_paths && !*qbits_all) build_all = true;  // big enough, but laid out by query
    if (*qbits_bytes < (size_t)batch * per_query) {
        if (*qbits_cache) {
            (void)hipStreamSynchronize(stream);  // an earlier call on this plan may still be reading it
            (void)hipFree(*qbits_cache);
            *qbits_cache = nullptr;
            *qbits_bytes = 0;
        }
        if (hipMalloc(qbits_cache, (size_t)batch * per_query) != hipSuccess) {
            set_error("path overlaps: cannot allocate the handle bitsets");
            return FLATGFA_ERR_HIP;
        }
        *qbits_bytes = (size_t)batch * per_query;
        build_all = all_paths;
    }
    *qbits_all = all_paths;
    uint32_t *qbits = *qbits_cache;
    const uint32_t lds = std::min(words, kBitsWinWords) * 4u;
    (void)hipFuncSetAttribute((const void *)k_handle_bits, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const uint32_t n_win = (words + kBitsWinWords - 1) / kBitsWinWords;
    int rc = FLATGFA_OK;
    for (uint32_t q0 = 0; q0 < n_q && rc == FLATGFA_OK; q0 += all_paths ? n_q : batch) {
        const uint32_t nq = all_paths ? n_q : std::min(batch, n_q - q0);
        if (!all_paths || build_all) {
            const uint32_t nb = all_paths ? g->n_paths : nq;
            const uint64_t jobs = (uint64_t)nb * 2u * n_win;
            ProfScope ps("k_handle_bits", stream);
            hipLaunchKernelGGL(k_handle_bits, dim3((uint32_t)std::min<uint64_t>(jobs, (uint64_t)n_cus * 16u)),
                               dim3(kBitsThreads), lds, stream, g->steps, g->path_begin, g->path_end, g->n_paths,
                               all_paths ? (const uint32_t *)nullptr : query_ids + q0, nb, g->n_segs, words, qbits, status);
        }
        {
            const uint64_t pairs = (uint64_t)nq * g->n_paths;
            const uint64_t blocks = (pairs + (kPairThreads / 64) - 1) / (kPairThreads / 64);
            ProfScope ps("k_pair_touch", stream);
            hipLaunchKernelGGL(k_pair_touch, dim3((uint32_t)std::min<uint64_t>(blocks, (uint64_t)n_cus * 64u)),
                               dim3(kPairThreads), 0, stream, g->steps, g->path_begin, g->path_end, qbits, words,
                               *coarse_cache, cwords, query_ids + q0, nq, g->n_paths, g->n_segs, all_paths ? 1u : 0u,
                               touch_out + (size_t)q0 * g->n_paths, status);
        }
        if (hipGetLastError() != hipSuccess) {
            set_error("path overlaps: kernel launch failed");
            rc = FLATGFA_ERR_HIP;
            if (all_paths) {  // never keep bitsets that may not have been built
                (void)hipFree(*qbits_cache);
                *qbits_cache = nullptr;
                *qbits_bytes = 0;
                *qbits_all = false;
            }
        }
    }
    return rc;
}
