// Path-pair overlap on gfx950 (BASELINE.json configs[4]).
//
// Semantics (cucapra/pollen slow_odgi/slow_odgi/overlap.py:6-14): path q touches path p when
// they are different paths and their sets of ORIENTED handles intersect.  The reference has no
// Rust implementation; slow_odgi is the reference.
//
//   k_handle_bits   one workgroup per (path, orientation): the handles of that orientation the
//                   path uses, as a bitset over segment ids built in LDS (1 bit per segment) and
//                   written to HBM once -> bits[path][orientation][words]
//   k_pair_touch    one wave per (query path, candidate path): ANDs the two bitsets 16 bytes per
//                   lane at a time and stops at the first common bit (wave ballot)
//
// Integer/bit work only; results are exact by construction.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/flatgfa.h"
#include "device_common.hpp"
#include "prof.hpp"

namespace fgfa_dev {
namespace {

constexpr int kBitsThreads = 1024;
constexpr uint32_t kBitsWinWords = 36864;  // 144 KiB of LDS: 1,179,648 segments per pass

__global__ __launch_bounds__(kBitsThreads) void k_handle_bits(const uint32_t *__restrict__ steps,
                                                               const uint32_t *__restrict__ path_begin,
                                                               const uint32_t *__restrict__ path_end, uint32_t n_paths,
                                                               uint32_t n_segs, uint32_t words, uint32_t *__restrict__ bits,
                                                               uint32_t *__restrict__ status) {
    extern __shared__ uint32_t seen[];
    const uint32_t n_win = (words + kBitsWinWords - 1) / kBitsWinWords;
    const uint64_t jobs = (uint64_t)n_paths * 2u * n_win;
    for (uint64_t job = blockIdx.x; job < jobs; job += gridDim.x) {
        const uint32_t win = (uint32_t)(job % n_win);
        const uint32_t orient = (uint32_t)((job / n_win) & 1u);
        const uint32_t p = (uint32_t)(job / (2ull * n_win));
        const uint32_t w0 = win * kBitsWinWords, nw = min(kBitsWinWords, words - w0);
        for (uint32_t i = threadIdx.x; i < nw; i += kBitsThreads) seen[i] = 0u;
        __syncthreads();
        const uint32_t b = path_begin[p], e = path_end[p];
        for (uint64_t i = (uint64_t)b + threadIdx.x; i < e; i += kBitsThreads) {
            const uint32_t h = steps[i], seg = h >> 1;
            if (seg >= n_segs) {
                *status = 1u;
                continue;
            }
            const uint32_t w = (seg >> 5) - w0;  // wraps below the window; the compare rejects it
            if ((h & 1u) == orient && w < nw) atomicOr(&seen[w], 1u << (seg & 31u));
        }
        __syncthreads();
        uint32_t *dst = bits + ((size_t)p * 2u + orient) * words + w0;
        for (uint32_t i = threadIdx.x; i < nw; i += kBitsThreads) dst[i] = seen[i];
        __syncthreads();
    }
}

constexpr int kPairThreads = 256;

__global__ __launch_bounds__(kPairThreads) void k_pair_touch(const uint32_t *__restrict__ bits, uint32_t words2,
                                                              const uint32_t *__restrict__ query_ids, uint32_t n_q,
                                                              uint32_t n_paths, uint8_t *__restrict__ out,
                                                              uint32_t *__restrict__ status) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * (kPairThreads / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (kPairThreads / 64);
    const uint64_t pairs = (uint64_t)n_q * n_paths;
    const uint32_t n4 = words2 / 4;  // words2 is a multiple of 4
    for (uint64_t pair = wave0; pair < pairs; pair += n_waves) {
        const uint32_t k = (uint32_t)(pair / n_paths), j = (uint32_t)(pair % n_paths);
        const uint32_t ip = query_ids[k];
        bool touch = false;
        if (ip >= n_paths) {
            if (lane == 0) *status = 1u;
        } else if (ip != j) {  // overlap.py:10-11: a path does not touch itself
            const uint4 *a = reinterpret_cast<const uint4 *>(bits + (size_t)ip * words2);
            const uint4 *b = reinterpret_cast<const uint4 *>(bits + (size_t)j * words2);
            for (uint32_t i = lane; i < ((n4 + 63u) & ~63u); i += 64) {
                bool hit = false;
                if (i < n4) {
                    const uint4 x = a[i], y = b[i];
                    hit = ((x.x & y.x) | (x.y & y.y) | (x.z & y.z) | (x.w & y.w)) != 0u;
                }
                if (__builtin_amdgcn_ballot_w64(hit)) {
                    touch = true;
                    break;
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
extern "C" int flatgfa_dev_path_overlaps_impl(const flatgfa_dev_graph_t *g, int n_cus, uint32_t **bits_cache,
                                              const uint32_t *query_ids, uint32_t n_q, uint8_t *touch_out,
                                              uint32_t *status, hipStream_t stream) {
    if (n_q == 0 || g->n_paths == 0) return FLATGFA_OK;
    const uint32_t words = (((g->n_segs + 31u) / 32u) + 3u) & ~3u;
    if (words == 0) {
        if (hipMemsetAsync(touch_out, 0, (size_t)n_q * g->n_paths, stream) != hipSuccess) return FLATGFA_ERR_HIP;
        return FLATGFA_OK;
    }
    if (!*bits_cache) {
        // the per-path handle bitsets depend only on the graph: build once, keep with the plan
        if (hipMalloc(bits_cache, (size_t)g->n_paths * 2u * words * 4u) != hipSuccess) {
            set_error("path overlaps: cannot allocate the per-path handle bitsets");
            return FLATGFA_ERR_HIP;
        }
        const uint32_t lds = std::min(words, kBitsWinWords) * 4u;
        (void)hipFuncSetAttribute((const void *)k_handle_bits, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const uint32_t n_win = (words + kBitsWinWords - 1) / kBitsWinWords;
        const uint64_t jobs = (uint64_t)g->n_paths * 2u * n_win;
        ProfScope ps("k_handle_bits", stream);
        hipLaunchKernelGGL(k_handle_bits, dim3((uint32_t)std::min<uint64_t>(jobs, (uint64_t)n_cus * 16u)),
                           dim3(kBitsThreads), lds, stream, g->steps, g->path_begin, g->path_end, g->n_paths, g->n_segs,
                           words, *bits_cache, status);
        if (hipGetLastError() != hipSuccess) {  // never keep bitsets that were not built
            (void)hipFree(*bits_cache);
            *bits_cache = nullptr;
            set_error("path overlaps: kernel launch failed");
            return FLATGFA_ERR_HIP;
        }
    }
    {
        const uint64_t pairs = (uint64_t)n_q * g->n_paths;
        const uint64_t blocks = (pairs + (kPairThreads / 64) - 1) / (kPairThreads / 64);
        ProfScope ps("k_pair_touch", stream);
        hipLaunchKernelGGL(k_pair_touch, dim3((uint32_t)std::min<uint64_t>(blocks, (uint64_t)n_cus * 64u)),
                           dim3(kPairThreads), 0, stream, *bits_cache, 2u * words, query_ids, n_q, g->n_paths, touch_out,
                           status);
    }
    if (hipGetLastError() != hipSuccess) {
        set_error("path overlaps: kernel launch failed");
        return FLATGFA_ERR_HIP;
    }
    return FLATGFA_OK;
}
