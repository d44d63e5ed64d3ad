// Design-space microbenchmarks for the step-scan kernel on gfx950 (not part of the product).
// Builds against libflatgfa.so only for the synthetic generator.  Each variant is timed with
// HIP events (median of REPS) on the cfg-L graph (1M segments / 100M steps) and, where it
// produces a result, checked against a host count.
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -Iinclude -Lpollen_amd/lib -lflatgfa \
//         -Wl,-rpath,$PWD/pollen_amd/lib -o tools/microbench && tools/microbench [pangenome|uniform]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "flatgfa.h"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int REPS = 7;

template <class F>
static float time_ms(F &&launch, hipStream_t s = nullptr) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();  // warm-up
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < REPS; ++r) {
        CK(hipEventRecord(a, s));
        launch();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return t[t.size() / 2];
}

// ------------------------------------------------------------------ M1: streaming reads ---
__global__ __launch_bounds__(256) void k_stream_u32(const uint32_t *__restrict__ s, uint64_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc ^= s[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream_u32x4(const uint4 *__restrict__ s, uint64_t n4, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
        uint4 v = s[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// ------------------------------------------------------------ M2: naive global atomics ---
__global__ __launch_bounds__(256) void k_hist_naive(const uint32_t *__restrict__ s, uint64_t n, uint32_t *hist) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        atomicAdd(&hist[s[i] >> 1], 1u);
}
// same, but each block owns a contiguous chunk (consecutive steps stay in one wave over time)
__global__ __launch_bounds__(256) void k_hist_naive_chunk(const uint32_t *__restrict__ s, uint64_t n, uint32_t chunk,
                                                           uint32_t *hist) {
    uint64_t b = (uint64_t)blockIdx.x * chunk, e = min(b + chunk, n);
    for (uint64_t i = b + threadIdx.x; i < e; i += 256) atomicAdd(&hist[s[i] >> 1], 1u);
}
// sequential ids: the best case for whatever coalescing the atomic path has
__global__ __launch_bounds__(256) void k_hist_seq(uint64_t n, uint32_t nsegs, uint32_t *hist) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        atomicAdd(&hist[(uint32_t)(i % nsegs)], 1u);
}

// ------------------------------------------- M3: difference-array depth (run detection) ---
// Each wave walks a contiguous span of a path in 64-step tiles.  A step starts a run when its
// segment id is not its predecessor's + 1; only run boundaries touch memory:
//   diff[id] += 1 at a run start, diff[prev + 1] -= 1 where the previous run ended.
// depth = prefix_sum(diff).  (diff has nsegs + 1 entries.)
template <bool COUNT_ONLY>
__global__ __launch_bounds__(1024) void k_depth_diff(const uint32_t *__restrict__ steps, const uint32_t *__restrict__ pb,
                                                      const uint32_t *__restrict__ pe, uint32_t npaths, int *diff,
                                                      unsigned long long *n_atomics) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned long long cnt = 0;
    for (uint32_t p = blockIdx.x; p < npaths; p += gridDim.x) {
        const uint64_t b = pb[p], e = pe[p], n = e - b;
        const uint64_t per = ((n + nw - 1) / nw + 63) / 64 * 64;  // whole tiles per wave
        const uint64_t lo = min(b + per * wave, e), hi = min(lo + per, e);
        uint32_t carry = 0;
        bool have_carry = false;
        for (uint64_t t = lo; t < hi; t += 64) {
            const uint64_t i = t + lane;
            const bool live = i < hi;
            const uint32_t id = live ? (steps[i] >> 1) : 0u;
            uint32_t prev = __shfl_up(id, 1, 64);
            bool has_prev = true;
            if (lane == 0) {
                prev = carry;
                has_prev = have_carry;
            }
            const bool start = live && (!has_prev || id != prev + 1);
            if (start) {
                if (COUNT_ONLY) {
                    cnt += has_prev ? 2 : 1;
                } else {
                    atomicAdd(&diff[id], 1);
                    if (has_prev) atomicAdd(&diff[prev + 1], -1);
                }
            }
            // last live lane's id carries into the next tile
            const unsigned long long lm = __ballot(live);
            const int last = 63 - __builtin_clzll(lm);
            carry = __shfl(id, last, 64);
            have_carry = true;
        }
        if (have_carry && lane == 0) {
            if (COUNT_ONLY) cnt += 1;
            else atomicAdd(&diff[carry + 1], -1);
        }
    }
    if (COUNT_ONLY) atomicAdd(n_atomics, cnt);
}

// --------------------------------------------------------- M4: per-path LDS bitmap (uniq) ---
constexpr uint32_t WIN_WORDS = 32768;  // 128 KiB

// MODE 0: returning atomicOr per step + global atomic on first visit   (the v1 product kernel)
// MODE 1: non-returning atomicOr per step, nothing global               (LDS cost alone)
// MODE 2: run-based: one lane per run sets <= 3 words, nothing global
// MODE 3: MODE 2 + final bitmap scan emitting interval transitions into a global diff array
// MODE 4: MODE 2 + final bitmap dump (128 KiB per path) to global
template <int MODE>
__global__ __launch_bounds__(1024) void k_uniq(const uint32_t *__restrict__ steps, const uint32_t *__restrict__ pb,
                                                const uint32_t *__restrict__ pe, uint32_t npaths, uint32_t nsegs,
                                                uint32_t *uniq, int *udiff, uint32_t *dump) {
    extern __shared__ uint32_t seen[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t nwords = (nsegs + 31) >> 5;
    for (uint32_t p = blockIdx.x; p < npaths; p += gridDim.x) {
        for (uint32_t w = threadIdx.x; w < nwords; w += blockDim.x) seen[w] = 0;
        __syncthreads();
        const uint64_t b = pb[p], e = pe[p], n = e - b;
        const uint64_t per = ((n + nw - 1) / nw + 63) / 64 * 64;
        const uint64_t lo = min(b + per * wave, e), hi = min(lo + per, e);
        for (uint64_t t = lo; t < hi; t += 64) {
            const uint64_t i = t + lane;
            const bool live = i < hi;
            const uint32_t id = live ? (steps[i] >> 1) : 0xFFFFFFFFu;
            if (MODE == 0) {
                if (live) {
                    const uint32_t bit = 1u << (id & 31);
                    const uint32_t old = atomicOr(&seen[id >> 5], bit);
                    if (!(old & bit)) atomicAdd(&uniq[id], 1u);
                }
            } else if (MODE == 1) {
                if (live) atomicOr(&seen[id >> 5], 1u << (id & 31));
            } else {
                const uint32_t prev = __shfl_up(id, 1, 64);
                const bool start = live && (lane == 0 || id != prev + 1);
                const unsigned long long sm = __ballot(start);
                const unsigned long long lm = __ballot(live);
                if (start) {
                    const unsigned long long above = (lane == 63) ? 0ull : (sm >> (lane + 1)) << (lane + 1);
                    const int nlive = __builtin_popcountll(lm);
                    const int next = above ? __builtin_ctzll(above) : nlive;
                    uint32_t first = id, last = id + (uint32_t)(next - lane) - 1;  // inclusive id range
                    for (uint32_t w = first >> 5; w <= (last >> 5); ++w) {
                        const uint32_t lo_bit = (w == (first >> 5)) ? (first & 31) : 0;
                        const uint32_t hi_bit = (w == (last >> 5)) ? (last & 31) : 31;
                        const uint32_t m = (hi_bit == 31 ? 0xFFFFFFFFu : ((1u << (hi_bit + 1)) - 1u)) & ~((1u << lo_bit) - 1u);
                        atomicOr(&seen[w], m);
                    }
                }
            }
        }
        __syncthreads();
        if (MODE == 3) {
            // interval transitions of the visited set: +1 where a visited run starts, -1 just past its end
            for (uint32_t w = threadIdx.x; w < nwords; w += blockDim.x) {
                const uint32_t cur = seen[w];
                const uint32_t below = w ? (seen[w - 1] >> 31) : 0u;
                const uint32_t shifted = (cur << 1) | below;       // bit k = visited(k-1)
                uint32_t starts = cur & ~shifted, ends = ~cur & shifted;
                while (starts) {
                    const int k = __builtin_ctz(starts);
                    starts &= starts - 1;
                    atomicAdd(&udiff[w * 32 + k], 1);
                }
                while (ends) {
                    const int k = __builtin_ctz(ends);
                    ends &= ends - 1;
                    atomicAdd(&udiff[w * 32 + k], -1);
                }
            }
            if (threadIdx.x == 0 && (nsegs & 31) == 0 && (seen[nwords - 1] >> 31)) atomicAdd(&udiff[nsegs], -1);
            __syncthreads();
        }
        if (MODE == 4) {
            uint32_t *dst = dump + (size_t)p * nwords;
            for (uint32_t w = threadIdx.x; w < nwords; w += blockDim.x) dst[w] = seen[w];
            __syncthreads();
        }
    }
}

// column sums of the dumped bitmaps: uniq[32w + k] = sum_p bit k of dump[p][w]
__global__ __launch_bounds__(256) void k_colsum(const uint32_t *__restrict__ dump, uint32_t npaths, uint32_t nwords,
                                                 uint32_t *uniq) {
    // one wave per 2 words: lane -> (word, bit); paths striped over the block's 4 waves... keep it simple:
    // thread = one word, loops over paths and keeps 32 counters in registers via bit-sliced adds
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    uint32_t plane[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // vertical counters, up to 1023 paths
    for (uint32_t p = 0; p < npaths; ++p) {
        uint32_t carry = dump[(size_t)p * nwords + w];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const uint32_t t = plane[k] & carry;
            plane[k] ^= carry;
            carry = t;
        }
    }
    for (int bit = 0; bit < 32; ++bit) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) c |= ((plane[k] >> bit) & 1u) << k;
        uniq[w * 32 + bit] = c;
    }
}

// inclusive prefix sum of a diff array (single block; only for checking, not timed as a product kernel)
__global__ void k_prefix_check(const int *diff, uint32_t n, uint32_t *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        for (uint32_t i = 0; i < n; ++i) {
            acc += diff[i];
            out[i] = (uint32_t)acc;
        }
    }
}

int main(int argc, char **argv) {
    const int model = (argc > 1 && !strcmp(argv[1], "uniform")) ? 1 : 0;
    const uint32_t S = 1000000, P = 1000, L = 100000;
    const uint64_t N = (uint64_t)P * L;
    printf("# microbench model=%s S=%u P=%u L=%u N=%llu\n", model ? "uniform" : "pangenome", S, P, L, (unsigned long long)N);
    flatgfa_t g = flatgfa_synth(1, S, P, L, model, false);
    const void *sp;
    uint64_t sn;
    flatgfa_pool(g, 4, &sp, &sn, nullptr);
    const uint32_t *hsteps = (const uint32_t *)sp;
    std::vector<uint32_t> hpb(P), hpe(P);
    for (uint32_t p = 0; p < P; ++p) { hpb[p] = p * L; hpe[p] = (p + 1) * L; }

    // host reference
    std::vector<uint32_t> ref_d(S, 0), ref_u(S, 0);
    {
        std::vector<uint32_t> stamp(S, 0xFFFFFFFFu);
        for (uint32_t p = 0; p < P; ++p)
            for (uint64_t i = hpb[p]; i < hpe[p]; ++i) {
                uint32_t id = hsteps[i] >> 1;
                ref_d[id]++;
                if (stamp[id] != p) { stamp[id] = p; ref_u[id]++; }
            }
    }

    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s CUs=%d clock=%d MHz L2=%d MB\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.l2CacheSize >> 20);

    uint32_t *d_steps, *d_pb, *d_pe, *d_hist, *d_out, *d_uniq, *d_dump;
    int *d_diff;
    unsigned long long *d_cnt;
    CK(hipMalloc(&d_steps, N * 4));
    CK(hipMemcpy(d_steps, hsteps, N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pb, P * 4));
    CK(hipMalloc(&d_pe, P * 4));
    CK(hipMemcpy(d_pb, hpb.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pe, hpe.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hist, (S + 64) * 4));
    CK(hipMalloc(&d_uniq, (S + 64) * 4));
    CK(hipMalloc(&d_diff, (S + 64) * 4));
    CK(hipMalloc(&d_out, (S + 64) * 4));
    CK(hipMalloc(&d_cnt, 8));
    const uint32_t nwords = (S + 31) / 32;
    CK(hipMalloc(&d_dump, (size_t)P * nwords * 4));
    std::vector<uint32_t> got(S);
    auto check = [&](const char *what, const uint32_t *dev, const std::vector<uint32_t> &ref) {
        CK(hipMemcpy(got.data(), dev, S * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (uint32_t i = 0; i < S; ++i) bad += got[i] != ref[i];
        printf("    check %-28s %s (%zu mismatches)\n", what, bad ? "FAIL" : "ok", bad);
    };
    auto report = [&](const char *name, float ms, double bytes) {
        printf("%-34s %9.3f ms  %8.1f Gsteps/s  %8.1f GB/s (algorithmic %.0f MB)\n", name, ms, N / ms / 1e6, bytes / ms / 1e6, bytes / 1e6);
        fflush(stdout);
    };
    const double Bsteps = 4.0 * N;

    for (int grid : {2048, 8192}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_stream_u32, dim3(grid), dim3(256), 0, 0, d_steps, N, d_out); });
        report((std::string("M1 stream_u32 grid=") + std::to_string(grid)).c_str(), ms, Bsteps);
        ms = time_ms([&] { hipLaunchKernelGGL(k_stream_u32x4, dim3(grid), dim3(256), 0, 0, (const uint4 *)d_steps, N / 4, d_out); });
        report((std::string("M1 stream_u32x4 grid=") + std::to_string(grid)).c_str(), ms, Bsteps);
    }

    float ms = time_ms([&] {
        CK(hipMemsetAsync(d_hist, 0, S * 4, 0));
        hipLaunchKernelGGL(k_hist_naive, dim3(4096), dim3(256), 0, 0, d_steps, N, d_hist);
    });
    report("M2 hist_naive gridstride", ms, Bsteps);
    check("hist_naive", d_hist, ref_d);
    ms = time_ms([&] {
        CK(hipMemsetAsync(d_hist, 0, S * 4, 0));
        hipLaunchKernelGGL(k_hist_naive_chunk, dim3((N + 16383) / 16384), dim3(256), 0, 0, d_steps, N, 16384u, d_hist);
    });
    report("M2 hist_naive chunk16k", ms, Bsteps);
    check("hist_naive_chunk", d_hist, ref_d);
    ms = time_ms([&] {
        CK(hipMemsetAsync(d_hist, 0, S * 4, 0));
        hipLaunchKernelGGL(k_hist_seq, dim3(4096), dim3(256), 0, 0, N, S, d_hist);
    });
    report("M2 hist_seq (no loads)", ms, 0);

    // M3
    CK(hipMemset(d_cnt, 0, 8));
    hipLaunchKernelGGL(k_depth_diff<true>, dim3(P), dim3(1024), 0, 0, d_steps, d_pb, d_pe, P, d_diff, d_cnt);
    unsigned long long cnt = 0;
    CK(hipMemcpy(&cnt, d_cnt, 8, hipMemcpyDeviceToHost));
    printf("    diff-array atomics: %llu (%.3f per step)\n", cnt, (double)cnt / N);
    ms = time_ms([&] {
        CK(hipMemsetAsync(d_diff, 0, (S + 1) * 4, 0));
        hipLaunchKernelGGL(k_depth_diff<false>, dim3(P), dim3(1024), 0, 0, d_steps, d_pb, d_pe, P, d_diff, d_cnt);
    });
    report("M3 depth_diff block/path 1024t", ms, Bsteps);
    hipLaunchKernelGGL(k_prefix_check, dim3(1), dim3(64), 0, 0, d_diff, S, d_out);
    check("depth_diff", d_out, ref_d);
    ms = time_ms([&] {
        hipLaunchKernelGGL(k_depth_diff<true>, dim3(P), dim3(1024), 0, 0, d_steps, d_pb, d_pe, P, d_diff, d_cnt);
    });
    report("M3 depth_diff COUNT_ONLY (no atomics)", ms, Bsteps);

    // M4
    CK(hipFuncSetAttribute((const void *)k_uniq<0>, hipFuncAttributeMaxDynamicSharedMemorySize, WIN_WORDS * 4));
    CK(hipFuncSetAttribute((const void *)k_uniq<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WIN_WORDS * 4));
    CK(hipFuncSetAttribute((const void *)k_uniq<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WIN_WORDS * 4));
    CK(hipFuncSetAttribute((const void *)k_uniq<3>, hipFuncAttributeMaxDynamicSharedMemorySize, WIN_WORDS * 4));
    CK(hipFuncSetAttribute((const void *)k_uniq<4>, hipFuncAttributeMaxDynamicSharedMemorySize, WIN_WORDS * 4));
    ms = time_ms([&] {
        CK(hipMemsetAsync(d_uniq, 0, S * 4, 0));
        hipLaunchKernelGGL(k_uniq<0>, dim3(P), dim3(1024), WIN_WORDS * 4, 0, d_steps, d_pb, d_pe, P, S, d_uniq, d_diff, d_dump);
    });
    report("M4.0 uniq rtn-or + global atomic", ms, Bsteps);
    check("uniq mode0", d_uniq, ref_u);
    ms = time_ms([&] { hipLaunchKernelGGL(k_uniq<1>, dim3(P), dim3(1024), WIN_WORDS * 4, 0, d_steps, d_pb, d_pe, P, S, d_uniq, d_diff, d_dump); });
    report("M4.1 uniq ds_or per step only", ms, Bsteps);
    ms = time_ms([&] { hipLaunchKernelGGL(k_uniq<2>, dim3(P), dim3(1024), WIN_WORDS * 4, 0, d_steps, d_pb, d_pe, P, S, d_uniq, d_diff, d_dump); });
    report("M4.2 uniq run-based ds_or only", ms, Bsteps);
    ms = time_ms([&] {
        CK(hipMemsetAsync(d_diff, 0, (S + 1) * 4, 0));
        hipLaunchKernelGGL(k_uniq<3>, dim3(P), dim3(1024), WIN_WORDS * 4, 0, d_steps, d_pb, d_pe, P, S, d_uniq, d_diff, d_dump);
    });
    report("M4.3 uniq run-based + interval diff", ms, Bsteps);
    hipLaunchKernelGGL(k_prefix_check, dim3(1), dim3(64), 0, 0, d_diff, S, d_out);
    check("uniq mode3 (interval diff)", d_out, ref_u);
    ms = time_ms([&] { hipLaunchKernelGGL(k_uniq<4>, dim3(P), dim3(1024), WIN_WORDS * 4, 0, d_steps, d_pb, d_pe, P, S, d_uniq, d_diff, d_dump); });
    report("M4.4 uniq run-based + bitmap dump", ms, Bsteps);
    float ms2 = time_ms([&] { hipLaunchKernelGGL(k_colsum, dim3((nwords + 255) / 256), dim3(256), 0, 0, d_dump, P, nwords, d_uniq); });
    report("M4.4b column sums of dumped bitmaps", ms2, (double)P * nwords * 4);
    check("uniq mode4 (dump+colsum)", d_uniq, ref_u);

    flatgfa_free(g);
    return 0;
}
