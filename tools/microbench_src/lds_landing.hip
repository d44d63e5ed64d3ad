// Would k_scan's step stream gain from landing in LDS instead of in pinned registers?  (round 5's review, item 2: the
// landing sets are 48 of k_scan's 124 registers and hold the kernel at four waves per SIMD.)
//
// What this measures, cold (a 1.6 GB array: six times the Infinity Cache): persistent workgroups stream blocks of 1024
// steps (4 KiB, four 1-KiB wave-wide reads, nt) and do k_scan's kind of work on them -- nothing / the run-start test of
// pass A (three vector instructions a step) / that and as much again (what pass B and the emit add) -- with
//   reg   the block in sixteen registers per lane, requested a block ahead (what hipcc makes of `next = load; use(cur)`:
//         one block in flight while the other is worked on),
//   lds   the block landed by global_load_lds_dwordx4 in one of two 4-KiB LDS slots of the wave, requested a block ahead
//         behind a counted vmcnt, taken out with four ds_read_b128 when it is worked on,
// at 16 waves a CU (one workgroup of 1024 threads: k_scan's shape), 20 (two of 640), 24 (two of 768) and 32 (two of 1024;
// reg only where 64 registers suffice).  LDS per workgroup: waves x 8 KiB, so two workgroups of ten waves fill a CU's 160 KB
// exactly -- with no room for k_scan's run queues (140 KB today); the lds rows at 20 waves and more therefore use 2-KiB
// blocks (512 steps), which is what a two-workgroup k_scan could afford at best.
//   hipcc --offload-arch=gfx950 -O3 lds_landing.hip -o lds_landing && ./lds_landing
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

// the work on one lane's sixteen steps (four groups of four consecutive ones): WORK 0 = a sum (so that the loads are used),
// 1 = count where a run starts (id != the id before + 1: compare, conditional add -- as pass A does per step),
// 2 = the same and a second dependent pass of three operations a step
template <int WORK, int G>
__device__ __forceinline__ uint32_t work(const u32x4 (&v)[G], uint32_t carry) {
    uint32_t acc = carry;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const uint32_t a = v[k].x >> 1, b = v[k].y >> 1, c = v[k].z >> 1, d = v[k].w >> 1;
        if (WORK == 0) {
            acc += a ^ b ^ c ^ d;
        } else {
            acc += (b != a + 1u) + (c != b + 1u) + (d != c + 1u) + (a != (acc & 0xFFFFu));
            if (WORK == 2) {
                uint32_t t = a * 3u + (acc >> 3);
                t = (t ^ b) + (c & 0xFFFu);
                t = (t << 1) ^ d;
                acc += t & 1u;
            }
        }
    }
    return acc;
}

template <int WORK>
__global__ __launch_bounds__(1024) void k_reg(const u32x4 *__restrict__ steps, uint64_t n_blocks, uint32_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u, waves = blockDim.x >> 6;
    const uint64_t wave_id = (uint64_t)blockIdx.x * waves + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * waves;
    // a wave walks a contiguous stretch of blocks (as a k_scan item is contiguous)
    const uint64_t lo = n_blocks * wave_id / n_waves, hi = n_blocks * (wave_id + 1) / n_waves;
    uint32_t acc = 0;
    u32x4 cur[4], nxt[4];
    if (lo < hi) {
        const u32x4 *p = steps + lo * 256 + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = __builtin_nontemporal_load(p + 64 * k);
    }
    for (uint64_t b = lo; b < hi; ++b) {
        if (b + 1 < hi) {
            const u32x4 *p = steps + (b + 1) * 256 + lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) nxt[k] = __builtin_nontemporal_load(p + 64 * k);
        }
        acc = work<WORK, 4>(cur, acc);
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
    }
    if (acc == 0x12345u) out[wave_id] = acc;  // (never true for the test data: the work is kept, nothing is written)
}

// one 1-KiB wave-wide read into LDS: every lane's 16 bytes land at lds_base + 16 * lane
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(lds_u32 *)p; }

// BLK: kibibytes per block (4 = 1024 steps as k_scan's, 2 = 512)
template <int WORK, int BLK>
__global__ __launch_bounds__(1024) void k_lds(const u32x4 *__restrict__ steps, uint64_t n_blocks4k, uint32_t *__restrict__ out) {
    extern __shared__ u32x4 slots[];  // per wave: two slots of BLK KiB
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t n_blocks = n_blocks4k * (4 / BLK);
    const uint64_t wave_id = (uint64_t)blockIdx.x * waves + wave, n_waves = (uint64_t)gridDim.x * waves;
    const uint64_t lo = n_blocks * wave_id / n_waves, hi = n_blocks * (wave_id + 1) / n_waves;
    constexpr int kPer = BLK * 64;  // u32x4s per block
    u32x4 *mine = slots + (size_t)wave * 2 * kPer;
    const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr(mine));  // (wave-uniform: it goes into M0)
    uint32_t acc = 0;
    const auto request = [&](uint64_t b, uint32_t slot) {
        const u32x4 *p = steps + b * kPer + lane;
#pragma unroll
        for (int k = 0; k < BLK; ++k) glds16(p + 64 * k, base + slot * (kPer * 16) + k * 1024);
    };
    if (lo < hi) request(lo, 0);
    for (uint64_t b = lo; b < hi; ++b) {
        const uint32_t slot = __builtin_amdgcn_readfirstlane((uint32_t)(b - lo) & 1u);
        if (b + 1 < hi) {
            request(b + 1, slot ^ 1u);
            if (BLK == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // block b's reads have landed; b + 1's are in flight
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        u32x4 cur[BLK];
#pragma unroll
        for (int k = 0; k < BLK; ++k) cur[k] = mine[slot * kPer + 64 * k + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc = work<WORK, BLK>(cur, acc);
    }
    if (acc == 0x12345u) out[wave_id] = acc;
}

__global__ void k_fill(uint32_t *p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t id = (uint32_t)(i % 1000003u) + ((i & 15u) == 7u ? 5u : 0u);  // runs of a dozen steps
        p[i] = (id << 1) | (uint32_t)(i & 1u);
    }
}

template <typename K>
static float time_kernel(K launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        ts.push_back(ms);
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const uint64_t n = 400ull * 1000 * 1000;  // 1.6 GB of steps: nothing of a pass is left in the 256 MiB Infinity Cache for the next
    const uint64_t n_blocks = n / 1024;
    uint32_t *steps = nullptr, *out = nullptr;
    CHECK(hipMalloc(&steps, n * 4));
    CHECK(hipMalloc(&out, 1 << 20));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, steps, n);
    CHECK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const u32x4 *s4 = reinterpret_cast<const u32x4 *>(steps);
    printf("%d CUs; %.1f GB of steps per pass; median of 5 launches\n", cus, n * 4 / 1e9);
    printf("%-44s %10s %10s %10s\n", "landing, waves per CU", "loads only", "+ pass A", "+ as much again");
    const auto row = [&](const char *name, float t0, float t1, float t2) {
        printf("%-44s %7.0f GB/s %7.0f GB/s %7.0f GB/s   (%.3f / %.3f / %.3f ms)\n", name, n * 4 / t0 / 1e6, n * 4 / t1 / 1e6, n * 4 / t2 / 1e6, t0, t1, t2);
    };
#define REG_ROW(NAME, THREADS, PER_CU)                                                                                         \
    row(NAME, time_kernel([&] { hipLaunchKernelGGL((k_reg<0>), dim3(cus * PER_CU), dim3(THREADS), 0, 0, s4, n_blocks, out); }, 5), \
        time_kernel([&] { hipLaunchKernelGGL((k_reg<1>), dim3(cus * PER_CU), dim3(THREADS), 0, 0, s4, n_blocks, out); }, 5),       \
        time_kernel([&] { hipLaunchKernelGGL((k_reg<2>), dim3(cus * PER_CU), dim3(THREADS), 0, 0, s4, n_blocks, out); }, 5))
#define LDS_ROW(NAME, THREADS, PER_CU, BLK)                                                                                    \
    do {                                                                                                                       \
        const size_t lds = (size_t)(THREADS / 64) * 2 * BLK * 1024;                                                            \
        hipFuncSetAttribute((const void *)k_lds<0, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        hipFuncSetAttribute((const void *)k_lds<1, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        hipFuncSetAttribute((const void *)k_lds<2, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        row(NAME, time_kernel([&] { hipLaunchKernelGGL((k_lds<0, BLK>), dim3(cus * PER_CU), dim3(THREADS), lds, 0, s4, n_blocks, out); }, 5), \
            time_kernel([&] { hipLaunchKernelGGL((k_lds<1, BLK>), dim3(cus * PER_CU), dim3(THREADS), lds, 0, s4, n_blocks, out); }, 5),       \
            time_kernel([&] { hipLaunchKernelGGL((k_lds<2, BLK>), dim3(cus * PER_CU), dim3(THREADS), lds, 0, s4, n_blocks, out); }, 5));      \
    } while (0)
    REG_ROW("reg, 8 waves (1 x 512)", 512, 1);
    REG_ROW("reg, 12 waves (1 x 768)", 768, 1);
    REG_ROW("reg, 16 waves (1 x 1024: k_scan's shape)", 1024, 1);
    REG_ROW("reg, 20 waves (2 x 640)", 640, 2);
    REG_ROW("reg, 24 waves (2 x 768)", 768, 2);
    REG_ROW("reg, 32 waves (2 x 1024)", 1024, 2);
    LDS_ROW("lds 4 KiB blocks, 16 waves (1 x 1024)", 1024, 1, 4);
    LDS_ROW("lds 2 KiB blocks, 16 waves (1 x 1024)", 1024, 1, 2);
    LDS_ROW("lds 2 KiB blocks, 20 waves (2 x 640)", 640, 2, 2);
    LDS_ROW("lds 2 KiB blocks, 24 waves (2 x 768)", 768, 2, 2);
    LDS_ROW("lds 2 KiB blocks, 32 waves (2 x 1024)", 1024, 2, 2);
    // a device-to-device copy of the same bytes, for scale (read + write)
    uint32_t *dst = nullptr;
    if (hipMalloc(&dst, n * 4) == hipSuccess) {
        const float t = time_kernel([&] { hipMemcpyAsync(dst, steps, n * 4, hipMemcpyDeviceToDevice, 0); }, 5);
        printf("hipMemcpy device to device: %.0f GB/s read + %.0f GB/s written (%.3f ms)\n", n * 4 / t / 1e6, n * 4 / t / 1e6, t);
        hipFree(dst);
    }
    hipFree(steps);
    hipFree(out);
    return 0;
}
