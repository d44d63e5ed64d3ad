// Microbenchmark: how fast can a persistent 1024-thread workgroup per CU stream 400 MB with
//   (a) coalesced 16-byte loads (lane l reads unit k*64 + l of its wave's 4 KB block), vs
//   (b) "lane owns 64 bytes": lane l reads units 4l..4l+3 with four back-to-back 16-byte loads,
//   (c) "lane owns 32 bytes": lane l reads units 2l, 2l+1 (two blocks of 2 KB per 4 KB),
//   (d) "quad reads 64 bytes": instruction k, lane l reads unit 16(l/4) + 4k + l%4, so that a
//       4x4 transpose inside each lane quad turns the block into layout (b).
// Build: hipcc --offload-arch=gfx950 -O3 tools/loadpat.hip -o /tmp/loadpat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ __launch_bounds__(1024) void k_pat(const u32x4 *__restrict__ src, uint64_t n_blocks4k, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t gw = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 16;
    // each wave owns a contiguous range of 4 KB blocks
    const uint64_t per = (n_blocks4k + nw - 1) / nw;
    const uint64_t b0 = gw * per, b1 = b0 + per < n_blocks4k ? b0 + per : n_blocks4k;
    uint32_t acc = 0;
    for (uint64_t b = b0; b < b1; ++b) {
        const u32x4 *p = src + b * 256;
        u32x4 v0, v1, v2, v3;
        if (MODE == 0) {
            p += lane;
            if (NT) { v0 = __builtin_nontemporal_load(p); v1 = __builtin_nontemporal_load(p + 64); v2 = __builtin_nontemporal_load(p + 128); v3 = __builtin_nontemporal_load(p + 192); }
            else { v0 = p[0]; v1 = p[64]; v2 = p[128]; v3 = p[192]; }
        } else if (MODE == 1) {
            p += lane * 4;
            if (NT) { v0 = __builtin_nontemporal_load(p); v1 = __builtin_nontemporal_load(p + 1); v2 = __builtin_nontemporal_load(p + 2); v3 = __builtin_nontemporal_load(p + 3); }
            else { v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3]; }
        } else if (MODE == 3) {
            p += (lane >> 2) * 16 + (lane & 3);
            if (NT) { v0 = __builtin_nontemporal_load(p); v1 = __builtin_nontemporal_load(p + 4); v2 = __builtin_nontemporal_load(p + 8); v3 = __builtin_nontemporal_load(p + 12); }
            else { v0 = p[0]; v1 = p[4]; v2 = p[8]; v3 = p[12]; }
        } else {
            p += lane * 2;
            if (NT) { v0 = __builtin_nontemporal_load(p); v1 = __builtin_nontemporal_load(p + 1); v2 = __builtin_nontemporal_load(p + 128); v3 = __builtin_nontemporal_load(p + 129); }
            else { v0 = p[0]; v1 = p[1]; v2 = p[128]; v3 = p[129]; }
        }
        acc += v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, bool NT>
static void run(const char *name, const u32x4 *src, uint64_t nb, uint32_t *out, int cus) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) k_pat<MODE, NT><<<cus, 1024>>>(src, nb, out);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) k_pat<MODE, NT><<<cus, 1024>>>(src, nb, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-40s %8.3f ms  %8.1f GB/s\n", name, ms, nb * 4096.0 / ms / 1e6);
}

int main() {
    const uint64_t bytes = 400ull << 20, nb = bytes / 4096;
    u32x4 *src;
    uint32_t *out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(src, 1, bytes));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    run<0, false>("coalesced", src, nb, out, cus);
    run<0, true>("coalesced nt", src, nb, out, cus);
    run<1, false>("lane owns 64 B", src, nb, out, cus);
    run<1, true>("lane owns 64 B nt", src, nb, out, cus);
    run<2, false>("lane owns 32 B", src, nb, out, cus);
    run<2, true>("lane owns 32 B nt", src, nb, out, cus);
    run<3, false>("quad reads 64 B", src, nb, out, cus);
    run<3, true>("quad reads 64 B nt", src, nb, out, cus);
    return 0;
}
