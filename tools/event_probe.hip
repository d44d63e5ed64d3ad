// What do HIP events say a kernel took?  A kernel that spins for a known time (the 100 MHz
// wall clock), timed two ways: events recorded around the launch (what ProfScope does) and
// hipExtLaunchKernelGGL's start + stop events.  MI355X, ROCm 7.0: 3.7 us and 2.0 us on top of
// the spin; on an empty launch of k_scan's shape (256 workgroups of 1024 threads with 142 KB of
// LDS) the second way read 8.4 us against 6 -- the library records events around its launches.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/event_probe tools/event_probe.hip && /tmp/event_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long r0 = wall_clock64();
    while (wall_clock64() - r0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = wall_clock64() - r0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    unsigned long long *d;
    CK(hipMalloc(&d, 8));
    hipEvent_t a, b, c, e;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c)); CK(hipEventCreate(&e));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (unsigned long long us : {10ull, 100ull}) {
        const unsigned long long ticks = us * 100;
        for (int rep = 0; rep < 4; ++rep) {
            float m1 = 0, m2 = 0;
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, ticks, d);  // (something before, as in a call)
            CK(hipEventRecord(a, s));
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, ticks, d);
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&m1, a, b));
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, ticks, d);
            hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, c, e, 0, ticks, d);
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&m2, c, e));
            printf("spin %llu us: events around the launch %.2f us | hipExtLaunchKernelGGL's own %.2f us\n", us, m1 * 1e3, m2 * 1e3);
        }
    }
    return 0;
}
