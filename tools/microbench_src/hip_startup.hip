// What a HIP process costs before and after its work on this box: wall-clock stamps around hipInit-equivalent,
// the first allocation, the first kernel launch, and (measured by the caller) process exit.
//   hip_startup [quick]   -- `quick`: leave with _exit(0) instead of returning from main
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <unistd.h>
__global__ void k(int *p) { if (p) p[0] = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const double t0 = now();
    int n = 0;
    hipGetDeviceCount(&n);
    const double t1 = now();
    hipSetDevice(0);
    int *p = nullptr;
    hipMalloc(&p, 4);
    const double t2 = now();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p);
    hipDeviceSynchronize();
    const double t3 = now();
    void *h = nullptr;
    hipHostMalloc(&h, 64 << 20);
    const double t4 = now();
    fprintf(stderr, "hipGetDeviceCount %.1f ms, first hipMalloc %.1f, first launch+sync %.1f, hipHostMalloc(64 MB) %.1f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    if (argc > 1 && !strcmp(argv[1], "quick")) _exit(0);
    return 0;
}
