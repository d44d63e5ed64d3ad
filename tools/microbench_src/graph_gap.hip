// What a kernel boundary costs between two dependent kernels of k_scan's and k_accum's shape (256 x 1024 threads, ~90 and
// ~45 us): launched on a stream, as a two-node hipGraph launched once per step, and as one graph of all steps.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_src/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void spin(unsigned long long ticks, unsigned *out) {  // s_memrealtime: 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    unsigned *out;
    CK(hipMalloc(&out, 256));
    CK(hipMemset(out, 0, 256));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int steps = 200;
    const auto run = [&](const char *what, auto fn) {
        fn();
        hipStreamSynchronize(s);
        const auto t0 = std::chrono::steady_clock::now();
        fn();
        hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
        printf("%-44s %.2f us per step (kernels spin 90 + 45 = 135)\n", what, us);
    };
    run("stream launches", [&] {
        for (int i = 0; i < steps; ++i) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, s, 9000ull, out);
            hipLaunchKernelGGL(spin, dim3(245), dim3(1024), 0, s, 4500ull, out);
        }
    });
    hipGraph_t g2, gall;
    hipGraphExec_t e2, eall;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, s, 9000ull, out);
    hipLaunchKernelGGL(spin, dim3(245), dim3(1024), 0, s, 4500ull, out);
    CK(hipStreamEndCapture(s, &g2));
    CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    run("a graph of the two kernels per step", [&] {
        for (int i = 0; i < steps; ++i) hipGraphLaunch(e2, s);
    });
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < steps; ++i) {
        hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, s, 9000ull, out);
        hipLaunchKernelGGL(spin, dim3(245), dim3(1024), 0, s, 4500ull, out);
    }
    CK(hipStreamEndCapture(s, &gall));
    CK(hipGraphInstantiate(&eall, gall, nullptr, nullptr, 0));
    run("one graph of all steps", [&] { hipGraphLaunch(eall, s); });
    run("one kernel per step (spin 135)", [&] {
        for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, s, 13500ull, out);
    });
    return 0;
}
