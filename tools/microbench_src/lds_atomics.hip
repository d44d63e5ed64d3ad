// LDS atomic throughput on one CU, sixteen waves: cycles per wave-instruction of ds_add_u32 /
// ds_or_rtn_b32 / ds_write_b32 / ds_read_b32 at random, path-like and conflict-free addresses in a 16 KB array.
//   hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int kIters = 2048;
template <int OP, int PAT>
__global__ __launch_bounds__(1024) void k(uint32_t *out, unsigned long long *cyc) {
    __shared__ uint32_t cells[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 1024) cells[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = threadIdx.x * 2654435761u + 12345u, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
        uint32_t a;
        if (PAT == 0) { x = x * 1664525u + 1013904223u; a = (x >> 12) & 4095u; }                 // random
        else if (PAT == 1) a = (wave * 251u + it * 640u + lane * 10u + ((it * 7u + lane) & 3u)) & 4095u;  // a path's records: ten segments apart
        else a = (it * 64u + lane) & 4095u;                                                            // conflict-free
        if (OP == 0) atomicAdd(&cells[a], 1u);
        else if (OP == 1) acc += atomicOr(&cells[a], 1u << (lane & 31));
        else if (OP == 2) cells[a] = it;
        else acc += cells[a];
        if (OP == 4) { atomicAdd(&cells[a], 1u); atomicAdd(&cells[a + 10], ~0u); acc += atomicOr(&cells[(a >> 5) + 64], 1u << (a & 31)); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = acc + cells[threadIdx.x];
}
template <int OP, int PAT>
void run(const char *name, int grid) {
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, grid * 1024 * 4); hipMalloc(&cyc, grid * 8);
    hipLaunchKernelGGL((k<OP, PAT>), dim3(grid), dim3(1024), 0, 0, out, cyc);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL((k<OP, PAT>), dim3(grid), dim3(1024), 0, 0, out, cyc); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (kIters * 16.0 * (OP == 4 ? 4 : 1));
    printf("%-34s %8.1f cycles (clock counter units) per wave-instruction per CU, kernel %.3f ms\n", name, per, ms);
    hipFree(out); hipFree(cyc);
}
int main() {
    const int grid = 256;
    run<0, 0>("ds_add_u32 random", grid); run<0, 1>("ds_add_u32 path-like", grid); run<0, 2>("ds_add_u32 conflict-free", grid);
    run<1, 0>("ds_or_rtn_b32 random", grid); run<1, 1>("ds_or_rtn_b32 path-like", grid); run<1, 2>("ds_or_rtn_b32 conflict-free", grid);
    run<2, 0>("ds_write_b32 random", grid); run<2, 2>("ds_write_b32 conflict-free", grid);
    run<3, 0>("ds_read_b32 random", grid); run<3, 2>("ds_read_b32 conflict-free", grid);
    run<4, 1>("2 adds + 1 or_rtn path-like (per instr)", grid);
    return 0;
}
