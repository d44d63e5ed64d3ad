// How fast can 400 MB of pageable (or file-mapped) host memory reach the device?
//   hipcc --offload-arch=gfx950 -O2 tools/h2d_test.hip -o tools/bin/h2d_test -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
int main(int argc, char **argv) {
    const size_t n = 400u << 20;
    char *src;
    if (argc > 1) {  // a file-mapped source, as flatgfa_load gives: h2d_test FILE [populate]
        { FILE *f = fopen(argv[1], "wb"); std::vector<char> z(1 << 20, 1); for (size_t i = 0; i < n; i += z.size()) fwrite(z.data(), 1, z.size(), f); fclose(f); }
        int fd = open(argv[1], O_RDONLY);
        double m0 = now();
        src = (char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE | (argc > 2 ? MAP_POPULATE : 0), fd, 0);
        printf("mmap%s %.1f ms\n", argc > 2 ? " (MAP_POPULATE)" : "", (now() - m0) * 1e3);
    } else {
        src = (char *)malloc(n);
        memset(src, 1, n);
    }
    char *dst;
    CK(hipMalloc(&dst, n));
    CK(hipMemcpy(dst, src, 4096, hipMemcpyHostToDevice));
    double t0 = now(), t1;
    if (argc <= 1) {
        CK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
        t1 = now();
        printf("plain hipMemcpy pageable        %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    }
    // chunked: T threads memcpy into pinned staging, async copies
    for (int T : {2, 4, 8}) {
        const size_t chunk = 8u << 20;
        const int nbuf = 2 * T;
        std::vector<char *> stage(nbuf);
        for (auto &p : stage) CK(hipHostMalloc(&p, chunk));
        hipStream_t st;
        CK(hipStreamCreate(&st));
        std::vector<hipEvent_t> ev(nbuf);
        for (auto &e : ev) CK(hipEventCreate(&e));
        t0 = now();
        const size_t nchunks = (n + chunk - 1) / chunk;
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                int round = 0;
                for (size_t c = t; c < nchunks; c += T, ++round) {
                    const int b = t * 2 + (round & 1);
                    if (round >= 2) CK(hipEventSynchronize(ev[b]));
                    const size_t off = c * chunk, len = std::min(chunk, n - off);
                    memcpy(stage[b], src + off, len);
                    CK(hipMemcpyAsync(dst + off, stage[b], len, hipMemcpyHostToDevice, st));
                    CK(hipEventRecord(ev[b], st));
                }
            });
        for (auto &x : th) x.join();
        CK(hipStreamSynchronize(st));
        t1 = now();
        printf("staged, %d threads               %.1f ms  %.1f GB/s\n", T, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        for (auto &p : stage) CK(hipHostFree(p));
    }
    return 0;
}
