// extern "C" surface of libflatgfa.so: the flatgfa-c drop-in (Part 1 of include/flatgfa.h)
// plus the additive loaders and depth queries (Part 2).  Device work is delegated to the
// flatgfa_dev_* entry points in depth_device.hip through the HIP runtime API only.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include <malloc.h>

#include <climits>

#include "../../include/flatgfa.h"
#include "device_common.hpp"
#include "temp_arena.hpp"
#include "flatgfa_core.hpp"

using fgfa_dev::set_error;

// A handle's stream comes from a per-device pool of the process and goes back to it idle when the handle is freed: creating a stream
// costs half a millisecond on this runtime, which is a third of what a resident graph's first answer costs.
namespace {
std::mutex g_stream_pool_mu;
std::vector<hipStream_t> g_stream_pool[64];
hipError_t stream_acquire(int device, hipStream_t *out) {
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        if (device >= 0 && device < 64 && !g_stream_pool[device].empty()) {
            *out = g_stream_pool[device].back();
            g_stream_pool[device].pop_back();
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void stream_release(int device, hipStream_t st) {
    if (!st) return;
    if (hipStreamSynchronize(st) != hipSuccess || device < 0 || device >= 64) {
        (void)hipGetLastError();
        (void)hipStreamDestroy(st);
        return;
    }
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    if (g_stream_pool[device].size() < 8) g_stream_pool[device].push_back(st);
    else (void)hipStreamDestroy(st);
}
}  // namespace

// The opaque store (flatgfa-c/src/lib.rs:16-28): either a heap store built by the parser /
// generator, or a borrowed view of a memory-mapped .flatgfa file.
struct CStore {
    fgfa::Store heap;
    std::unique_ptr<fgfa::MappedFile> file;
    fgfa::View view;

    // Resident device image (structure of arrays), created on first use.
    std::mutex op_mu;   // one depth query at a time per handle
    std::mutex dev_mu;
    bool on_device = false;
    int device = 0;
    uint32_t *d_steps = nullptr;
    uint32_t *d_small = nullptr;  // one allocation behind the five arrays below
    uint32_t *d_path_begin = nullptr, *d_path_end = nullptr, *d_seg_len = nullptr;
    uint32_t *d_depth = nullptr, *d_uniq = nullptr;
    uint64_t *d_sums = nullptr;  // [2 * paths] scratch of flatgfa_path_depth (inside d_small)
    double h2d_ms = 0, plan_ms = 0;  // what becoming resident took: the copies, and the plan's creation (flatgfa_residency_ms)
    // The plan's creation is the first query (flatgfa_dev_plan_create_first): d_depth / d_uniq hold seg_depth_with_uniq of the
    // graph until another query overwrites them, and the first node-depth call of the handle takes them as they are.
    bool first_answer = false;
    int first_rc = 0;
    std::vector<uint32_t> h_path_begin, h_path_end;
    std::vector<uint32_t> h_soa;  // the host image of the small arrays (spans, segment lengths) that was uploaded: kept, not unmapped behind the upload
    flatgfa_dev_plan_t *plan = nullptr;
    hipStream_t stream = nullptr;
    // the prepared query of the last flatgfa_seg_depth_subset call, reused while the subset is the same
    std::vector<uint32_t> sub_ids;
    flatgfa_dev_plan_t *sub_plan = nullptr;
    uint32_t *d_sub_spans = nullptr;  // the subset's begin[] then end[]
    std::atomic<int> steps_ok{-1};  // -1 = not checked yet: do all step handles name a segment? (host-side walks index by them)

    ~CStore() {
        if (plan) flatgfa_dev_plan_destroy(plan);
        if (sub_plan) flatgfa_dev_plan_destroy(sub_plan);
        if (d_sub_spans) (void)hipFree(d_sub_spans);
        for (uint32_t *p : {d_steps, d_small})
            if (p) (void)hipFree(p);
        stream_release(device, stream);
    }
};

// Host code that follows step handles into the segment pool (path lengths, the GFA printer, the
// interval walk) asks this first; the kernels check the ids themselves.
static bool steps_name_segments(CStore *cs) {
    if (cs->steps_ok < 0) cs->steps_ok = fgfa::validate_step_ids(cs->view) ? 1 : 0;
    if (!cs->steps_ok) set_error("a step refers to a segment id that is out of range");
    return cs->steps_ok == 1;
}

// (sharded.hip reads the pools of a handle)
const fgfa::View &flatgfa_capi_view(flatgfa_t gfa) { return gfa->view; }

#define CAPI_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            return FLATGFA_ERR_HIP;                                                         \
        }                                                                                   \
    } while (0)

extern "C" {

const char *flatgfa_last_error(void) { return fgfa_dev::last_error(); }

// ------------------------------------------------------------- Part 1 ---

static flatgfa_t parse_common(const uint8_t *data, size_t len, bool stream_mode) {
    auto cs = std::make_unique<CStore>();
    std::string err;
    if (!fgfa::parse_gfa(data ? data : (const uint8_t *)"", len, &cs->heap, &err, stream_mode)) {
        set_error(err);
        return nullptr;
    }
    cs->view = cs->heap.view();
    return cs.release();
}

flatgfa_t flatgfa_parse_bytes(const uint8_t *data, size_t len) { return parse_common(data, len, false); }
flatgfa_t flatgfa_parse_stream_bytes(const uint8_t *data, size_t len) { return parse_common(data, len, true); }

flatgfa_t flatgfa_parse(const char *filename) {
    if (!filename) { set_error("flatgfa_parse: NULL filename"); return nullptr; }
    fgfa::MappedFile f;
    std::string err;
    if (!f.open(filename, &err)) { set_error(err); return nullptr; }
    return flatgfa_parse_bytes(f.data, f.size);
}

flatgfa_t flatgfa_load(const char *filename) {
    if (!filename) { set_error("flatgfa_load: NULL filename"); return nullptr; }
    auto cs = std::make_unique<CStore>();
    cs->file = std::make_unique<fgfa::MappedFile>();
    std::string err;
    if (!cs->file->open(filename, &err) || !fgfa::view_flatgfa(cs->file->data, cs->file->size, &cs->view, &err)) {
        set_error(err);
        return nullptr;
    }
    return cs.release();
}

flatgfa_t flatgfa_synth(uint64_t seed, uint32_t n_segs, uint32_t n_paths, uint32_t steps_per_path, int model,
                        bool with_seq) {
    if (n_segs == 0 || (model < 0 || model > 4) || (uint64_t)n_paths * steps_per_path > 0xFFFFFFFFull ||
        n_segs > 0x7FFFFFFFu) {
        set_error("flatgfa_synth: bad shape");
        return nullptr;
    }
    auto cs = std::make_unique<CStore>();
    fgfa::synth_store(seed, n_segs, n_paths, steps_per_path, model, with_seq, &cs->heap);
    cs->view = cs->heap.view();
    return cs.release();
}

void flatgfa_free(flatgfa_t gfa) { delete gfa; }

uint32_t flatgfa_get_segment_count(flatgfa_t gfa) { return (uint32_t)gfa->view.segs.len; }

flatgfa_string_t flatgfa_get_seq(flatgfa_t gfa, uint32_t segment_id) {
    flatgfa_string_t s{nullptr, 0};
    const fgfa::View &v = gfa->view;
    if (segment_id >= v.segs.len) return s;
    fgfa::Span sp = v.segs[segment_id].seq;
    s.data = v.seq_data.data + sp.start;
    s.len = (int)sp.len();
    return s;
}

uint32_t flatgfa_path_count(flatgfa_t gfa) { return (uint32_t)gfa->view.paths.len; }

flatgfa_string_t flatgfa_get_path_name(flatgfa_t gfa, uint32_t path_index) {
    flatgfa_string_t s{nullptr, 0};
    const fgfa::View &v = gfa->view;
    if (path_index >= v.paths.len) return s;
    fgfa::Span sp = v.paths[path_index].name;
    s.data = v.name_data.data + sp.start;
    s.len = (int)sp.len();
    return s;
}

uint32_t flatgfa_get_path_step_count(flatgfa_t gfa, uint32_t path_index) {
    const fgfa::View &v = gfa->view;
    if (path_index >= v.paths.len) return UINT32_MAX;
    return v.paths[path_index].steps.len();
}

bool flatgfa_get_step(flatgfa_t gfa, uintptr_t path_index, uintptr_t step_index, flatgfa_handle_t *out) {
    const fgfa::View &v = gfa->view;
    if (path_index >= v.paths.len) return false;
    fgfa::Span sp = v.paths[path_index].steps;
    if (step_index >= sp.len()) return false;
    fgfa::Handle h = v.steps[sp.start + step_index];
    out->segment_id = h.segment();
    out->is_forward = h.is_forward();
    return true;
}

// ------------------------------------------------------------- Part 2 ---

int flatgfa_pool(flatgfa_t gfa, int ix, const void **data, uint64_t *len, uint64_t *elem_size) {
    if (!gfa || ix < 0 || ix > 10) { set_error("flatgfa_pool: bad argument"); return FLATGFA_ERR_ARG; }
    if (data) *data = gfa->view.pool_data(ix);
    if (len) *len = gfa->view.pool_len(ix);
    if (elem_size) *elem_size = fgfa::kPoolElemSize[ix];
    return FLATGFA_OK;
}

int64_t flatgfa_find_path(flatgfa_t gfa, const uint8_t *name, size_t len) {
    if (!gfa) return -1;
    return gfa->view.find_path(name, len);
}

int flatgfa_write_flatgfa(flatgfa_t gfa, const char *filename) {
    if (!gfa || !filename) { set_error("flatgfa_write_flatgfa: NULL argument"); return FLATGFA_ERR_ARG; }
    size_t n = fgfa::flatgfa_file_size(gfa->view);
    std::vector<uint8_t> buf(n);
    fgfa::dump_flatgfa(gfa->view, buf.data());
    FILE *f = fopen(filename, "wb");
    if (!f) { set_error(std::string("cannot create ") + filename); return FLATGFA_ERR_IO; }
    size_t w = fwrite(buf.data(), 1, n, f);
    if (fclose(f) != 0 || w != n) { set_error(std::string("short write to ") + filename); return FLATGFA_ERR_IO; }
    return FLATGFA_OK;
}

int flatgfa_write_flatgfa_prealloc(flatgfa_t gfa, const char *filename, const uint8_t *gfa_text, size_t text_len, uint32_t factor) {
    if (!gfa || !filename || (text_len && !gfa_text)) { set_error("flatgfa_write_flatgfa_prealloc: NULL argument"); return FLATGFA_ERR_ARG; }
    uint64_t cap[11];
    std::string err;
    if (gfa_text) {
        if (!fgfa::estimate_toc(gfa_text, text_len, cap, &err)) { set_error(err); return FLATGFA_ERR_BOUNDS; }
    } else {
        fgfa::guess_toc(factor, cap);
    }
    size_t n = 0;
    if (!fgfa::prealloc_file_size(gfa->view, cap, &n, &err)) { set_error(err); return FLATGFA_ERR_BOUNDS; }
    // The file is the sum of the CAPACITIES (up to 2^46 bytes; `-p 1000` is already gigabytes), most of it
    // never written: made sparse with ftruncate, then the table of contents and each pool's `len`
    // entries written at their offsets -- what the reference's map_new_file + file::init amount to
    // (memfile.rs:12-33, file.rs:255-272).  A fresh file reads as zeros behind what is written.
    const int fd = open(filename, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { set_error(std::string("cannot create ") + filename); return FLATGFA_ERR_IO; }
    bool ok = ftruncate(fd, (off_t)n) == 0;
    const auto put = [&](const void *p, size_t bytes, uint64_t at) {
        const char *c = (const char *)p;
        while (ok && bytes) {
            const ssize_t w = pwrite(fd, c, bytes, (off_t)at);
            if (w <= 0) { ok = false; break; }
            c += w;
            at += (uint64_t)w;
            bytes -= (size_t)w;
        }
    };
    fgfa::Toc toc;
    toc.magic = fgfa::kMagic;
    for (int i = 0; i < 11; ++i) toc.pool[i] = fgfa::TocSize{gfa->view.pool_len(i), cap[i]};
    put(&toc, sizeof toc, 0);
    uint64_t off = sizeof toc;
    for (int i = 0; i < 11; ++i) {
        put(gfa->view.pool_data(i), gfa->view.pool_len(i) * fgfa::kPoolElemSize[i], off);
        off += cap[i] * fgfa::kPoolElemSize[i];
    }
    if (close(fd) != 0 || !ok) { set_error(std::string("short write to ") + filename); return FLATGFA_ERR_IO; }
    return FLATGFA_OK;
}

static int give_text(const std::string &s, char **text, size_t *len) {
    char *p = (char *)malloc(s.size() + 1);
    if (!p) { set_error("out of memory"); return FLATGFA_ERR_IO; }
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    *text = p;
    if (len) *len = s.size();
    return FLATGFA_OK;
}

int flatgfa_translate_prealloc(const uint8_t *gfa_text, size_t text_len, int from_stream, const char *filename, uint32_t factor) {
    if (!filename || (text_len && !gfa_text)) { set_error("flatgfa_translate_prealloc: NULL argument"); return FLATGFA_ERR_ARG; }
    const uint8_t *text = gfa_text ? gfa_text : (const uint8_t *)"";
    uint64_t cap[11];
    std::string err;
    if (from_stream) {
        fgfa::guess_toc(factor, cap);
    } else if (!fgfa::estimate_toc(text, text_len, cap, &err)) {
        set_error(err);
        return FLATGFA_ERR_BOUNDS;
    }
    size_t n = 0;
    if (!fgfa::toc_file_size(cap, &n, &err)) { set_error(err); return FLATGFA_ERR_BOUNDS; }
    // memfile::map_new_file (memfile.rs:24-33): created, sized (sparse: it reads as zeros), mapped shared
    const int fd = open(filename, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { set_error(std::string("cannot create ") + filename); return FLATGFA_ERR_IO; }
    if (ftruncate(fd, (off_t)n) != 0) {
        close(fd);
        set_error(std::string("cannot size ") + filename);
        return FLATGFA_ERR_IO;
    }
    void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { set_error(std::string("cannot map ") + filename); return FLATGFA_ERR_IO; }
    const bool ok = fgfa::parse_gfa_prealloc(text, text_len, from_stream != 0, cap, (uint8_t *)m, &err);
    const bool flushed = msync(m, n, MS_SYNC) == 0;  // (mmap.flush())
    munmap(m, n);
    if (!ok) {
        set_error(err);
        return err.rfind("preallocated flatgfa:", 0) == 0 ? FLATGFA_ERR_BOUNDS : FLATGFA_ERR_PARSE;
    }
    if (!flushed) { set_error(std::string("cannot flush ") + filename); return FLATGFA_ERR_IO; }
    return FLATGFA_OK;
}

int flatgfa_print_gfa(flatgfa_t gfa, char **text, size_t *len) {
    if (!gfa || !text) { set_error("flatgfa_print_gfa: NULL argument"); return FLATGFA_ERR_ARG; }
    if (!steps_name_segments(gfa)) return FLATGFA_ERR_BOUNDS;
    std::string out, err;
    if (!fgfa::print_gfa(gfa->view, &out, &err)) { set_error(err); return FLATGFA_ERR_BOUNDS; }
    return give_text(out, text, len);
}

void flatgfa_free_text(char *text) { free(text); }

int flatgfa_format_float(double x, int digits, char *out, int cap) {
    std::string s = fgfa::format_float(x, digits);
    int n = (int)std::min<size_t>(s.size(), cap > 0 ? (size_t)cap : 0);
    memcpy(out, s.data(), (size_t)n);
    return n;
}

// ---- device residency ----

// The pinned staging buffers are the process's own, made on first use and kept: allocating and
// freeing eight of them per upload cost 2.9 + 3.8 ms of the 17 ms a cfg-L graph took to become
// resident (FLATGFA_TIMING).  One upload at a time uses them.
namespace {
constexpr size_t kChunk = 8u << 20;
constexpr int kMaxUploadThreads = 16;
struct StagePool {
    std::mutex mu;  // held for the whole of an upload
    char *stage[2 * kMaxUploadThreads] = {};
    hipEvent_t ev[2 * kMaxUploadThreads] = {};
    int device = -1;
    void release() {
        for (int i = 0; i < 2 * kMaxUploadThreads; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (stage[i]) (void)hipHostFree(stage[i]);
            ev[i] = nullptr;
            stage[i] = nullptr;
        }
        device = -1;
    }
    hipError_t ensure(int dev, int n) {
        if (device != dev) release();  // (events belong to a device)
        device = dev;
        for (int i = 0; i < n; ++i) {
            if (!stage[i]) {
                const hipError_t rc = hipHostMalloc((void **)&stage[i], kChunk, hipHostMallocDefault);
                if (rc != hipSuccess) { stage[i] = nullptr; return rc; }
            }
            if (!ev[i]) {
                const hipError_t rc = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
                if (rc != hipSuccess) { ev[i] = nullptr; return rc; }
            }
        }
        return hipSuccess;
    }
};
StagePool *stage_pool() {
    static StagePool *p = new StagePool();  // never destroyed: the HIP runtime may be gone by the time static destructors run
    return p;
}
}  // namespace

// Host -> device copy of a large pageable (or file-mapped) region.  A plain hipMemcpy stages it
// through the runtime's own pinned buffer on one thread (10-24 GB/s here, less when the source
// is a mapped file that still has to be faulted in); a few threads copying 8 MB chunks into
// pinned buffers and queueing async copies reach the PCIe rate (~49 GB/s measured with four,
// tools/h2d_test.hip; a freshly mapped file adds ~17 ms of first-touch page faults per 400 MB,
// which pread() into the pinned buffers does not beat).  Small regions take the plain route.
static hipError_t upload(void *dst, const void *src, size_t bytes, hipStream_t stream) {
    static const int kThreads = [] {
        const char *e = getenv("FLATGFA_UPLOAD_THREADS");
        const int n = e ? atoi(e) : 4;
        return n < 1 ? 1 : n > kMaxUploadThreads ? kMaxUploadThreads : n;
    }();
    const bool timing = getenv("FLATGFA_TIMING") != nullptr;
    auto tick = [t = std::chrono::steady_clock::now(), timing](const char *what) mutable {
        if (!timing) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "upload: %-31s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    };
    if (bytes < 4 * kChunk) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    int device = 0;
    hipError_t rc = hipGetDevice(&device);
    if (rc != hipSuccess) return rc;
    StagePool &pool = *stage_pool();
    std::lock_guard<std::mutex> lk(pool.mu);
    rc = pool.ensure(device, 2 * kThreads);
    char **stage = pool.stage;
    hipEvent_t *ev = pool.ev;
    std::atomic<int> failed{(int)rc};
    tick("pinned buffers + events");
    if (rc == hipSuccess) {
        const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
        std::vector<std::thread> workers;
        for (int t = 0; t < kThreads; ++t)
            workers.emplace_back([&, t]() {
                const auto w0 = std::chrono::steady_clock::now();
                const auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count(); };
                if (hipSetDevice(device) != hipSuccess) { failed = (int)hipErrorInvalidDevice; return; }
                const double t_dev = since();
                const auto populate = [&](size_t off, size_t len) {
#ifdef MADV_POPULATE_READ
                    // a freshly mapped file: let the kernel map the chunk's pages in one go instead of taking a fault per
                    // page inside the memcpy (on anonymous memory it only maps what the copy would touch anyway)
                    static const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
                    const uintptr_t a0 = ((uintptr_t)src + off) & ~(page - 1);
                    (void)madvise((void *)a0, ((uintptr_t)src + off + len) - a0, MADV_POPULATE_READ);
#else
                    (void)off, (void)len;
#endif
                };
                int round = 0;
                for (size_t c = (size_t)t; c < n_chunks && !failed; c += kThreads, ++round) {
                    const int b = 2 * t + (round & 1);
                    const double t_w = since();
                    hipError_t e = round >= 2 ? hipEventSynchronize(ev[b]) : hipSuccess;  // the buffer's previous copy is done
                    const size_t off = c * kChunk, len = std::min(kChunk, bytes - off);
                    const double t_a = since();
                    if (e == hipSuccess) {
                        populate(off, len);
                        memcpy(stage[b], (const char *)src + off, len);
                        const double t_b = since();
                        e = hipMemcpyAsync((char *)dst + off, stage[b], len, hipMemcpyHostToDevice, stream);
                        if (timing && t == 0 && round < 3)
                            fprintf(stderr, "upload: worker 0 chunk %2d at %6.2f ms (set device %.2f): waited %.2f, fault + stage %.2f, queue %.2f ms\n", round, t_w, t_dev, t_a - t_w, t_b - t_a, since() - t_b);
                    }
                    if (e == hipSuccess) e = hipEventRecord(ev[b], stream);
                    if (e != hipSuccess) failed = (int)e;
                }
            });
        for (auto &w : workers) w.join();
        tick("workers: fault in, stage, queue");
        const hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess && !failed) failed = (int)e;
        tick("copies drained");
    }
    return (hipError_t)failed.load();
}

int flatgfa_keep_host_memory(int on) {
#ifdef __GLIBC__
    // (never map: large blocks come from the heap and go back to its free lists; never trim: the heap's top stays)
    (void)mallopt(M_MMAP_MAX, on ? 0 : 65536);
    (void)mallopt(M_TRIM_THRESHOLD, on ? INT_MAX : 128 * 1024);
#else
    (void)on;
#endif
    return FLATGFA_OK;
}

int flatgfa_warm_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device is visible; the depth queries have no CPU fallback");
        return FLATGFA_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("device index out of range"); return FLATGFA_ERR_ARG; }
    CAPI_HIP(hipSetDevice(device));
    {   // the staging buffers an upload will want (kept by the process)
        static const int kThreads = [] {
            const char *e = getenv("FLATGFA_UPLOAD_THREADS");
            const int n = e ? atoi(e) : 4;
            return n < 1 ? 1 : n > kMaxUploadThreads ? kMaxUploadThreads : n;
        }();
        StagePool &pool = *stage_pool();
        std::lock_guard<std::mutex> lk(pool.mu);
        CAPI_HIP(pool.ensure(device, 2 * kThreads));
        // the process's first asynchronous copy and first launch (queues, the library's code object)
        uint32_t *d = nullptr;
        hipStream_t s = nullptr;
        CAPI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipError_t e = hipMalloc(&d, 4096);
        if (e == hipSuccess) e = hipMemcpyAsync(d, pool.stage[0], 4096, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            fgfa_dev::warm_launch(s);  // (loads the library's code object)
            e = hipStreamSynchronize(s);
        }
        if (d) (void)hipFree(d);
        (void)hipStreamDestroy(s);
        CAPI_HIP(e);
    }
    return FLATGFA_OK;
}

static int ensure_device(CStore *cs, int device) {
    std::lock_guard<std::mutex> lk(cs->dev_mu);
    if (cs->on_device) {
        if (device >= 0 && device != cs->device) { set_error("graph is already resident on another device"); return FLATGFA_ERR_ARG; }
        CAPI_HIP(hipSetDevice(cs->device));
        return FLATGFA_OK;
    }
    if (device < 0) device = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device is visible; the depth queries have no CPU fallback");
        return FLATGFA_ERR_NO_DEVICE;
    }
    if (device >= ndev) { set_error("device index out of range"); return FLATGFA_ERR_ARG; }
    const fgfa::View &v = cs->view;
    if (v.steps.len > 0xFFFFFFFFull || v.segs.len > 0x80000000ull || v.paths.len > 0xFFFFFFFFull) {
        set_error("graph too large for 32-bit ids");
        return FLATGFA_ERR_TOO_LARGE;
    }
    const bool timing = getenv("FLATGFA_TIMING") != nullptr;  // diagnostic: where making a graph resident spends its time
    const auto t_enter = std::chrono::steady_clock::now();
    auto tick = [t = t_enter, timing](const char *what) mutable {
        if (!timing) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "to_device: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    };
    // reversed or overlong step spans: where the reference would panic on the slice index (pool.rs:341-347)
    const size_t N = v.steps.len, P = v.paths.len, S = v.segs.len;
    for (size_t i = 0; i < P; ++i) {
        const fgfa::Span sp = v.paths[i].steps;
        if (sp.start > sp.end || (size_t)sp.end > N) {
            set_error("path " + std::to_string(i) + " has a step span outside the steps pool");
            return FLATGFA_ERR_BOUNDS;
        }
    }
    CAPI_HIP(hipSetDevice(device));
    // Everything is built into locals and handed to the handle only when all of it exists: a
    // failure half way (out of memory, say) leaves the handle as it was, and releases the rest.
    struct Image {
        int device = 0;
        hipStream_t stream = nullptr;
        uint32_t *steps = nullptr, *small = nullptr;
        uint32_t *pb = nullptr, *pe = nullptr, *seg_len = nullptr, *depth = nullptr, *uniq = nullptr, *sums = nullptr;  // inside `small`
        flatgfa_dev_plan_t *plan = nullptr;
        bool keep = false;
        ~Image() {
            if (keep) return;
            if (plan) flatgfa_dev_plan_destroy(plan);
            for (uint32_t *p : {steps, small})
                if (p) (void)hipFree(p);
            stream_release(device, stream);
        }
    } im;
    im.device = device;
    CAPI_HIP(stream_acquire(device, &im.stream));
    tick("device + stream");
    // AoS (packed, align-1) -> SoA.  Byte copies only: the file regions may be unaligned.  The
    // small arrays -- path spans, segment lengths, the two result vectors -- share one device
    // allocation and one copy: a hipMalloc costs about a millisecond whatever its size.
    const size_t Pa = (P + 63) & ~(size_t)63, Sa = (S + 63) & ~(size_t)63;  // 256-byte aligned sub-arrays
    std::vector<uint32_t> host(2 * Pa + Sa);
    uint32_t *h_pb = host.data(), *h_pe = host.data() + Pa, *h_len = host.data() + 2 * Pa;
    // (on a thread of its own, beside the upload of the steps: a million segments' lengths out of a freshly mapped file are a
    // millisecond of page faults that the link need not wait for)
    std::thread soa([&] {
        for (size_t i = 0; i < P; ++i) {
            h_pb[i] = v.paths[i].steps.start;
            h_pe[i] = v.paths[i].steps.end;
        }
        for (size_t i = 0; i < S; ++i) h_len[i] = v.segs[i].seq.len();
    });
    struct Joiner {
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{soa};
    if (N) {
        CAPI_HIP(hipMalloc(&im.steps, N * 4));
        tick("steps: hipMalloc");
        CAPI_HIP(upload(im.steps, v.steps.data, N * 4, im.stream));
    }
    tick("steps: upload");
    soa.join();
    tick("span arrays on the host (beside the upload)");
    if (P || S) {
        CAPI_HIP(hipMalloc(&im.small, (2 * Pa + 3 * Sa + 4 * Pa) * 4));  // (the last 4 * Pa words: two u64 sums per path)
        CAPI_HIP(fgfa_dev::plan_memcpy(im.small, host.data(), host.size() * 4, hipMemcpyHostToDevice));  // (through pinned staging, like the steps: pages the runtime pins for a copy cost the next dispatch when the host drops them, NOTES R6.6c)
        im.pb = im.small;
        im.pe = im.small + Pa;
        im.seg_len = im.small + 2 * Pa;
        im.depth = im.seg_len + Sa;
        im.uniq = im.depth + Sa;
        im.sums = im.uniq + Sa;  // (256-byte aligned: every sub-array is)
    }
    flatgfa_dev_graph_t g{im.steps, (uint64_t)N, im.pb, im.pe, (uint32_t)P, (uint32_t)S, im.seg_len};
    tick("paths, segments, outputs");
    const auto t_plan = std::chrono::steady_clock::now();
    int first_rc = FLATGFA_OK;
    im.plan = S ? flatgfa_dev_plan_create_first(&g, h_pb, h_pe, im.depth, im.uniq, &first_rc) : flatgfa_dev_plan_create(&g, h_pb, h_pe);
    const auto t_done = std::chrono::steady_clock::now();
    tick("plan (scratch + item lists)");
    if (!im.plan) return FLATGFA_ERR_HIP;  // the spans were checked above: what is left is the HIP runtime (see flatgfa_last_error)
    im.keep = true;
    cs->h2d_ms = std::chrono::duration<double, std::milli>(t_plan - t_enter).count();
    cs->plan_ms = std::chrono::duration<double, std::milli>(t_done - t_plan).count();
    cs->device = device;
    cs->stream = im.stream;
    cs->d_steps = im.steps;
    cs->d_small = im.small;
    cs->d_path_begin = im.pb;
    cs->d_path_end = im.pe;
    cs->d_seg_len = im.seg_len;
    cs->d_depth = im.depth;
    cs->d_uniq = im.uniq;
    cs->d_sums = reinterpret_cast<uint64_t *>(im.sums);
    cs->h_path_begin.assign(h_pb, h_pb + P);
    cs->h_path_end.assign(h_pe, h_pe + P);
    cs->h_soa.swap(host);  // (kept until the handle goes: four megabytes unmapped here would be ten milliseconds on the first query's copy)
    cs->plan = im.plan;
    cs->first_answer = S != 0;
    cs->first_rc = first_rc;
    cs->on_device = true;
    return FLATGFA_OK;
}

int flatgfa_to_device(flatgfa_t gfa, int device) {
    if (!gfa) { set_error("flatgfa_to_device: NULL handle"); return FLATGFA_ERR_ARG; }
    return ensure_device(gfa, device);
}

int flatgfa_residency_ms(flatgfa_t gfa, double *h2d_ms, double *plan_ms) {
    if (!gfa) { set_error("flatgfa_residency_ms: NULL handle"); return FLATGFA_ERR_ARG; }
    std::lock_guard<std::mutex> lk(gfa->dev_mu);
    if (!gfa->on_device) { set_error("flatgfa_residency_ms: the graph is not resident"); return FLATGFA_ERR_ARG; }
    if (h2d_ms) *h2d_ms = gfa->h2d_ms;
    if (plan_ms) *plan_ms = gfa->plan_ms;
    return FLATGFA_OK;
}

// Runs the node-depth kernels and leaves u32 results in cs->d_depth / cs->d_uniq.
static int run_seg_depth(CStore *cs, bool want_uniq) {
    int rc = ensure_device(cs, -1);
    if (rc) return rc;
    if (cs->first_answer) {  // (the query that sized the plan left both vectors here: the reference's consumers ask once per graph, cmds.rs:234-285)
        cs->first_answer = false;
        if (cs->first_rc) set_error("a step refers to a segment id (or a query to a path id) that is out of range");
        return cs->first_rc;
    }
    rc = flatgfa_dev_seg_depth(cs->plan, cs->d_depth, want_uniq ? cs->d_uniq : nullptr, cs->stream);
    if (rc) return rc;
    return flatgfa_dev_status(cs->plan, cs->stream);
}

static int fetch_widen(CStore *cs, const uint32_t *dev, uint64_t *out) {
    const size_t S = cs->view.segs.len;
    std::vector<uint32_t> tmp(S);
    if (S) CAPI_HIP(hipMemcpy(tmp.data(), dev, S * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < S; ++i) out[i] = tmp[i];  // Vec<usize>
    return FLATGFA_OK;
}

// Both result vectors as the device left them (u32; they lie next to each other in the handle's allocation, 256-byte aligned), fetched
// in ONE copy -- through one of the process's pinned staging buffers where they fit one: the link's rate, a pageable target gets a third
// of it -- and handed to `use(depth32, uniq32)`.
extern "C++" {
template <class F>
static int with_both_u32(CStore *gfa, F use) {
    const size_t S = gfa->view.segs.len, gap = (size_t)(gfa->d_uniq - gfa->d_depth);
    if ((gap + S) * 4 <= kChunk) {
        StagePool &pool = *stage_pool();
        std::lock_guard<std::mutex> lk(pool.mu);
        if (pool.ensure(gfa->device, 1) == hipSuccess) {
            CAPI_HIP(hipMemcpyAsync(pool.stage[0], gfa->d_depth, (gap + S) * 4, hipMemcpyDeviceToHost, gfa->stream));
            CAPI_HIP(hipStreamSynchronize(gfa->stream));
            const uint32_t *src = reinterpret_cast<const uint32_t *>(pool.stage[0]);
            return use(src, src + gap);
        }
        (void)hipGetLastError();
    }
    std::unique_ptr<uint32_t[]> tmp(new uint32_t[gap + S]);
    CAPI_HIP(hipMemcpy(tmp.get(), gfa->d_depth, (gap + S) * 4, hipMemcpyDeviceToHost));
    return use(tmp.get(), tmp.get() + gap);
}
}  // extern "C++"

int flatgfa_seg_depth(flatgfa_t gfa, uint64_t *depth_out, uint64_t *uniq_out) {
    if (!gfa || (!depth_out && gfa->view.segs.len)) { set_error("flatgfa_seg_depth: NULL argument"); return FLATGFA_ERR_ARG; }
    std::lock_guard<std::mutex> op(gfa->op_mu);
    int rc = run_seg_depth(gfa, uniq_out != nullptr);
    if (rc) return rc;
    if (!uniq_out) return fetch_widen(gfa, gfa->d_depth, depth_out);
    const size_t S = gfa->view.segs.len;
    if (!S) return FLATGFA_OK;
    return with_both_u32(gfa, [&](const uint32_t *d32, const uint32_t *u32) {  // widened to Vec<usize> on a few threads
        // (the caller's vectors are as a rule fresh pages: 16 MB of first touches for a million segments, which is what the threads share)
        const unsigned per = S >= (1u << 18) ? 4u : 1u;
        std::vector<std::thread> others;
        const auto widen = [&](unsigned k) {
            const uint32_t *src = k < per ? d32 : u32;
            uint64_t *dst = k < per ? depth_out : uniq_out;
            const unsigned j = k % per;
            for (size_t i = S * j / per, e = S * (j + 1) / per; i < e; ++i) dst[i] = src[i];
        };
        for (unsigned k = 1; k < 2 * per; ++k) others.emplace_back(widen, k);
        widen(0);
        for (auto &t : others) t.join();
        return (int)FLATGFA_OK;
    });
}

int flatgfa_path_depth(flatgfa_t gfa, const uint32_t *path_ids, uint32_t n_ids, uint64_t *length_out,
                       double *mean_out) {
    if (!gfa || (n_ids && (!path_ids || !length_out || !mean_out))) {
        set_error("flatgfa_path_depth: NULL argument");
        return FLATGFA_ERR_ARG;
    }
    for (uint32_t k = 0; k < n_ids; ++k)
        if (path_ids[k] >= gfa->view.paths.len) { set_error("flatgfa_path_depth: path id out of range"); return FLATGFA_ERR_BOUNDS; }
    // pass 1 over ALL paths (depth.rs:94-99), pass 2 over the requested ones (:104-108).  The device
    // forms every path's two sums in the pass that accumulates node depth (one read of the
    // steps); the requested ones are picked here.
    std::lock_guard<std::mutex> op(gfa->op_mu);
    int rc = ensure_device(gfa, -1);
    if (rc) return rc;
    const size_t P = gfa->view.paths.len;
    if (n_ids == 0 || P == 0) return run_seg_depth(gfa, false);
    uint64_t *d_sums = gfa->d_sums;  // (the handle's own: a hipMalloc per call cost five times the kernels it bracketed)
    std::vector<uint64_t> sums(P * 2);
    gfa->first_answer = false;  // (d_depth is written again)
    rc = flatgfa_dev_path_depth_all(gfa->plan, gfa->d_depth, d_sums, d_sums + P, gfa->stream);
    if (!rc) rc = flatgfa_dev_status(gfa->plan, gfa->stream);
    if (!rc && hipMemcpy(sums.data(), d_sums, P * 16, hipMemcpyDeviceToHost) != hipSuccess) rc = FLATGFA_ERR_HIP;
    if (rc) return rc;
    for (uint32_t k = 0; k < n_ids; ++k) {
        const uint64_t ln = sums[path_ids[k]], ws = sums[P + path_ids[k]];
        length_out[k] = ln;
        // the one floating-point operation on this path: depth.rs:129
        mean_out[k] = (double)ws / (double)ln;
    }
    return FLATGFA_OK;
}

int flatgfa_depth_table(flatgfa_t gfa, char **text, size_t *len) {
    if (!gfa || !text) { set_error("flatgfa_depth_table: NULL argument"); return FLATGFA_ERR_ARG; }
    const size_t S = gfa->view.segs.len;
    if (!S) {
        std::string out;
        fgfa::emit_seg_depth(gfa->view, nullptr, nullptr, &out);
        return give_text(out, text, len);
    }
    // the table straight from the device's 32-bit counts into the buffer the caller gets: no widened vectors, no intermediate string
    std::lock_guard<std::mutex> op(gfa->op_mu);
    int rc = run_seg_depth(gfa, true);
    if (rc) return rc;
    return with_both_u32(gfa, [&](const uint32_t *d32, const uint32_t *u32) {
        size_t n = 0;
        char *p = fgfa::emit_seg_depth_u32_malloc(gfa->view, d32, u32, &n);
        if (!p) { set_error("out of memory"); return (int)FLATGFA_ERR_IO; }
        *text = p;
        if (len) *len = n;
        return (int)FLATGFA_OK;
    });
}

int flatgfa_path_depth_table(flatgfa_t gfa, const uint32_t *path_ids, uint32_t n_ids, char **text, size_t *len) {
    if (!gfa || !text) { set_error("flatgfa_path_depth_table: NULL argument"); return FLATGFA_ERR_ARG; }
    std::vector<uint32_t> all;
    if (!path_ids) {
        all.resize(gfa->view.paths.len);
        for (size_t i = 0; i < all.size(); ++i) all[i] = (uint32_t)i;
        path_ids = all.data();
        n_ids = (uint32_t)all.size();
    }
    std::vector<uint64_t> lens(n_ids);
    std::vector<double> means(n_ids);
    int rc = flatgfa_path_depth(gfa, path_ids, n_ids, lens.data(), means.data());
    if (rc) return rc;
    std::string out;
    fgfa::emit_path_depth(gfa->view, path_ids, n_ids, lens.data(), means.data(), &out);
    return give_text(out, text, len);
}

int flatgfa_path_depth_bed(flatgfa_t gfa, const uint32_t *path_ids, uint32_t n_ids, char **text, size_t *len) {
    if (!gfa || !text) { set_error("flatgfa_path_depth_bed: NULL argument"); return FLATGFA_ERR_ARG; }
    std::vector<uint32_t> all;
    if (!path_ids) {
        all.resize(gfa->view.paths.len);
        for (size_t i = 0; i < all.size(); ++i) all[i] = (uint32_t)i;
        path_ids = all.data();
        n_ids = (uint32_t)all.size();
    }
    // as_bed keeps what path_depth computed as `lengths` (depth.rs:104-108, 173-183): same call, depths dropped
    std::vector<uint64_t> lens(n_ids);
    std::vector<double> means(n_ids);
    int rc = flatgfa_path_depth(gfa, path_ids, n_ids, lens.data(), means.data());
    if (rc) return rc;
    std::string out;
    for (uint32_t k = 0; k < n_ids; ++k) {
        const fgfa::Path &p = gfa->view.paths[path_ids[k]];
        out.append((const char *)gfa->view.name_data.data + p.name.start, p.name.len());
        out += "\t0\t" + std::to_string(lens[k]) + "\n";
    }
    return give_text(out, text, len);
}

// ---- f1: path-pair overlap (slow_odgi/overlap.py) ----

int flatgfa_path_overlaps(flatgfa_t gfa, const uint32_t *query_ids, uint32_t n_q, uint8_t *touch_out) {
    if (!gfa || (n_q && (!query_ids || !touch_out))) { set_error("flatgfa_path_overlaps: NULL argument"); return FLATGFA_ERR_ARG; }
    const size_t P = gfa->view.paths.len;
    for (uint32_t k = 0; k < n_q; ++k)
        if (query_ids[k] >= P) { set_error("flatgfa_path_overlaps: path id out of range"); return FLATGFA_ERR_BOUNDS; }
    if (n_q == 0 || P == 0) return FLATGFA_OK;
    std::lock_guard<std::mutex> op(gfa->op_mu);
    int rc = ensure_device(gfa, -1);
    if (rc) return rc;
    uint32_t *d_q = nullptr;
    uint8_t *d_t = nullptr;
    CAPI_HIP(hipMalloc(&d_q, (size_t)n_q * 4));
    if (hipMalloc(&d_t, (size_t)n_q * P) != hipSuccess) { (void)hipFree(d_q); set_error("hipMalloc failed"); return FLATGFA_ERR_HIP; }
    rc = FLATGFA_OK;
    if (hipMemcpyAsync(d_q, query_ids, (size_t)n_q * 4, hipMemcpyHostToDevice, gfa->stream) != hipSuccess) rc = FLATGFA_ERR_HIP;
    if (!rc) rc = flatgfa_dev_path_overlaps(gfa->plan, d_q, n_q, d_t, gfa->stream);
    if (!rc) rc = flatgfa_dev_status(gfa->plan, gfa->stream);
    if (!rc && hipMemcpy(touch_out, d_t, (size_t)n_q * P, hipMemcpyDeviceToHost) != hipSuccess) rc = FLATGFA_ERR_HIP;
    (void)hipFree(d_q);
    (void)hipFree(d_t);
    return rc;
}

int flatgfa_overlap_table(flatgfa_t gfa, const uint32_t *query_ids, uint32_t n_q, char **text, size_t *len) {
    if (!gfa || !text) { set_error("flatgfa_overlap_table: NULL argument"); return FLATGFA_ERR_ARG; }
    const size_t P = gfa->view.paths.len;
    std::vector<uint8_t> touch((size_t)n_q * P);
    int rc = flatgfa_path_overlaps(gfa, query_ids, n_q, touch.data());
    if (rc) return rc;
    if (!steps_name_segments(gfa)) return FLATGFA_ERR_BOUNDS;
    std::vector<uint64_t> plen(n_q);  // len(pathseq[ip]): the path's length in base pairs
    for (uint32_t k = 0; k < n_q; ++k) plen[k] = fgfa::path_length(gfa->view, query_ids[k]);
    std::string out;
    fgfa::emit_overlap(gfa->view, query_ids, n_q, plen.data(), touch.data(), &out);
    return give_text(out, text, len);
}

// ---- f2: window / interval depth (ops/window_depth.rs) ----

int flatgfa_interval_depth(flatgfa_t gfa, uint32_t path_index, const uint64_t *starts, const uint64_t *ends,
                           uint64_t n, double *depth_out) {
    if (!gfa || (n && (!starts || !ends || !depth_out))) { set_error("flatgfa_interval_depth: NULL argument"); return FLATGFA_ERR_ARG; }
    if (path_index >= gfa->view.paths.len) { set_error("flatgfa_interval_depth: path id out of range"); return FLATGFA_ERR_BOUNDS; }
    std::vector<uint64_t> depth(gfa->view.segs.len);
    int rc = flatgfa_seg_depth(gfa, depth.data(), nullptr);  // interval_depth calls seg_depth, window_depth.rs:177
    if (rc) return rc;
    std::vector<fgfa::BedEntry> win(n);
    for (uint64_t i = 0; i < n; ++i) win[i] = fgfa::BedEntry{0u, 0u, starts[i], ends[i]};
    fgfa::interval_depth(gfa->view, depth.data(), path_index, win.data(), n, depth_out);
    return FLATGFA_OK;
}

static int bed_depth_common(flatgfa_t gfa, uint32_t path_index, const fgfa::Bed &bed, char **text, size_t *len) {
    std::vector<uint64_t> depth(gfa->view.segs.len);
    int rc = flatgfa_seg_depth(gfa, depth.data(), nullptr);
    if (rc) return rc;
    std::vector<double> out(bed.entries.size());
    fgfa::interval_depth(gfa->view, depth.data(), path_index, bed.entries.data(), bed.entries.size(), out.data());
    std::string s;
    fgfa::emit_interval_depth(bed, out.data(), &s);
    return give_text(s, text, len);
}

int flatgfa_window_depth_table(flatgfa_t gfa, uint32_t path_index, uint64_t window, char **text, size_t *len) {
    if (!gfa || !text) { set_error("flatgfa_window_depth_table: NULL argument"); return FLATGFA_ERR_ARG; }
    if (path_index >= gfa->view.paths.len) { set_error("window depth: path not found"); return FLATGFA_ERR_BOUNDS; }
    if (window == 0) { set_error("window depth: window size must be positive"); return FLATGFA_ERR_ARG; }  // div_ceil by zero panics
    if (!steps_name_segments(gfa)) return FLATGFA_ERR_BOUNDS;
    const fgfa::Path &p = gfa->view.paths[path_index];
    fgfa::Bed bed;
    fgfa::make_windows(gfa->view.name_data.data + p.name.start, p.name.len(), 0, fgfa::path_length(gfa->view, path_index),
                       window, &bed);
    return bed_depth_common(gfa, path_index, bed, text, len);
}

int flatgfa_bed_depth_table(flatgfa_t gfa, const uint8_t *bed_text, size_t bed_len, char **text, size_t *len) {
    if (!gfa || !text || (bed_len && !bed_text)) { set_error("flatgfa_bed_depth_table: NULL argument"); return FLATGFA_ERR_ARG; }
    fgfa::Bed bed;
    std::string err;
    if (!fgfa::parse_bed(bed_text, bed_len, &bed, &err)) { set_error(err); return FLATGFA_ERR_BOUNDS; }
    if (bed.entries.empty()) { set_error("BED: no intervals"); return FLATGFA_ERR_BOUNDS; }  // entries.all()[0] panics
    // all intervals are taken to lie on the path named by the first entry (window_depth.rs:204-210)
    const fgfa::BedEntry &e0 = bed.entries[0];
    int64_t path = gfa->view.find_path(bed.name_data.data() + e0.name_start, e0.name_end - e0.name_start);
    if (path < 0) { set_error("BED: path not found in graph"); return FLATGFA_ERR_BOUNDS; }
    return bed_depth_common(gfa, (uint32_t)path, bed, text, len);
}

// ---- f3: node depth over a subset of paths (odgi depth -d -s) ----

int flatgfa_seg_depth_subset(flatgfa_t gfa, const uint32_t *path_ids, uint32_t n_ids, uint64_t *depth_out,
                             uint64_t *uniq_out) {
    if (!gfa || (n_ids && !path_ids) || (!depth_out && gfa->view.segs.len)) {
        set_error("flatgfa_seg_depth_subset: NULL argument");
        return FLATGFA_ERR_ARG;
    }
    const size_t P = gfa->view.paths.len, S = gfa->view.segs.len;
    for (uint32_t k = 0; k < n_ids; ++k)
        if (path_ids[k] >= P) { set_error("flatgfa_seg_depth_subset: path id out of range"); return FLATGFA_ERR_BOUNDS; }
    std::lock_guard<std::mutex> op(gfa->op_mu);
    int rc = ensure_device(gfa, -1);
    if (rc) return rc;
    // A prepared query over the same resident steps, with only the chosen paths' spans.  It is kept:
    // asking again for the same subset (a table after the vectors, say) costs the kernels alone.
    const bool same = gfa->sub_plan && gfa->sub_ids.size() == n_ids &&
                      std::equal(gfa->sub_ids.begin(), gfa->sub_ids.end(), path_ids);
    if (!same) {
        if (gfa->sub_plan) flatgfa_dev_plan_destroy(gfa->sub_plan);
        if (gfa->d_sub_spans) (void)hipFree(gfa->d_sub_spans);
        gfa->sub_plan = nullptr;
        gfa->d_sub_spans = nullptr;
        gfa->sub_ids.clear();
        std::vector<uint32_t> span(2 * (size_t)n_ids);
        for (uint32_t k = 0; k < n_ids; ++k) {
            span[k] = gfa->h_path_begin[path_ids[k]];
            span[n_ids + k] = gfa->h_path_end[path_ids[k]];
        }
        if (n_ids) {
            CAPI_HIP(hipMalloc(&gfa->d_sub_spans, span.size() * 4));
            if (hipMemcpy(gfa->d_sub_spans, span.data(), span.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(gfa->d_sub_spans);
                gfa->d_sub_spans = nullptr;
                set_error("flatgfa_seg_depth_subset: copying the path spans to the device failed");
                return FLATGFA_ERR_HIP;
            }
        }
        flatgfa_dev_graph_t g{gfa->d_steps, (uint64_t)gfa->view.steps.len, gfa->d_sub_spans,
                              gfa->d_sub_spans ? gfa->d_sub_spans + n_ids : nullptr, n_ids, (uint32_t)S, gfa->d_seg_len};
        gfa->sub_plan = flatgfa_dev_plan_create(&g, span.data(), span.data() + n_ids);
        if (!gfa->sub_plan) return FLATGFA_ERR_HIP;
        gfa->sub_ids.assign(path_ids, path_ids + n_ids);
    }
    gfa->first_answer = false;  // (the handle's result buffers now hold the subset's counts)
    rc = flatgfa_dev_seg_depth(gfa->sub_plan, gfa->d_depth, uniq_out ? gfa->d_uniq : nullptr, gfa->stream);
    if (!rc) rc = flatgfa_dev_status(gfa->sub_plan, gfa->stream);
    if (!rc) rc = fetch_widen(gfa, gfa->d_depth, depth_out);
    if (!rc && uniq_out) rc = fetch_widen(gfa, gfa->d_uniq, uniq_out);
    return rc;
}

}  // extern "C"
