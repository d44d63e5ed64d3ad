// flatgfa_sharded_*: node depth / unique depth / path depth of ONE graph sharded over the GPUs
// of a node by one process (include/flatgfa.h Part 2).
//
// depth and uniq are sums over paths of per-path contributions (ops/depth.rs:25-36), so the steps
// are cut into `n_shards` contiguous stretches of near-equal size, at path boundaries where one is
// near the even cut and inside a path where none is (fewer paths than shards; one path longer
// than a shard's share).  Every shard lives on its own device -- a slice of the steps, rebased
// spans, the segment lengths, a depth plan, a stream, a host thread that enqueues for it -- and
// one call is: the local kernels on every shard, ONE all-reduce (RCCL, ncclUint32 / ncclSum) of the
// fused [depth | uniq | touch] vector, and a fix-up of uniq for the K paths that were cut: a piece
// counts as a path of its own on its shard, so a segment touched by m pieces of one path was counted
// m times; touch_k (the 0/1 vector "piece of split path k touches s", which is unique depth over
// that piece alone) sums to m, and uniq[s] -= max(m - 1, 0).  Exact: integer sums.  The K touch
// vectors travel PACKED: a path has at most n_shards pieces, so its count needs b = bits(n_shards)
// bits, and 32 / b counters share a word -- with eight shards all K <= 7 of them fit ONE u32 per
// segment (the collective carries 12 bytes per segment, not 8 + 4 K); fields cannot carry into each
// other because no sum exceeds n_shards.
// Path depth needs no reduction beyond the node depth: every shard measures its pieces against
// the reduced vector and the host adds the pieces of a cut path up (both sums of
// measure_path, depth.rs:116-131, are sums over steps).
//
// RCCL is loaded at first use (dlopen), not linked: librccl.so is 570 MB, and mapping it costs
// half a second that every `fgfa` run and every load of this library would pay for a collective
// only multi-GPU callers use.  When two shards share a device (tests on a one-GPU box) the
// exchange is a device-side add instead; FLATGFA_SHARD_FORCE_RCCL=1 makes a one-shard handle go
// through a communicator of size one, so that the RCCL route runs there too.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums; every call goes through the table below

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/flatgfa.h"
#include "device_common.hpp"
#include "flatgfa_core.hpp"

using fgfa_dev::set_error;

const fgfa::View &flatgfa_capi_view(flatgfa_t gfa);  // capi.cpp

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// The process's RCCL, or null with the reason in *why.
const Rccl *rccl(std::string *why) {
    static std::mutex mu;
    static Rccl r;
    static std::string err;
    std::lock_guard<std::mutex> lk(mu);
    if (!r.lib && err.empty()) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            err = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?");
        } else {
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
            if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
                err = "librccl.so lacks ncclCommInitAll / ncclAllReduce";
                r.lib = nullptr;
            }
        }
    }
    if (!r.lib) {
        if (why) *why = err;
        return nullptr;
    }
    return &r;
}

// uniq[s] -= sum over the split paths of (pieces that touch s) - 1.  Split path k's count is field k % per_word
// (`bits` wide) of word k / per_word of the packed touch vector.
__global__ __launch_bounds__(256) void k_fix_uniq(uint32_t *__restrict__ uniq, const uint32_t *__restrict__ touch, uint32_t n_segs, uint32_t n_split,
                                                  uint32_t bits, uint32_t per_word) {
    const uint32_t mask = (1u << bits) - 1u;
    for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < n_segs; s += gridDim.x * 256u) {
        uint32_t over = 0;
        for (uint32_t k0 = 0; k0 < n_split; k0 += per_word) {
            uint32_t w = touch[(size_t)(k0 / per_word) * n_segs + s];
            for (uint32_t k = k0; k < n_split && k < k0 + per_word && w; ++k, w >>= bits) {
                const uint32_t m = w & mask;
                over += m > 1u ? m - 1u : 0u;
            }
        }
        if (over) uniq[s] -= over;
    }
}

// a piece's 0/1 touch vector into its field of the packed word (the packed vector is zeroed at the start of every call)
__global__ __launch_bounds__(256) void k_pack_touch(uint32_t *__restrict__ packed, const uint32_t *__restrict__ touch, uint32_t n_segs, uint32_t shift) {
    for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < n_segs; s += gridDim.x * 256u) {
        const uint32_t t = touch[s];
        if (t) packed[s] |= (t & 1u) << shift;  // (a shard's pieces belong to different split paths: no two of its kernels write one field)
    }
}

__global__ __launch_bounds__(256) void k_add_into(uint32_t *__restrict__ acc, const uint32_t *__restrict__ x, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) acc[i] += x[i];
}

// The calling thread's current device is its own business: whatever the entry points switch to, they switch back.
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

struct Piece {
    uint32_t path;    // the path this is (a piece of)
    uint32_t lb, le;  // its steps in the shard's slice
    int split;        // -1, or the ordinal of the split path it is a piece of
};

struct Shard {
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t step_lo = 0, step_hi = 0;  // the stretch of the steps pool the slice covers
    std::vector<Piece> pieces;
    uint32_t *d_steps = nullptr, *d_small = nullptr;  // d_small: [path_begin | path_end | seg_len | ids]
    uint32_t *d_pb = nullptr, *d_pe = nullptr, *d_seg_len = nullptr, *d_ids = nullptr;
    flatgfa_dev_plan_t *plan = nullptr;
    struct Touch {
        int split;
        uint32_t *d_span = nullptr;  // {begin, end} of the piece
        flatgfa_dev_plan_t *plan = nullptr;
    };
    std::vector<Touch> touch;
    uint32_t *d_send = nullptr, *d_recv = nullptr, *d_tmp = nullptr;  // [(2 + W) * S] each (W words of packed touch counters per segment); 2 S scratch
    uint64_t *d_sums = nullptr;                                       // [2 * pieces]
    uint32_t *d_ones = nullptr;                                       // [2] flatgfa_sharded_ranks_seen: this shard's one, and the sum
    uint32_t ranks_seen = 0;
    ncclComm_t comm = nullptr;
    // the shard's host thread
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int cmd = 0;  // 0 = idle
    bool busy = false;
    int rc = FLATGFA_OK;
    std::string err;
};

enum { kCmdQuit = 1, kCmdLocal, kCmdExchange, kCmdSync, kCmdPathSums, kCmdCountRanks };

}  // namespace

struct flatgfa_sharded {
    flatgfa_t gfa = nullptr;
    uint32_t S = 0, P = 0, K = 0;
    uint32_t bits = 1, per_word = 32, W = 0;  // a touch counter's width, counters per word, words per segment (see k_fix_uniq)
    int n = 0;
    bool use_rccl = false;
    bool with_uniq = true;  // of the call in flight
    std::vector<std::unique_ptr<Shard>> sh;
    std::vector<uint32_t> split_paths;  // path id of every split ordinal
    std::mutex op_mu;
};

namespace {

#define SH_HIP(expr)                                                                    \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            s.err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return FLATGFA_ERR_HIP;                                                     \
        }                                                                               \
    } while (0)

size_t vec_count(const flatgfa_sharded &h, bool with_uniq) { return (size_t)h.S * (with_uniq ? 2 + h.W : 1); }

// the local kernels of one call on shard s (its thread; its device is current)
int shard_local(flatgfa_sharded &h, Shard &s) {
    if (h.S == 0) return FLATGFA_OK;
    int rc = flatgfa_dev_seg_depth(s.plan, s.d_send, h.with_uniq ? s.d_send + h.S : nullptr, s.stream);
    if (rc) { s.err = flatgfa_last_error(); return rc; }
    if (h.with_uniq && h.W) {
        SH_HIP(hipMemsetAsync(s.d_send + 2 * (size_t)h.S, 0, (size_t)h.W * h.S * 4, s.stream));
        for (Shard::Touch &t : s.touch) {
            rc = flatgfa_dev_seg_depth(t.plan, s.d_tmp, s.d_tmp + h.S, s.stream);  // (unique depth over one piece: 0 or 1)
            if (rc) { s.err = flatgfa_last_error(); return rc; }
            const uint32_t k = (uint32_t)t.split;
            hipLaunchKernelGGL(k_pack_touch, dim3(std::min<uint32_t>((h.S + 255u) / 256u, 2048u)), dim3(256), 0, s.stream,
                               s.d_send + (size_t)(2 + k / h.per_word) * h.S, s.d_tmp + h.S, h.S, (k % h.per_word) * h.bits);
            SH_HIP(hipGetLastError());
        }
    }
    return FLATGFA_OK;
}

// the collective and the fix-up, enqueued behind the local kernels
int shard_exchange(flatgfa_sharded &h, Shard &s) {
    if (h.S == 0) return FLATGFA_OK;
    if (h.use_rccl) {
        const Rccl *r = rccl(nullptr);
        const ncclResult_t e = r->AllReduce(s.d_send, s.d_recv, vec_count(h, h.with_uniq), ncclUint32, ncclSum, s.comm, s.stream);
        if (e != ncclSuccess) { s.err = std::string("ncclAllReduce: ") + r->GetErrorString(e); return FLATGFA_ERR_HIP; }
    }
    if (h.with_uniq && h.K && (h.use_rccl || h.n == 1)) {
        hipLaunchKernelGGL(k_fix_uniq, dim3(std::min<uint32_t>((h.S + 255u) / 256u, 2048u)), dim3(256), 0, s.stream, s.d_recv + h.S,
                           s.d_recv + 2 * (size_t)h.S, h.S, h.K, h.bits, h.per_word);
        SH_HIP(hipGetLastError());
    }
    return FLATGFA_OK;
}

int shard_sync(flatgfa_sharded &h, Shard &s) {
    int rc = flatgfa_dev_status(s.plan, s.stream);
    if (rc) { s.err = flatgfa_last_error(); return rc; }
    for (Shard::Touch &t : s.touch) {
        rc = flatgfa_dev_status(t.plan, s.stream);
        if (rc) { s.err = flatgfa_last_error(); return rc; }
    }
    return FLATGFA_OK;
}

// measure_path's two sums for every piece of the shard, against the reduced node depth
int shard_path_sums(flatgfa_sharded &h, Shard &s) {
    if (s.pieces.empty()) return FLATGFA_OK;
    const uint32_t np = (uint32_t)s.pieces.size();
    int rc = flatgfa_dev_path_sums(s.plan, s.d_ids, np, s.d_recv, s.d_sums, s.d_sums + np, s.stream);
    if (rc) { s.err = flatgfa_last_error(); return rc; }
    return shard_sync(h, s);
}

// Where the steps are cut (host only; flatgfa_shard_cuts exposes it to the CPU test suite).  `cum[p]` = path steps
// before path p, counted along the path order.  Cut r lies at the path boundary nearest to the even cut
// T * r / n when that is within an eighth of a shard's share of it -- or whenever paths must stay whole --
// and inside the path otherwise.
void shard_cuts(const std::vector<uint64_t> &cum, int n_shards, bool may_split, std::vector<uint64_t> *cut_out) {
    const size_t P = cum.size() - 1;
    const uint64_t T = cum[P];
    std::vector<uint64_t> &cut = *cut_out;
    cut.assign((size_t)n_shards + 1, 0);
    cut[(size_t)n_shards] = T;
    for (int r = 1; r < n_shards; ++r) {
        const uint64_t target = (uint64_t)((__uint128_t)T * r / n_shards);
        size_t p = (size_t)(std::upper_bound(cum.begin(), cum.end(), target) - cum.begin());  // cum[p - 1] <= target < cum[p]
        p = p ? p - 1 : 0;
        const uint64_t lo = cum[p], hi = cum[std::min(p + 1, P)];
        const uint64_t near = target - lo <= hi - target ? lo : hi;
        const uint64_t off = near > target ? near - target : target - near;
        const uint64_t c = (!may_split || off * 8 * n_shards <= T) ? near : target;
        cut[(size_t)r] = std::max(c, cut[(size_t)r - 1]);
    }
}

// How many ranks the exchange really spans: every shard contributes a one.
int shard_count_ranks(flatgfa_sharded &h, Shard &s) {
    if (!s.d_ones) SH_HIP(hipMalloc(&s.d_ones, 8));
    const uint32_t one[2] = {1u, 0u};
    SH_HIP(hipMemcpyAsync(s.d_ones, one, 8, hipMemcpyHostToDevice, s.stream));
    if (h.use_rccl) {
        const Rccl *r = rccl(nullptr);
        const ncclResult_t e = r->AllReduce(s.d_ones, s.d_ones + 1, 1, ncclUint32, ncclSum, s.comm, s.stream);
        if (e != ncclSuccess) { s.err = std::string("ncclAllReduce: ") + r->GetErrorString(e); return FLATGFA_ERR_HIP; }
    } else {  // (flatgfa_sharded_ranks_seen sums them the way the vectors are summed, then reads every shard's copy)
        SH_HIP(hipStreamSynchronize(s.stream));
        return FLATGFA_OK;
    }
    SH_HIP(hipMemcpyAsync(&s.ranks_seen, s.d_ones + 1, 4, hipMemcpyDeviceToHost, s.stream));
    SH_HIP(hipStreamSynchronize(s.stream));
    return FLATGFA_OK;
}

void shard_thread(flatgfa_sharded *h, Shard *s) {
    (void)hipSetDevice(s->device);
    for (;;) {
        int cmd;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->cmd != 0; });
            cmd = s->cmd;
        }
        int rc = FLATGFA_OK;
        s->err.clear();
        if (cmd == kCmdLocal) rc = shard_local(*h, *s);
        else if (cmd == kCmdExchange) rc = shard_exchange(*h, *s);
        else if (cmd == kCmdSync) rc = shard_sync(*h, *s);
        else if (cmd == kCmdPathSums) rc = shard_path_sums(*h, *s);
        else if (cmd == kCmdCountRanks) rc = shard_count_ranks(*h, *s);
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->rc = rc;
            s->cmd = 0;
            s->busy = false;
        }
        s->cv.notify_all();
        if (cmd == kCmdQuit) return;
    }
}

// every shard's thread runs `cmd`; returns the first failure (and sets the caller's error text)
int run_all(flatgfa_sharded &h, int cmd) {
    for (auto &s : h.sh) {
        std::lock_guard<std::mutex> lk(s->mu);
        s->cmd = cmd;
        s->busy = true;
        s->cv.notify_all();
    }
    int rc = FLATGFA_OK;
    for (auto &s : h.sh) {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return !s->busy; });
        if (s->rc && !rc) {
            rc = s->rc;
            set_error("shard on device " + std::to_string(s->device) + ": " + s->err);
        }
    }
    return rc;
}

// Shards that share a device (or a handle without RCCL): the sum of every shard's `cnt` words at send(shard) on shard 0's
// device, `then` behind it on that stream, the result handed back to every recv(shard).
template <class Send, class Recv, class Then>
int sum_by_adds(flatgfa_sharded &h, size_t cnt, Send send, Recv recv, Then then) {
    DeviceGuard guard;
    Shard &root = *h.sh[0];
    Shard &s = root;  // (for SH_HIP's error text)
    for (auto &x : h.sh) {
        SH_HIP(hipSetDevice(x->device));
        SH_HIP(hipStreamSynchronize(x->stream));
    }
    SH_HIP(hipSetDevice(root.device));
    SH_HIP(hipMemcpyAsync(recv(root), send(root), cnt * 4, hipMemcpyDeviceToDevice, root.stream));
    struct Bounce {  // (shards on other devices are copied here first)
        uint32_t *p = nullptr;
        ~Bounce() { if (p) (void)hipFree(p); }
    } bounce;
    for (int i = 1; i < h.n; ++i) {
        const uint32_t *src = send(*h.sh[i]);
        if (h.sh[i]->device != root.device) {
            if (!bounce.p) SH_HIP(hipMalloc(&bounce.p, cnt * 4));
            SH_HIP(hipMemcpyPeerAsync(bounce.p, root.device, src, h.sh[i]->device, cnt * 4, root.stream));
            src = bounce.p;
        }
        hipLaunchKernelGGL(k_add_into, dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 2048)), dim3(256), 0, root.stream, recv(root), src, cnt);
    }
    then(root);
    SH_HIP(hipGetLastError());
    for (int i = 1; i < h.n; ++i)
        SH_HIP(hipMemcpyPeerAsync(recv(*h.sh[i]), h.sh[i]->device, recv(root), root.device, cnt * 4, root.stream));
    SH_HIP(hipStreamSynchronize(root.stream));
    return FLATGFA_OK;
}

int exchange_by_adds(flatgfa_sharded &h) {
    if (h.S == 0) return FLATGFA_OK;
    return sum_by_adds(
        h, vec_count(h, h.with_uniq), [](Shard &x) { return x.d_send; }, [](Shard &x) { return x.d_recv; },
        [&](Shard &root) {
            if (h.with_uniq && h.K)
                hipLaunchKernelGGL(k_fix_uniq, dim3(std::min<uint32_t>((h.S + 255u) / 256u, 2048u)), dim3(256), 0, root.stream,
                                   root.d_recv + h.S, root.d_recv + 2 * (size_t)h.S, h.S, h.K, h.bits, h.per_word);
        });
}

int enqueue_locked(flatgfa_sharded &h, bool with_uniq) {
    h.with_uniq = with_uniq;
    int rc = run_all(h, kCmdLocal);
    if (rc) return rc;
    if (h.use_rccl || h.n == 1) return run_all(h, kCmdExchange);  // (one shard without RCCL: send is recv)
    Shard &s = *h.sh[0];
    rc = exchange_by_adds(h);
    if (rc) set_error(s.err);
    return rc;
}

}  // namespace

extern "C" {

void flatgfa_sharded_free(flatgfa_sharded_t *h) {
    if (!h) return;
    DeviceGuard guard;
    for (auto &s : h->sh) {
        if (s->th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(s->mu);
                s->cmd = kCmdQuit;
                s->busy = true;
            }
            s->cv.notify_all();
            s->th.join();
        }
        (void)hipSetDevice(s->device);
        if (s->comm) {
            if (const Rccl *r = rccl(nullptr)) (void)r->CommDestroy(s->comm);
        }
        if (s->plan) flatgfa_dev_plan_destroy(s->plan);
        for (Shard::Touch &t : s->touch) {
            if (t.plan) flatgfa_dev_plan_destroy(t.plan);
            if (t.d_span) (void)hipFree(t.d_span);
        }
        for (void *p : {(void *)s->d_steps, (void *)s->d_small, (void *)(s->d_recv != s->d_send ? s->d_recv : nullptr), (void *)s->d_send, (void *)s->d_tmp,
                        (void *)s->d_sums, (void *)s->d_ones})
            if (p) (void)hipFree(p);
        if (s->stream) (void)hipStreamDestroy(s->stream);
    }
    delete h;
}

flatgfa_sharded_t *flatgfa_sharded_create(flatgfa_t gfa, const int *devices, int n_shards, unsigned flags) {
    if (!gfa || n_shards < 1 || n_shards > 64) { set_error("flatgfa_sharded_create: bad argument"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device is visible; the depth queries have no CPU fallback");
        return nullptr;
    }
    DeviceGuard guard;
    const fgfa::View &v = flatgfa_capi_view(gfa);
    if (v.steps.len > 0xFFFFFFFFull || v.segs.len > 0x80000000ull || v.paths.len > 0xFFFFFFFFull) {
        set_error("graph too large for 32-bit ids");
        return nullptr;
    }
    const size_t N = v.steps.len, P = v.paths.len, S = v.segs.len;
    auto h = std::unique_ptr<flatgfa_sharded, void (*)(flatgfa_sharded *)>(new flatgfa_sharded(), flatgfa_sharded_free);
    h->gfa = gfa;
    h->S = (uint32_t)S;
    h->P = (uint32_t)P;
    h->n = n_shards;
    bool distinct = true;
    for (int i = 0; i < n_shards; ++i) {
        auto s = std::make_unique<Shard>();
        s->device = devices ? devices[i] : i % ndev;
        if (s->device < 0 || s->device >= ndev) { set_error("flatgfa_sharded_create: device index out of range"); return nullptr; }
        for (auto &o : h->sh) distinct = distinct && o->device != s->device;
        h->sh.push_back(std::move(s));
    }
    // ---- where the steps are cut ----
    // Path spans in path order, checked as the single-device route does (pool.rs:341-347 would panic).
    bool ordered = true;  // every path's steps lie behind the path before it: a shard is a stretch of the pool
    std::vector<uint64_t> cum(P + 1, 0);
    for (size_t p = 0; p < P; ++p) {
        const fgfa::Span sp = v.paths[p].steps;
        if (sp.start > sp.end || (size_t)sp.end > N) {
            set_error("path " + std::to_string(p) + " has a step span outside the steps pool");
            return nullptr;
        }
        if (p && sp.start < v.paths[p - 1].steps.end) ordered = false;
        cum[p + 1] = cum[p] + sp.len();
    }
    const uint64_t T = cum[P];
    const bool may_split = ordered && !(flags & FLATGFA_SHARD_WHOLE_PATHS);
    std::vector<uint64_t> cut;
    shard_cuts(cum, n_shards, may_split, &cut);
    // ---- the shards' pieces ----
    std::vector<int> split_of(P, -1);
    {
        size_t p = 0;
        for (int r = 0; r < n_shards; ++r) {
            Shard &s = *h->sh[r];
            const uint64_t a = cut[r], b = cut[r + 1];
            while (p < P && cum[p + 1] <= a) ++p;  // paths that end before the stretch (empty ones among them: they have no steps to walk anywhere)
            uint64_t lo_step = ~0ull, hi_step = 0;
            std::vector<std::pair<uint32_t, std::pair<uint64_t, uint64_t>>> mine;  // path, the piece's stretch of the steps pool
            for (size_t q = p; q < P && cum[q] < b; ++q) {
                const uint64_t pa = std::max(cum[q], a), pb = std::min(cum[q + 1], b);
                if (pa >= pb) continue;
                const uint64_t g0 = v.paths[q].steps.start + (pa - cum[q]), g1 = v.paths[q].steps.start + (pb - cum[q]);
                mine.push_back({(uint32_t)q, {g0, g1}});
                if ((pa != cum[q] || pb != cum[q + 1]) && split_of[q] < 0) {
                    split_of[q] = (int)h->split_paths.size();
                    h->split_paths.push_back((uint32_t)q);
                }
                lo_step = std::min(lo_step, g0);
                hi_step = std::max(hi_step, g1);
            }
            if (lo_step > hi_step) lo_step = hi_step = 0;
            s.step_lo = lo_step;
            s.step_hi = hi_step;
            for (auto &m : mine) s.pieces.push_back(Piece{m.first, (uint32_t)(m.second.first - lo_step), (uint32_t)(m.second.second - lo_step), -1});
        }
        for (auto &s : h->sh)
            for (Piece &pc : s->pieces) pc.split = split_of[pc.path];
    }
    h->K = (uint32_t)h->split_paths.size();
    // a split path has at most n_shards pieces: its touch count fits bits(n_shards) bits
    h->bits = 1;
    while ((1u << h->bits) <= (uint32_t)n_shards) h->bits += 1;
    h->per_word = 32u / h->bits;
    h->W = (h->K + h->per_word - 1) / h->per_word;
    if (S == 0 && T != 0) {  // (as the single-device route: a step names a segment, and there is none)
        set_error("a step refers to a segment id that is out of range");
        return nullptr;
    }
    // ---- RCCL, or adds ----
    const bool force = getenv("FLATGFA_SHARD_FORCE_RCCL") != nullptr;
    h->use_rccl = distinct && (n_shards > 1 || force) && !(flags & FLATGFA_SHARD_NO_RCCL);
    if (h->use_rccl) {
        std::string why;
        const Rccl *r = rccl(&why);
        if (!r) { set_error(why); return nullptr; }
        std::vector<int> devs;
        for (auto &s : h->sh) devs.push_back(s->device);
        std::vector<ncclComm_t> comms(n_shards);
        const ncclResult_t e = r->CommInitAll(comms.data(), n_shards, devs.data());
        if (e != ncclSuccess) { set_error(std::string("ncclCommInitAll: ") + r->GetErrorString(e)); return nullptr; }
        for (int i = 0; i < n_shards; ++i) h->sh[i]->comm = comms[i];
    }
    // ---- the shards' images ----
    const size_t cnt = (size_t)S * (2 + h->W);
    for (auto &sp : h->sh) {
        Shard &s = *sp;
#define CR_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
            return nullptr;                                                                   \
        }                                                                                     \
    } while (0)
        CR_HIP(hipSetDevice(s.device));
        CR_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        const size_t n_loc = (size_t)(s.step_hi - s.step_lo), np = s.pieces.size();
        const size_t Pa = (np + 63) & ~(size_t)63, Sa = (S + 63) & ~(size_t)63;
        std::vector<uint32_t> host(3 * Pa + Sa, 0);
        for (size_t i = 0; i < np; ++i) {
            host[i] = s.pieces[i].lb;
            host[Pa + i] = s.pieces[i].le;
            host[2 * Pa + Sa + i] = (uint32_t)i;
        }
        for (size_t i = 0; i < S; ++i) host[2 * Pa + i] = v.segs[i].seq.len();
        if (n_loc) {
            CR_HIP(hipMalloc(&s.d_steps, n_loc * 4));
            CR_HIP(hipMemcpy(s.d_steps, (const char *)v.steps.data + s.step_lo * 4, n_loc * 4, hipMemcpyHostToDevice));  // (byte copy: the pool may be unaligned)
        }
        CR_HIP(hipMalloc(&s.d_small, std::max<size_t>(host.size(), 1) * 4));
        CR_HIP(hipMemcpy(s.d_small, host.data(), host.size() * 4, hipMemcpyHostToDevice));
        s.d_pb = s.d_small;
        s.d_pe = s.d_small + Pa;
        s.d_seg_len = s.d_small + 2 * Pa;
        s.d_ids = s.d_small + 2 * Pa + Sa;
        flatgfa_dev_graph_t g{s.d_steps, (uint64_t)n_loc, s.d_pb, s.d_pe, (uint32_t)np, (uint32_t)S, s.d_seg_len};
        s.plan = flatgfa_dev_plan_create(&g, host.data(), host.data() + Pa);
        if (!s.plan) return nullptr;
        for (size_t i = 0; i < np; ++i) {
            if (s.pieces[i].split < 0) continue;
            Shard::Touch t;
            t.split = s.pieces[i].split;
            const uint32_t span[2] = {s.pieces[i].lb, s.pieces[i].le};
            CR_HIP(hipMalloc(&t.d_span, 8));
            s.touch.push_back(t);  // (owned by the shard from here on: freed with it whatever fails below)
            Shard::Touch &tt = s.touch.back();
            CR_HIP(hipMemcpy(tt.d_span, span, 8, hipMemcpyHostToDevice));
            flatgfa_dev_graph_t gt{s.d_steps, (uint64_t)n_loc, tt.d_span, tt.d_span + 1, 1u, (uint32_t)S, s.d_seg_len};
            tt.plan = flatgfa_dev_plan_create(&gt, span, span + 1);
            if (!tt.plan) return nullptr;
        }
        if (S) {
            CR_HIP(hipMalloc(&s.d_send, cnt * 4));
            CR_HIP(hipMemset(s.d_send, 0, cnt * 4));  // (the touch vectors of the split paths this shard holds no piece of stay zero)
            CR_HIP(hipStreamSynchronize(nullptr));     // (the shard's own stream does not wait for the null stream)
            if (n_shards == 1 && !h->use_rccl) {
                s.d_recv = s.d_send;
            } else {
                CR_HIP(hipMalloc(&s.d_recv, cnt * 4));
            }
            if (!s.touch.empty()) CR_HIP(hipMalloc(&s.d_tmp, 2 * S * 4));
        }
        if (np) CR_HIP(hipMalloc(&s.d_sums, np * 16));
#undef CR_HIP
    }
    for (auto &s : h->sh) s->th = std::thread(shard_thread, h.get(), s.get());
    return h.release();
}

int flatgfa_sharded_layout(flatgfa_sharded_t *h, int shard, int *device, uint64_t *step_begin, uint64_t *step_end, uint32_t *first_path,
                           uint32_t *n_pieces, uint32_t *n_split_paths, int *uses_rccl) {
    if (!h || shard < 0 || shard >= h->n) { set_error("flatgfa_sharded_layout: bad argument"); return FLATGFA_ERR_ARG; }
    const Shard &s = *h->sh[shard];
    if (device) *device = s.device;
    if (step_begin) *step_begin = s.step_lo;
    if (step_end) *step_end = s.step_hi;
    if (first_path) *first_path = s.pieces.empty() ? h->P : s.pieces.front().path;
    if (n_pieces) *n_pieces = (uint32_t)s.pieces.size();
    if (n_split_paths) *n_split_paths = h->K;
    if (uses_rccl) *uses_rccl = h->use_rccl ? 1 : 0;
    return FLATGFA_OK;
}

int flatgfa_shard_cuts(const uint64_t *path_steps, uint32_t n_paths, int n_shards, unsigned flags, uint64_t *cuts_out) {
    if ((n_paths && !path_steps) || n_shards < 1 || n_shards > 64 || !cuts_out) { set_error("flatgfa_shard_cuts: bad argument"); return FLATGFA_ERR_ARG; }
    std::vector<uint64_t> cum((size_t)n_paths + 1, 0), cut;
    for (uint32_t p = 0; p < n_paths; ++p) cum[p + 1] = cum[p] + path_steps[p];
    shard_cuts(cum, n_shards, !(flags & FLATGFA_SHARD_WHOLE_PATHS), &cut);
    for (int r = 0; r <= n_shards; ++r) cuts_out[r] = cut[(size_t)r];
    return FLATGFA_OK;
}

int flatgfa_sharded_ranks_seen(flatgfa_sharded_t *h) {
    if (!h) { set_error("flatgfa_sharded_ranks_seen: NULL handle"); return FLATGFA_ERR_ARG; }
    std::lock_guard<std::mutex> lk(h->op_mu);
    const int rc = run_all(*h, kCmdCountRanks);
    if (rc) return rc;
    if (!h->use_rccl) {  // the ones go the way the vectors go (exchange_by_adds), and every shard reads its own copy of the sum
        if (h->n > 1) {
            const int rc2 = sum_by_adds(*h, 1, [](Shard &x) { return x.d_ones; }, [](Shard &x) { return x.d_ones + 1; }, [](Shard &) {});
            if (rc2) { set_error(h->sh[0]->err); return rc2; }
        }
        DeviceGuard guard;
        for (auto &s : h->sh) {
            if (hipSetDevice(s->device) != hipSuccess ||
                hipMemcpy(&s->ranks_seen, s->d_ones + (h->n > 1 ? 1 : 0), 4, hipMemcpyDeviceToHost) != hipSuccess) {
                set_error("flatgfa_sharded_ranks_seen: reading a shard's count failed");
                return FLATGFA_ERR_HIP;
            }
        }
    }
    for (auto &s : h->sh)
        if (s->ranks_seen != h->sh[0]->ranks_seen) { set_error("flatgfa_sharded_ranks_seen: the shards disagree"); return FLATGFA_ERR_HIP; }
    return (int)h->sh[0]->ranks_seen;
}

int flatgfa_sharded_enqueue(flatgfa_sharded_t *h, int with_uniq) {
    if (!h) { set_error("flatgfa_sharded_enqueue: NULL handle"); return FLATGFA_ERR_ARG; }
    std::lock_guard<std::mutex> lk(h->op_mu);
    return enqueue_locked(*h, with_uniq != 0);
}

int flatgfa_sharded_sync(flatgfa_sharded_t *h) {
    if (!h) { set_error("flatgfa_sharded_sync: NULL handle"); return FLATGFA_ERR_ARG; }
    std::lock_guard<std::mutex> lk(h->op_mu);
    return run_all(*h, kCmdSync);
}

static int fetch_locked(flatgfa_sharded_t *h, int shard, uint64_t *depth_out, uint64_t *uniq_out);

int flatgfa_sharded_fetch(flatgfa_sharded_t *h, int shard, uint64_t *depth_out, uint64_t *uniq_out) {
    if (!h || shard < 0 || shard >= h->n || (!depth_out && h->S)) { set_error("flatgfa_sharded_fetch: bad argument"); return FLATGFA_ERR_ARG; }
    std::lock_guard<std::mutex> lk(h->op_mu);
    return fetch_locked(h, shard, depth_out, uniq_out);
}

// (the caller holds op_mu)
static int fetch_locked(flatgfa_sharded_t *h, int shard, uint64_t *depth_out, uint64_t *uniq_out) {
    if (uniq_out && !h->with_uniq) { set_error("flatgfa_sharded_fetch: the last call computed node depth only"); return FLATGFA_ERR_ARG; }
    Shard &s = *h->sh[shard];
    if (h->S == 0) return FLATGFA_OK;
    DeviceGuard guard;
    std::vector<uint32_t> tmp((size_t)h->S * (uniq_out ? 2 : 1));
    if (hipSetDevice(s.device) != hipSuccess || hipMemcpy(tmp.data(), s.d_recv, tmp.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("flatgfa_sharded_fetch: device to host copy failed");
        return FLATGFA_ERR_HIP;
    }
    for (uint32_t i = 0; i < h->S; ++i) depth_out[i] = tmp[i];  // Vec<usize>
    if (uniq_out)
        for (uint32_t i = 0; i < h->S; ++i) uniq_out[i] = tmp[(size_t)h->S + i];
    return FLATGFA_OK;
}

int flatgfa_sharded_seg_depth(flatgfa_sharded_t *h, uint64_t *depth_out, uint64_t *uniq_out) {
    if (!h || (!depth_out && h->S)) { set_error("flatgfa_sharded_seg_depth: NULL argument"); return FLATGFA_ERR_ARG; }
    // one critical section from the enqueue to the copy out: another thread's call on the same handle
    // in between would overwrite the reduced vectors (or leave them without unique depth)
    std::lock_guard<std::mutex> lk(h->op_mu);
    int rc = enqueue_locked(*h, uniq_out != nullptr);
    if (!rc) rc = run_all(*h, kCmdSync);
    if (rc) return rc;
    return fetch_locked(h, 0, depth_out, uniq_out);
}

uint64_t flatgfa_sharded_collective_bytes(flatgfa_sharded_t *h, int with_uniq) {
    if (!h) return 0;
    return (uint64_t)vec_count(*h, with_uniq != 0) * 4u;
}

int flatgfa_sharded_path_depth(flatgfa_sharded_t *h, const uint32_t *path_ids, uint32_t n_ids, uint64_t *length_out, double *mean_out) {
    if (!h || (n_ids && (!path_ids || !length_out || !mean_out))) { set_error("flatgfa_sharded_path_depth: NULL argument"); return FLATGFA_ERR_ARG; }
    for (uint32_t k = 0; k < n_ids; ++k)
        if (path_ids[k] >= h->P) { set_error("flatgfa_sharded_path_depth: path id out of range"); return FLATGFA_ERR_BOUNDS; }
    std::lock_guard<std::mutex> lk(h->op_mu);
    // pass 1 over ALL paths (depth.rs:94-99): the reduced node depth on every shard; pass 2: every
    // shard's pieces against it (:116-131), the pieces of a cut path added up on the host
    int rc = enqueue_locked(*h, false);
    if (!rc) rc = run_all(*h, kCmdSync);
    if (!rc) rc = run_all(*h, kCmdPathSums);
    if (rc) return rc;
    std::vector<uint64_t> ln(h->P, 0), ws(h->P, 0);
    DeviceGuard guard;
    for (auto &sp : h->sh) {
        Shard &s = *sp;
        const size_t np = s.pieces.size();
        if (!np) continue;
        std::vector<uint64_t> sums(2 * np);
        if (hipSetDevice(s.device) != hipSuccess || hipMemcpy(sums.data(), s.d_sums, np * 16, hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("flatgfa_sharded_path_depth: device to host copy failed");
            return FLATGFA_ERR_HIP;
        }
        for (size_t i = 0; i < np; ++i) {
            ln[s.pieces[i].path] += sums[i];
            ws[s.pieces[i].path] += sums[np + i];
        }
    }
    for (uint32_t k = 0; k < n_ids; ++k) {
        length_out[k] = ln[path_ids[k]];
        mean_out[k] = (double)ws[path_ids[k]] / (double)ln[path_ids[k]];  // the one floating-point operation on this path: depth.rs:129
    }
    return FLATGFA_OK;
}

}  // extern "C"
