// Plan-time temporaries of the host (depth_fast.hip, depth_device.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

namespace fgfa_dev {

// ---- plan-time temporaries ----
// A plan over a million paths sorts, deals and lists them on the host: a dozen vectors of 4-36 MB each, made and dropped per plan.
// Dropped means unmapped (glibc maps what is larger than its threshold, and trims the heap), and on this driver a process that
// unmaps memory has its GPU queues quiesced and restored by a delayed work item -- the NEXT dispatch then waits 10-30 ms, in steps
// of the kernel's 10 ms tick, now and then half a second (a million paths of a hundred steps: 49 ms to the first answer, 12.5 with
// MALLOC_MMAP_MAX_=0 in the environment: NOTES R6.6c).  So the large temporaries of a plan's creation come from blocks the thread
// keeps: handed out by bumping a pointer, all taken back at once when the next creation starts, given back to the system only
// beyond a quarter of a gigabyte.  Vectors below 4 MB take the ordinary heap.
class TempArena {
  public:
    ~TempArena() { for (Block &b : blocks_) free(b.p); }
    void *take(size_t bytes) {
        bytes = (bytes + 63) & ~(size_t)63;
        for (; cur_ < blocks_.size(); ++cur_) {
            Block &b = blocks_[cur_];
            if (b.cap - b.used >= bytes) {
                void *p = b.p + b.used;
                b.used += bytes;
                return p;
            }
        }
        const size_t cap = std::max<size_t>(bytes, (size_t)64 << 20);
        char *p = (char *)aligned_alloc(64, cap);
        if (!p) throw std::bad_alloc();
        blocks_.push_back(Block{p, cap, bytes});
        cur_ = blocks_.size() - 1;
        return p;
    }
    void enter() {
        if (depth_++ == 0) {
            for (Block &b : blocks_) b.used = 0;
            cur_ = 0;
        }
    }
    void leave() {
        if (--depth_ != 0) return;
        size_t kept = 0, n = 0;
        for (; n < blocks_.size() && kept + blocks_[n].cap <= ((size_t)256 << 20); ++n) kept += blocks_[n].cap;
        for (size_t i = std::max<size_t>(n, 1); i < blocks_.size(); ++i) free(blocks_[i].p);  // (the first block is kept whatever its size)
        blocks_.resize(std::min(blocks_.size(), std::max<size_t>(n, 1)));
    }

  private:
    struct Block {
        char *p;
        size_t cap, used;
    };
    std::vector<Block> blocks_;
    size_t cur_ = 0;
    int depth_ = 0;
};
inline thread_local TempArena t_arena;
struct TempScope {
    TempScope() { t_arena.enter(); }
    ~TempScope() { t_arena.leave(); }
};
template <class T>
struct TempAlloc {
    using value_type = T;
    static constexpr size_t kSmall = (size_t)4 << 20;  // (what a hundred thousand paths need stays on the heap: its blocks are reused warm, and glibc does not map them)
    TempAlloc() = default;
    template <class U>
    TempAlloc(const TempAlloc<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        return static_cast<T *>(bytes < kSmall ? ::operator new(bytes) : t_arena.take(bytes));
    }
    void deallocate(T *p, size_t n) noexcept {
        if (n * sizeof(T) < kSmall) ::operator delete(p);  // (the arena's are taken back all at once)
    }
    template <class U>
    bool operator==(const TempAlloc<U> &) const { return true; }
    template <class U>
    bool operator!=(const TempAlloc<U> &) const { return false; }
};
template <class T>
using Vec = std::vector<T, TempAlloc<T>>;


// A copy between the host's pageable memory and the device, synchronous like hipMemcpy -- through the process's own pinned
// staging buffers from four megabytes up: the runtime would otherwise pin the caller's pages for the transfer, and pages registered with
// the GPU that the kernel then moves or unmaps cost the process's queues the same eviction (depth_device.hip: plan_memcpy).
hipError_t plan_memcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);

}  // namespace fgfa_dev
