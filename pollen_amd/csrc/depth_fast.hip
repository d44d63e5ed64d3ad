// The bucketed node-depth path for gfx950 (seg_depth_with_uniq / seg_depth, ops/depth.rs:15-56):
// no global atomics on the data path.  DESIGN.md section 3 is the long version.
//
//   k_scan        (pass 1, stateless)  persistent workgroups walk one path (or one piece of a long
//                 path) at a time.  The steps are cut into blocks of 1024; a wave takes a block
//                 with four fully coalesced 1 KiB reads (streamed past the L2: nt), so that every
//                 lane holds four groups of four consecutive steps, finds where the maximal +1
//                 runs of segment ids start (-1 runs in an item that mostly walks the ids
//                 downwards: a contig on the reverse strand), and queues (start id, position) per run.
//                 A run's length is the distance to the next queued start, so each run becomes
//                 ONE range record (start, length) instead of `length` histogram updates.  A
//                 record goes to the bucket of its segment window; buckets are split into one
//                 private sub-bucket per workgroup, so the append cursor is an LDS counter and a
//                 workgroup's partial lines stay in its own XCD's L2.  The kernel keeps no
//                 per-path state: after the last wave has left an item it snapshots the cursors,
//                 which tells pass 2 which records of a sub-bucket belong to which path.
//                 A graph beyond 16 M segments is walked once per range of 16 M (k_scan<ranged>
//                 clips every run to the range).
//   k_scan_dense  (pass 1 for graphs with next to no runs)  every step is a record of length one;
//                 the workgroup partitions tiles of 8192 steps by window in LDS, so that a window's
//                 records leave as stretches of consecutive addresses instead of 64 scattered stores.
//   k_scan_short  (pass 1 for paths of at most 2048 steps, and "medium" paths with few runs)
//                 every wave walks whole paths on its own and claims the path's segments in a
//                 per-wave hash set of bitset words; its records carry what they count for.
//   k_accum       (pass 2)  one workgroup per window (several, adding their counts up, when the
//                 graph has fewer windows than CUs).  The "seen" bitset of depth.rs:23-34 lives
//                 here, per (path, window): 512 bytes of LDS instead of one bit per segment of the
//                 whole graph.  A wave walks the records of one path's group after the other (a
//                 path too long for one wave is walked by all sixteen on a shared bitset),
//                 claims each record's segments with returning LDS ORs (the bits that were already
//                 set are revisits), and applies the record as a +1/-1 pair to an LDS difference
//                 array for depth -- and its revisited stretches to a second one; uniq = depth -
//                 revisits -- which are prefix-summed and written with 16-byte stores.  For path
//                 depth the same kernel, once the window's depth is final, turns every record
//                 into two differences of window-local prefix sums (sum len, sum depth * len).
//
// Exactness: every step lies in exactly one run, so it contributes +1 to exactly one depth
// record; every (path, segment) pair that occurs sets exactly one bit of its path's bitset: the
// lane whose OR found it clear met a first visit, every other a revisit.  Sums of +1s are order-independent, hence the
// results equal depth.rs bit for bit under any scheduling.  Sub-buckets have a fixed capacity; a
// record that does not fit raises a flag, and flatgfa_dev_status completes the call on a larger
// plan (or through the atomic kernels) before it reports success.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <queue>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "depth_fast_kernels.hpp"
#include "temp_arena.hpp"
#include "prof.hpp"

namespace fgfa_dev {
namespace {

// Plan time: how many runs (as k_scan_short cuts them: +1 continuations, cut at multiples of 32)
// each path has, and whether it walks the segment ids strictly upwards or strictly downwards from its first
// step to its last (mono[p] = 1: it never meets a segment twice, so a wave-per-path kernel need not look).
// `ext` (or null): per path what k_item_dirs and k_first_ids would say of it as a whole item -- {ascents, descents, steps that follow
// their predecessor upwards (id + 1), downwards (id - 1), the first id, the last id} -- so that a plan none of whose paths is cut
// needs neither of those kernels (one read of the steps instead of two on the way to the first answer).
// A workgroup takes one PIECE of a path at a time (gridDim.y pieces per path: a graph of four chromosome-long paths fills the
// chip like one of a thousand) -- or, where its next four paths are short, each of its waves one of them: 16 bytes a lane and
// four loads in flight, the counts as population counts of wave-wide predicates.  With one piece per path the workgroup writes the path's words; with more, every piece ADDS its counts to words
// the host has cleared (integer adds: any order), piece 0 writes the two ids, and `mono` is the host's to derive from `ext`.
// (the six counts of one wave over steps [lo, hi) of the path [b, e): the same in every lane)
struct RunCounts {
    uint32_t run = 0, down = 0, asc = 0, desc = 0, up = 0, dn = 0;
};
// A wave walks the groups of 256 steps first, first + outer, ... of [lo, hi), four of them (inner apart) at a time: lane l holds steps
// g0 + 4 l .. + 3 of a group, 16 bytes, four loads in flight.  All of the wave's lanes come here together (the loop is wave-uniform).
__device__ __forceinline__ void count_steps(const uint32_t *__restrict__ steps, uint64_t n_steps, uint64_t b, uint64_t lo, uint64_t hi, uint32_t lane,
                                            uint64_t first, uint32_t inner, uint32_t outer, RunCounts &c) {
    // one group: `prev` = the step before the lane's first
    const auto count = [&](uint64_t g0, const uint4 &v, bool interior) {
        const uint32_t s[4] = {v.x, v.y, v.z, v.w};
        uint32_t prev = (uint32_t)__shfl_up((int)v.w, 1, 64);
        if (lane == 0) prev = g0 > b ? steps[g0 - 1] : 0u;
        const uint64_t i0 = g0 + 4u * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t i = i0 + (uint64_t)j;
            const uint32_t id = s[j] >> 1, before = (j ? s[j - 1] : prev) >> 1;
            const bool valid = interior || (i >= lo && i < hi), first_step = !interior && i == b;
            const bool follows_up = id == before + 1u, follows_down = id + 1u == before;
            c.run += (uint32_t)__popcll(__ballot(valid && (first_step || !follows_up || (id & 31u) == 0u)));
            c.down += (uint32_t)__popcll(__ballot(valid && (first_step || !follows_down || (before & 31u) == 0u)));
            c.asc += (uint32_t)__popcll(__ballot(valid && !first_step && id > before));
            c.desc += (uint32_t)__popcll(__ballot(valid && !first_step && id < before));
            c.up += (uint32_t)__popcll(__ballot(valid && !first_step && follows_up));
            c.dn += (uint32_t)__popcll(__ballot(valid && !first_step && follows_down));
        }
    };
    const auto load = [&](uint64_t g0) -> uint4 {
        const uint64_t i0 = g0 + 4u * lane;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i0 < hi) {
            if (i0 + 4u <= n_steps) {
                v = *reinterpret_cast<const uint4 *>(steps + i0);
            } else {  // (the pool's last, partial quad)
                v.x = steps[i0];
                if (i0 + 1u < n_steps) v.y = steps[i0 + 1u];
                if (i0 + 2u < n_steps) v.z = steps[i0 + 2u];
            }
        }
        return v;
    };
    for (uint64_t g = first; g < hi; g += outer) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = g + (uint64_t)inner * k < hi ? load(g + (uint64_t)inner * k) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t g0 = g + (uint64_t)inner * k;
            if (g0 < hi) {
                if (g0 > lo && g0 > b && g0 + 256u <= hi) count(g0, v[k], true);
                else count(g0, v[k], false);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_count_runs(const uint32_t *__restrict__ steps, const uint32_t *__restrict__ pb,
                                                     const uint32_t *__restrict__ pe, uint32_t n_paths, uint64_t n_steps,
                                                     uint32_t *__restrict__ runs, uint32_t *__restrict__ runs_down, uint32_t *__restrict__ mono,
                                                     uint32_t *__restrict__ ext) {
    __shared__ uint32_t tot[6];  // runs, runs read backwards, ascents, descents, +1 steps, -1 steps
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t pieces = gridDim.y, piece = blockIdx.y;
    constexpr uint32_t kWavePath = 2048;  // steps: the workgroup's next four paths that short, and each of its waves takes one by itself
    uint32_t p = blockIdx.x;
    while (p < n_paths) {
        // (the same answer in every thread of the workgroup: the two ways must not mix, one of them has barriers)
        bool each_its_own = pieces == 1;
        for (uint32_t k = 0; k < 4 && each_its_own; ++k) {
            const uint32_t q = p + k * gridDim.x;
            if (q < n_paths) each_its_own = pe[q] - pb[q] <= kWavePath;
        }
        if (each_its_own) {  // a million paths of a hundred steps: no barrier, no LDS, four paths at a time (4.1 -> 0.4 ms)
            const uint32_t q = p + wave * gridDim.x;
            if (q < n_paths) {
                const uint64_t b = pb[q], e = pe[q];
                RunCounts c;
                count_steps(steps, n_steps, b, b, e, lane, b & ~3ull, 256u, 1024u, c);
                if (lane == 0) {
                    runs[q] = c.run;
                    runs_down[q] = c.down;
                    const uint32_t pairs = e > b ? (uint32_t)(e - b - 1) : 0u;
                    mono[q] = (e > b && (c.asc == pairs || c.desc == pairs)) ? 1u : 0u;
                    if (ext) {
                        uint32_t *x = ext + 6 * (size_t)q;
                        x[0] = c.asc;
                        x[1] = c.desc;
                        x[2] = c.up;
                        x[3] = c.dn;
                        x[4] = e > b ? steps[b] >> 1 : 0u;
                        x[5] = e > b ? steps[e - 1] >> 1 : 0u;
                    }
                }
            }
            p += 4 * gridDim.x;
            continue;
        }
        if (threadIdx.x < 6) tot[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t b = pb[p], e = pe[p];
        // this piece: steps [lo, hi) of [b, e), cut at multiples of four steps (16 bytes) of the pool
        uint64_t lo = b, hi = e;
        if (pieces > 1) {
            const uint64_t len = e - b;
            lo = piece ? (b + len * piece / pieces) & ~3ull : b;
            hi = piece + 1 < pieces ? (b + len * (piece + 1) / pieces) & ~3ull : e;
            lo = lo < b ? b : lo;
            hi = hi < lo ? lo : hi;
        }
        // a wave takes every fourth group of 256 steps, four of its groups at a time
        RunCounts c;
        count_steps(steps, n_steps, b, lo, hi, lane, (lo & ~3ull) + 256ull * wave, 1024u, 4096u, c);
        if (lane == 0) {
            atomicAdd(&tot[0], c.run);
            atomicAdd(&tot[1], c.down);
            atomicAdd(&tot[2], c.asc);
            atomicAdd(&tot[3], c.desc);
            atomicAdd(&tot[4], c.up);
            atomicAdd(&tot[5], c.dn);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t *x = ext ? ext + 6 * (size_t)p : nullptr;
            if (pieces == 1) {
                runs[p] = tot[0];
                runs_down[p] = tot[1];
                const uint32_t pairs = e > b ? (uint32_t)(e - b - 1) : 0u;
                mono[p] = (e > b && (tot[2] == pairs || tot[3] == pairs)) ? 1u : 0u;
                if (x) {
                    x[0] = tot[2];
                    x[1] = tot[3];
                    x[2] = tot[4];
                    x[3] = tot[5];
                }
            } else {  // (x != nullptr: the host asks for pieces only together with `ext`)
                atomicAdd(&runs[p], tot[0]);
                atomicAdd(&runs_down[p], tot[1]);
                atomicAdd(&x[0], tot[2]);
                atomicAdd(&x[1], tot[3]);
                atomicAdd(&x[2], tot[4]);
                atomicAdd(&x[3], tot[5]);
            }
            if (x && piece == 0) {
                x[4] = e > b ? steps[b] >> 1 : 0u;
                x[5] = e > b ? steps[e - 1] >> 1 : 0u;
            }
        }
        __syncthreads();
        p += gridDim.x;
    }
}

// Plan time: the steps of the listed paths (x = first step, y = one past the last, z = where the copy
// starts) in reverse order.  One workgroup per path at a time.
__global__ __launch_bounds__(256) void k_reverse_copy(const uint32_t *__restrict__ steps, const uint4 *__restrict__ list, uint32_t n,
                                                      uint32_t *__restrict__ out) {
    for (uint32_t j = blockIdx.x; j < n; j += gridDim.x) {
        const uint4 d = list[j];
        for (uint32_t i = threadIdx.x; i < d.y - d.x; i += 256) out[d.z + i] = steps[d.y - 1u - i];
    }
}

// Plan time: which way each of k_scan's items runs through the segment ids.  Bit 0 of items[j].z = 1 when
// more of its steps follow their predecessor downwards (id - 1) than upwards (id + 1), else 0.
// One workgroup per item at a time.
// the segment a path starts at (plan creation: paths of equal length are dealt out in this order)
__global__ __launch_bounds__(256) void k_first_ids(const uint32_t *__restrict__ steps, const uint32_t *__restrict__ at, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = steps[at[i]] >> 1;
}

// ... and whether it walks them strictly upwards or strictly downwards from its first step to its last (mono[j] =
// {1 up / 2 down / 3 a single step / 0 neither, its first id, its last id, -}): a whole path that does never meets a
// segment twice (items[j].z bit 31 is set right here); the pieces of a split path are put together by the host.
__global__ __launch_bounds__(256) void k_item_dirs(const uint32_t *__restrict__ steps, uint4 *__restrict__ items, uint32_t n_items,
                                                    unsigned long long *__restrict__ n_runs, uint4 *__restrict__ mono) {
    __shared__ uint32_t up, down, asc, desc;
    for (uint32_t j = blockIdx.x; j < n_items; j += gridDim.x) {
        if (threadIdx.x == 0) up = down = asc = desc = 0;
        __syncthreads();
        const uint64_t b = items[j].x, e = items[j].y;
        uint32_t u = 0, d = 0, a = 0, c = 0;
        for (uint64_t i = b + 1 + threadIdx.x; i < e; i += 256) {
            const uint32_t id = steps[i] >> 1, before = steps[i - 1] >> 1;
            u += id == before + 1u ? 1u : 0u;
            d += id + 1u == before ? 1u : 0u;
            a += id > before ? 1u : 0u;
            c += id < before ? 1u : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) {
            u += __shfl_down(u, off, 64);
            d += __shfl_down(d, off, 64);
            a += __shfl_down(a, off, 64);
            c += __shfl_down(c, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&up, u);
            atomicAdd(&down, d);
            atomicAdd(&asc, a);
            atomicAdd(&desc, c);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t pairs = (uint32_t)(e - b - 1);  // (an item has at least one step)
            const uint32_t state = (asc == pairs ? 1u : 0u) | (desc == pairs ? 2u : 0u);
            uint32_t z = (items[j].z & ~1u) | (down > up ? 1u : 0u);
            if (state && (z >> 1) == 0u) z |= kItemNoClaim;  // a whole path (the pieces of a split one: the host)
            items[j].z = z;
            if (mono) mono[j] = make_uint4(state, steps[b] >> 1, steps[e - 1] >> 1, 0u);
            atomicAdd(n_runs, (unsigned long long)(e - b) - max(up, down));  // the records k_scan will make of the item (but for window crossings)
            atomicAdd(n_runs + 1, (unsigned long long)min(asc, desc));         // its steps against the grain (what spoils a window for the per-block no-claim marks)
        }
        __syncthreads();
    }
}


// Plan time: which stretches of an item need no claim although its path is not monotone as a whole.  A path that enters
// a window ONCE and walks it one way -- every step above (or every step below) the one before it while it stays -- meets none
// of the window's segments twice, whatever it does elsewhere; a record in such a window counts for depth and unique
// depth alike (depth.rs:30-34's test is true for each of its steps) and claims nothing.  Two bits per (path, window) in `vis`:
// bit 0 = entered, bit 1 = not so (entered again, a step on the spot, or a change of direction inside).  Order does not
// matter, so all steps are looked at in parallel.  One workgroup per item at a time.
__global__ __launch_bounds__(256) void k_visit_bits(const uint32_t *__restrict__ steps, const uint4 *__restrict__ items, uint32_t n_items,
                                                     const uint32_t *__restrict__ pbeg, uint32_t wb, uint32_t n_win, uint32_t *__restrict__ vis) {
    for (uint32_t j = blockIdx.x; j < n_items; j += gridDim.x) {
        const uint4 it = items[j];
        const uint64_t first = pbeg[it.w];
        for (uint64_t t = (uint64_t)it.x + threadIdx.x; t < it.y; t += 256) {
            const uint32_t id = steps[t] >> 1, w = id >> wb;
            if (w >= n_win) continue;  // (an id beyond the graph: the call reports it)
            const uint64_t idx = (uint64_t)it.w * n_win + w;
            uint32_t *cell = vis + (idx >> 4);
            const uint32_t sh = 2u * (uint32_t)(idx & 15u);
            const uint32_t id1 = t > first ? steps[t - 1] >> 1 : ~0u;
            if (t == first || (id1 >> wb) != w) {
                if (atomicOr(cell, 1u << sh) & (1u << sh)) atomicOr(cell, 2u << sh);
            } else {
                bool bad = id == id1;
                if (t >= first + 2) {
                    const uint32_t id2 = steps[t - 2] >> 1;
                    if ((id2 >> wb) == w) bad = bad || ((id > id1) != (id1 > id2)) || id1 == id2;
                }
                if (bad) atomicOr(cell, 2u << sh);
            }
        }
    }
}
// ... and which 16-step chunks of the items' blocks lie in such windows only: one bit per chunk of the step array (k_scan ANDs a
// block's 64).  A thread per chunk; `n_flagged` counts them.  Spans may overlap (the type allows it): a chunk two paths walk must
// qualify for both, so the launch with `clear` takes the bit away again wherever an item's chunk does not (and counts those).
__global__ __launch_bounds__(256) void k_chunk_flags(const uint32_t *__restrict__ steps, const uint4 *__restrict__ items, uint32_t n_items, bool clear,
                                                      uint32_t wb, uint32_t n_win, const uint32_t *__restrict__ vis, uint32_t *__restrict__ cflags,
                                                      unsigned long long *__restrict__ n_flagged) {
    for (uint32_t j = blockIdx.x; j < n_items; j += gridDim.x) {
        const uint4 it = items[j];
        if (!clear && (it.z >> 31)) continue;  // (an item whose whole path qualifies carries the tag anyway: its chunks neither need marks nor count as a gain)
        const uint64_t t0 = std::min<uint64_t>(((uint64_t)it.x + 15) & ~15ull, it.y);  // (k_scan's make_item: the blocks start here ...)
        const uint64_t n_chunks = ((uint64_t)it.y - t0) / 16;                           // (... and hold whole chunks only)
        uint32_t mine = 0;
        for (uint64_t c = threadIdx.x; c < n_chunks; c += 256) {
            bool ok = true;
            for (uint32_t k = 0; k < 16; ++k) {
                const uint32_t w = (steps[t0 + 16 * c + k] >> 1) >> wb;
                const uint64_t idx = (uint64_t)it.w * n_win + w;
                ok = ok && w < n_win && ((vis[idx >> 4] >> (2u * (uint32_t)(idx & 15u))) & 3u) == 1u;
            }
            const uint64_t bit = t0 / 16 + c;
            if (!clear && ok) {
                atomicOr(&cflags[bit >> 5], 1u << (bit & 31u));
                mine += 1;
            } else if (clear && !ok) {
                if (atomicAnd(&cflags[bit >> 5], ~(1u << (bit & 31u))) & (1u << (bit & 31u))) mine += 1;
            }
        }
        if (mine) atomicAdd(clear ? n_flagged + 1 : n_flagged, (unsigned long long)mine);
    }
}

// ... and which of k_scan's BLOCKS are made of such chunks only: the bit of a block's first chunk in `bflags` (k_scan: one scalar
// load and a bit test per block).  Blocks as make_item cuts them; a thread per block; set and clear launches as above.
__global__ __launch_bounds__(256) void k_block_flags(const uint4 *__restrict__ items, uint32_t n_items, uint64_t n_steps, bool clear,
                                                      const uint32_t *__restrict__ cflags, uint32_t *__restrict__ bflags, unsigned long long *__restrict__ n_flagged) {
    for (uint32_t j = blockIdx.x; j < n_items; j += gridDim.x) {
        const uint4 it = items[j];
        if (!clear && (it.z >> 31)) continue;
        const uint64_t t0 = std::min<uint64_t>(((uint64_t)it.x + 15) & ~15ull, it.y);
        uint64_t chunks = ((uint64_t)it.y - t0) / 16;
        if ((chunks % 64) && t0 + ((chunks + 63) / 64) * 1024 > n_steps) chunks -= chunks % 64;  // (make_item: a last block that would reach past the steps is left to the tail tiles)
        const uint64_t nblk = (chunks + 63) / 64;
        unsigned long long mine = 0;
        for (uint64_t b = threadIdx.x; b < nblk; b += 256) {
            const uint64_t c0 = t0 / 16 + 64 * b;
            const uint32_t n = (uint32_t)std::min<uint64_t>(64, chunks - 64 * b), sh = (uint32_t)(c0 & 31u);
            const uint32_t *cw = cflags + (c0 >> 5);
            const unsigned long long lo = (unsigned long long)cw[0] | ((unsigned long long)cw[1] << 32);
            const unsigned long long v = (lo >> sh) | (sh ? (unsigned long long)cw[2] << (64u - sh) : 0ull);
            const unsigned long long need = n >= 64u ? ~0ull : (1ull << n) - 1ull;
            const bool ok = (v & need) == need;
            if (!clear && ok) {
                atomicOr(&bflags[c0 >> 5], 1u << sh);
                mine += n;
            } else if (clear && !ok) {
                (void)atomicAnd(&bflags[c0 >> 5], ~(1u << sh));
            }
        }
        if (mine) atomicAdd(n_flagged, mine);
    }
}

#define FAST_TRY(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            return false;                                                                   \
        }                                                                                   \
    } while (0)

// The bucket array: capacity per (window, sub-bucket), bounded by the 24-bit slot arithmetic of
// put() and by memory: up to 2^30 records (4 GB) without asking; beyond -- a graph of many windows
// whose paths run along it leaves most sub-buckets empty and needs the few others deep -- up to a
// quarter of the device memory that is free, 64 GB at most (FLATGFA_BUCKET_GB; the diagnostic build
// of k_scan keeps the 32-bit offsets).
// Returns 1 when it is allocated, 0 when the capacity would be too small to be useful, -1 on a HIP error.
// The bucket array of the last plan that went, kept per device for the next one that comes (flatgfa_dev_release_scratch gives it
// back): allocating and freeing gigabytes of device memory in a row is what the driver does worst -- a plan made and dropped
// beside five that stay took 5.4 ms or, every other time, 80-300 ms (tools/flow_probe2.py: all 4.8 ms with the array kept) --
// and plans are made again in earnest: flatgfa_dev_plan_steps_changed, a pipeline's lanes, a handle per request.
struct BucketCache {
    uint32_t *p = nullptr;
    uint64_t bytes = 0;
};
static BucketCache g_bucket_cache[64];
static std::mutex g_bucket_cache_mu;
static uint32_t *bucket_cache_take(uint64_t bytes) {
    std::lock_guard<std::mutex> lk(g_bucket_cache_mu);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
    BucketCache &c = g_bucket_cache[dev];
    if (c.p && c.bytes >= bytes && c.bytes <= 2 * bytes) {  // (an array up to twice the size asked for will do: the capacity is a floor)
        uint32_t *p = c.p;
        c = BucketCache();
        return p;
    }
    return nullptr;
}
static void bucket_cache_give(uint32_t *p, uint64_t bytes) {
    uint32_t *drop = p;
    {
        std::lock_guard<std::mutex> lk(g_bucket_cache_mu);
        int dev = -1;
        hipPointerAttribute_t at;  // (the array's own device: the calling thread's current one may be another)
        if (hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device;
        if (dev >= 0 && dev < 64 && bytes <= (8ull << 30) && bytes >= g_bucket_cache[dev].bytes) {
            drop = g_bucket_cache[dev].p;  // (the larger of the two stays)
            g_bucket_cache[dev] = BucketCache{p, bytes};
        } else {
            (void)hipGetLastError();
        }
    }
    if (drop) (void)hipFree(drop);
}
}  // namespace
void fast_release_scratch() {
    std::lock_guard<std::mutex> lk(g_bucket_cache_mu);
    int cur = -1;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (int dev = 0; dev < 64; ++dev) {
        if (!g_bucket_cache[dev].p) continue;
        if (hipSetDevice(dev) == hipSuccess) (void)hipFree(g_bucket_cache[dev].p);
        g_bucket_cache[dev] = BucketCache();
    }
    if (have) (void)hipSetDevice(cur);
    (void)hipGetLastError();
}
namespace {
int alloc_buckets(FastPlan *fp, uint64_t want_cap) {
    const uint64_t slots = (uint64_t)fp->n_win * fp->n_slots;
    uint64_t max_cap = ((1ull << 30) - 1) / ((uint64_t)(fp->n_win + 1) * fp->n_slots);
    if (want_cap > max_cap && !fp->dbg && fp->tagged && !fp->n_short && !fp->n_medium && !fp->n_tiny) {  // (k_scan's tagged builds only)
        size_t free_b = 0, total_b = 0;
        uint64_t budget = 64ull << 30;
        if (const char *e = getenv("FLATGFA_BUCKET_GB")) budget = strtoull(e, nullptr, 10) << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<uint64_t>(budget, (uint64_t)free_b / 4 + (fp->buckets ? (slots + fp->n_slots) * (uint64_t)fp->cap * 4 : 0));
        else (void)hipGetLastError();
        max_cap = std::max<uint64_t>(max_cap, budget / 4 / ((uint64_t)(fp->n_win + 1) * fp->n_slots));
    }
    uint64_t cap = std::min(want_cap, max_cap);
    cap = std::min<uint64_t>(cap, ((1ull << 24) - 1) / fp->n_slots);  // window * (n_slots * cap) + pos is a 24-bit multiply
    cap &= cap >= 64 ? ~31ull : ~3ull;  // sub-buckets start on 128-byte lines: neighbours (other workgroups, other XCDs) never share one
    if (cap < 4) return 0;
    if (fp->buckets && cap <= fp->cap) return 1;  // (at its limit: the array stays as it is)
    uint32_t *fresh = bucket_cache_take((slots + fp->n_slots) * cap * 4);
    const hipError_t e = fresh ? hipSuccess : hipMalloc(&fresh, (slots + fp->n_slots) * cap * 4);  // (the new one first: a plan that cannot grow keeps what it has)
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error(std::string("hipMalloc(buckets): ") + hipGetErrorString(e));
        return fp->buckets ? 1 : -1;
    }
    if (fp->buckets) bucket_cache_give(fp->buckets, (slots + fp->n_slots) * (uint64_t)fp->cap * 4);
    fp->buckets = fresh;
    fp->cap = (uint32_t)cap;
    return 1;
}

}  // namespace

static int run_range(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                     uint32_t *status, hipStream_t stream, const PathSums *ps, bool count_only);
static thread_local uint32_t t_scan_workgroups = 0;  // fast_plan_create's `scan_workgroups`, for the ranges it makes
static thread_local bool t_prefer_packed = false;    // ... and its `prefer_packed`

// The A/B switches of past measurements (which policy a plan takes: NOTES.md) read the environment in measurement
// builds only (-DFGFA_MEASURE, tools/variants.sh); the product library takes the policy that won.
static inline bool measure_switch(const char *name) {
#ifdef FGFA_MEASURE
    return getenv(name) != nullptr;
#else
    (void)name;
    return false;
#endif
}

// The plan of one range of segments, [seg_base, seg_base + n_range): the whole graph, or one of
// the ranges of a graph beyond 16 M segments.
static bool create_range(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, FastPlan *fp, uint32_t seg_base,
                         uint32_t n_range, uint32_t max_win, uint32_t force_wb, uint32_t siblings = 1) {
    *fp = FastPlan();
    fp->seg_base = seg_base;
    fp->n_range = n_range;
    const bool ranged = seg_base != 0 || n_range != g.n_segs;
    // Windows of 4096 segments up to 4 M segments, of 8192 beyond (pass 2 keeps a window's
    // difference array and per-path bitsets in LDS).
    uint32_t wb = g.n_segs <= 1024u * 4096u ? 12u : 13u;
    if (force_wb) wb = force_wb;  // (fast_plan_create: 4096-segment windows on a larger graph, for the sake of its split paths)
    if (const char *f = test_hook("FLATGFA_WB")) wb = (uint32_t)strtoul(f, nullptr, 10);
    const uint32_t n_win = (uint32_t)(((uint64_t)n_range + (1u << wb) - 1) >> wb);
    if (n_win > max_win) return true;
    int dev = 0;
    FAST_TRY(hipGetDevice(&dev));
    fp->n_cus = (uint32_t)device_cu_count(dev);
    fp->n_slots = fp->n_cus;
    // Pass 1's persistent workgroups: one per CU -- or fewer for a plan that is one lane of a pipeline (calls in flight:
    // flatgfa_dev_pipeline_create), where the CUs it leaves are another call's.  FLATGFA_SCAN_WGS=n: tests, measurements.
    if (t_scan_workgroups) fp->n_slots = std::max(8u, std::min<uint32_t>(fp->n_cus, t_scan_workgroups));
    if (const char *f = test_hook("FLATGFA_SCAN_WGS")) fp->n_slots = std::max(8u, std::min<uint32_t>(fp->n_cus, (uint32_t)strtoul(f, nullptr, 10)));
    if (fp->n_slots > kMaxSlots) return true;
    fp->n_win = n_win;
    fp->wb = wb;
    fp->nwp = (n_win + 1u + 63u) & ~63u;  // (one entry more than windows: a packed plan's offset table ends with the region's end)
    fp->lds_bytes_scan = scan_lds_bytes(fp->nwp, n_win > kMaxWin);
    if (fp->lds_bytes_scan + 64 > kLdsLimit) return true;
#ifdef FGFA_MEASURE  // ablations that make the results wrong by construction exist in measurement builds only (tools/variants.sh)
    if (const char *d = test_hook("FLATGFA_DEBUG_SKIP")) fp->dbg = (uint32_t)strtoul(d, nullptr, 10);
#endif
    if (fp->dbg && ranged) return true;  // the diagnostic build of k_scan has no registers left for ranges
    // Paths of at most `short_max` steps are walked by single waves (k_scan_short), unless their
    // last block would reach beyond the step array.  Those kernels address at most 256 windows
    // of 4096 segments.
    uint64_t short_max = (fp->dbg || ranged || wb != kShortWinBits || n_win > kShortMaxWin || g.n_segs > kShortMaxSegs) ? 0 : kShortMax;  // (the wave-per-path kernels know nothing of ranges)
    if (const char *forced = test_hook("FLATGFA_SHORT_MAX")) short_max = std::min<uint64_t>(short_max, strtoull(forced, nullptr, 10));
    // Which kernel walks a path depends on how many runs it has: short paths must fit the run queue,
    // paths with at most kMediumRuns runs are walked wave by wave too, by pairs of waves that share a
    // bigger hash set (k_scan_short's medium variant).  The counts come from a one-off kernel.
    Vec<uint32_t> runs, runs_down, mono, ext;  // (ext: six facts per path, see k_count_runs; for up to 2^18 paths)
    if (short_max) {
        // (counted over the spans this plan walks -- a plan may be given others than the graph's own)
        uint32_t *d_runs = nullptr;
        const size_t np = g.n_paths;
        const bool want_ext = np <= (1u << 18) && !measure_switch("FLATGFA_NO_ITEM_DIRS");
        // (device layout: [runs | runs_down | mono | ext x 6 | begin | end], np words each: one copy in, one copy out)
        const size_t n_out = want_ext ? 9 : 3;
        FAST_TRY(hipMalloc(&d_runs, np * (n_out + 2) * 4));
        Vec<uint32_t> spans(2 * np);
        std::copy(hb, hb + np, spans.begin());
        std::copy(he, he + np, spans.begin() + (ptrdiff_t)np);
        hipError_t e = plan_memcpy(d_runs + n_out * np, spans.data(), np * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d_runs);
            FAST_TRY(e);
        }
        // (few long paths: several pieces each, so that the kernel fills the chip -- the pieces add to cleared words)
        uint32_t pieces = 1;
        if (want_ext && np < fp->n_cus * 8u && g.n_steps / np >= (1u << 16)) pieces = std::min<uint32_t>((fp->n_cus * 8u + (uint32_t)np - 1u) / (uint32_t)np, 4096u);
        if (const char *forced = test_hook("FLATGFA_COUNT_PIECES")) pieces = want_ext ? (uint32_t)std::min(std::max(1, atoi(forced)), 4096) : 1u;
        if (pieces > 1) {
            e = hipMemsetAsync(d_runs, 0, n_out * np * 4, nullptr);
            if (e != hipSuccess) {
                (void)hipFree(d_runs);
                FAST_TRY(e);
            }
        }
        hipLaunchKernelGGL(k_count_runs, dim3(std::min<uint32_t>(g.n_paths, fp->n_cus * 8u), pieces), dim3(256), 0, nullptr, g.steps,
                           d_runs + n_out * np, d_runs + (n_out + 1) * np, g.n_paths, g.n_steps, d_runs, d_runs + np, d_runs + 2 * np,
                           want_ext ? d_runs + 3 * np : nullptr);
        Vec<uint32_t> out(n_out * np);
        e = plan_memcpy(out.data(), d_runs, n_out * np * 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_runs);
        FAST_TRY(e);
        runs.assign(out.begin(), out.begin() + (ptrdiff_t)np);
        runs_down.assign(out.begin() + (ptrdiff_t)np, out.begin() + (ptrdiff_t)(2 * np));
        mono.assign(out.begin() + (ptrdiff_t)(2 * np), out.begin() + (ptrdiff_t)(3 * np));
        if (want_ext) ext.assign(out.begin() + (ptrdiff_t)(3 * np), out.end());
        if (pieces > 1) {  // (no one piece knows: a path is monotone when all of its pairs ascend or all of them descend)
            for (size_t p = 0; p < np; ++p) {
                const uint32_t pairs = he[p] > hb[p] ? he[p] - hb[p] - 1u : 0u;
                mono[p] = (he[p] > hb[p] && (ext[6 * p] == pairs || ext[6 * p + 1] == pairs)) ? 1u : 0u;
            }
        }
        // (FLATGFA_NO_CLAIM=0: every path claims, monotone or not -- tests and measurements)
        if (const char *nc = test_hook("FLATGFA_NO_CLAIM"); nc && nc[0] == '0') std::fill(mono.begin(), mono.end(), 0u);
    }
    plan_tick("range: k_count_runs + its copies");
    const bool short_any = test_hook("FLATGFA_SHORT_ANY") != nullptr;  // tests: let k_scan_short find out and hand back
    const bool no_rev = measure_switch("FLATGFA_NO_REVERSED_COPIES");
    // A wave-per-path kernel only knows runs that go up.  A path that walks the ids downwards (a
    // contig on the reverse strand) has far fewer runs when it is read backwards, and the order of a
    // path's steps does not matter to the counts: such a path is walked from a reversed copy of its
    // steps, made here once (rev_steps; every copy starts at a multiple of 16).
    // (the *_mono lists: paths that walk the ids strictly one way -- the wave-per-path kernels skip their claims)
    Vec<uint4> items, short_items, medium_items, short_rev, medium_rev, whole, rev_list, tiny_items;
    Vec<uint4> short_mono, medium_mono, short_rev_mono, medium_rev_mono, tiny_mono;
    const bool no_tiny = test_hook("FLATGFA_NO_TINY") != nullptr;  // (measurements, tests: tiny paths go to k_scan_short as before)
    uint64_t rev_len = 0;
    for (uint32_t p = 0; p < g.n_paths; ++p) {
        const uint64_t b = hb[p], e = he[p], n = e - b;
        if (n == 0) continue;
        const bool in_reach = ((e + 15) & ~15ull) <= g.n_steps;  // the last block must not read past the step array
        const bool down = short_max && !no_rev && runs_down[p] < runs[p] && rev_len + n + 2048 < 0xFFFFFFFFull;
        const uint32_t rn = short_max ? (down ? runs_down[p] : runs[p]) : 0u;
        const bool is_tiny = n <= std::min<uint64_t>(short_max, kTinyMax) && !no_tiny;  // (k_scan_tiny: a wave holds the whole path)
        const bool is_short = !is_tiny && n <= short_max && (down || in_reach) && (rn + 16 <= kQCap || short_any);
        const bool is_medium = !is_tiny && !is_short && short_max && (down || in_reach) && rn <= kMediumRuns;
        const bool mn = short_max && mono[p] != 0u;
        if (is_tiny) {
            (mn ? tiny_mono : tiny_items).push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else if ((is_short || is_medium) && down) {
            const uint32_t at = (uint32_t)rev_len;
            rev_list.push_back(make_uint4((uint32_t)b, (uint32_t)e, at, p));
            (is_short ? (mn ? short_rev_mono : short_rev) : (mn ? medium_rev_mono : medium_rev)).push_back(make_uint4(at, at + (uint32_t)n, kNoSlot, p));
            rev_len += (n + 15) & ~15ull;
        } else if (is_short) {
            (mn ? short_mono : short_items).push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else if (is_medium) {
            (mn ? medium_mono : medium_items).push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        } else {
            whole.push_back(make_uint4((uint32_t)b, (uint32_t)e, kNoSlot, p));
        }
    }
    // Paths of equal length -- the ties of the sort below -- are walked in the order of where they
    // start: k_scan's workgroups take the items one after the other, so neighbours in the list go to
    // different workgroups, and it is the paths that start near each other that meet in a window.  A
    // sub-bucket then holds one path's records of its window, not those of the five or six that
    // chance gave one workgroup (the capacity every sub-bucket gets is the fullest one's, §2).
    if (whole.size() > 1 && !measure_switch("FLATGFA_KEEP_PATH_ORDER")) {
        Vec<uint32_t> at(whole.size()), first(whole.size(), 0u);
        if (!ext.empty()) {  // (the counting kernel looked already)
            for (size_t i = 0; i < whole.size(); ++i) first[i] = ext[6 * (size_t)whole[i].w + 4];
        } else {
        for (size_t i = 0; i < whole.size(); ++i) at[i] = whole[i].x;
        uint32_t *d_at = nullptr;
        FAST_TRY(hipMalloc(&d_at, whole.size() * 8));
        hipError_t e = plan_memcpy(d_at, at.data(), whole.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_first_ids, dim3((uint32_t)((whole.size() + 255) / 256)), dim3(256), 0, nullptr, g.steps, d_at, (uint32_t)whole.size(),
                               d_at + whole.size());
            e = plan_memcpy(first.data(), d_at + whole.size(), whole.size() * 4, hipMemcpyDeviceToHost);
        }
        (void)hipFree(d_at);
        FAST_TRY(e);
        }
        Vec<uint32_t> order(whole.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });
        Vec<uint4> sorted(whole.size());
        for (size_t i = 0; i < whole.size(); ++i) sorted[i] = whole[order[i]];
        whole.swap(sorted);
    }
    plan_tick("range: paths classified, k_first_ids, sorted");
    if (!rev_list.empty()) {
        fp->n_rev_steps = (uint32_t)(rev_len + 1024);  // (a block is read whole)
        FAST_TRY(hipMalloc(&fp->rev_steps, (size_t)fp->n_rev_steps * 4));
        FAST_TRY(hipMemset(fp->rev_steps, 0, (size_t)fp->n_rev_steps * 4));
        uint4 *d_list = nullptr;
        FAST_TRY(hipMalloc(&d_list, rev_list.size() * sizeof(uint4)));
        hipError_t e = plan_memcpy(d_list, rev_list.data(), rev_list.size() * sizeof(uint4), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_reverse_copy, dim3(std::min<uint32_t>((uint32_t)rev_list.size(), fp->n_cus * 8u)), dim3(256), 0, nullptr,
                               g.steps, d_list, (uint32_t)rev_list.size(), fp->rev_steps);
            e = hipDeviceSynchronize();
        }
        (void)hipFree(d_list);
        FAST_TRY(e);
    }
    // k_scan's work items: whole paths, except that a path longer than `piece` steps is cut into
    // equal pieces, so that graphs with few long paths still fill the chip.  The workgroups take the
    // items, longest first, in a fixed snake order (item_of): with a hundred paths of a million steps
    // and pieces of N / (2 CUs), three pieces fall to some workgroups and two to most (+28 % on the
    // longest).  So the deal is played through on the host for a few piece sizes, every item charged
    // a few blocks' worth for its turnaround, and the size with the shortest longest hand is taken
    // (the largest such size: for 1000 paths of 100 k steps, no cutting at all).
    // (z: bit 0 = the item walks the ids downwards, set by k_item_dirs; from bit 1 up, 1 + the
    // ordinal of the split path the item is a piece of, or 0 for a whole path)
    const auto cut = [&](uint64_t piece, Vec<uint4> *out) -> uint32_t {
        uint32_t n_split = 0;
        for (const uint4 &w : whole) {
            const uint64_t b = w.x, n = (uint64_t)w.y - w.x;
            const uint32_t k = (uint32_t)((n + piece - 1) / piece);
            const uint32_t z = k > 1 ? (++n_split) << 1 : 0u;
            for (uint32_t j = 0; j < k; ++j)
                out->push_back(make_uint4((uint32_t)(b + n * j / k), (uint32_t)(b + n * (j + 1) / k), z, w.w));
        }
        return n_split;
    };
    const auto longer = [](const uint4 &a, const uint4 &b) { return a.y - a.x > b.y - b.x; };
    uint64_t piece = 0;
    if (const char *forced = test_hook("FLATGFA_PIECE_STEPS")) {
        piece = (std::max<uint64_t>(256, strtoull(forced, nullptr, 10)) + 255) & ~255ull;
    } else {
        constexpr uint64_t kTurnaround = 8192;  // steps a workgroup could have walked while it changes items
        uint64_t long_steps = 0;
        for (const uint4 &w : whole) long_steps += w.y - w.x;
        uint64_t best = ~0ull;
        for (const uint32_t twice_m : {4u, 5u, 6u, 8u, 10u, 12u, 16u, 20u, 24u}) {  // pieces of N / (m CUs), m = 2 .. 12
            uint64_t cand = std::max<uint64_t>(32768, (2 * long_steps + (uint64_t)twice_m * fp->n_slots - 1) / ((uint64_t)twice_m * fp->n_slots));
            cand = (cand + 255) & ~255ull;
            Vec<uint4> trial;
            cut(cand, &trial);
            std::stable_sort(trial.begin(), trial.end(), longer);
            Vec<uint64_t> hand(fp->n_slots, 0);
            for (size_t i = 0; i < trial.size(); ++i) {
                const size_t round = i / fp->n_slots, pos = i % fp->n_slots;
                hand[(round & 1) ? fp->n_slots - 1 - pos : pos] += (uint64_t)(trial[i].y - trial[i].x) + kTurnaround;
            }
            const uint64_t longest = *std::max_element(hand.begin(), hand.end());
            if (longest + longest / 64 < best) {  // smaller pieces have to win by more than 1.5 %
                best = longest;
                piece = cand;
            }
            if (cand == 32768) break;
        }
        // Cutting has a price beyond the turnaround where pass 2 cannot give all split paths a bitset
        // of their own kind -- none at all with 8192-segment windows, 128 with smaller ones: the plan
        // then takes smaller windows, more ranges, or walks the paths in groups (fast_plan_create).  A
        // graph of thousands of paths rarely needs its long ones cut: k_scan's workgroups take the
        // items longest first as they get to them, and if whole paths dealt that way (to the least
        // loaded workgroup each) leave the longest hand within a tenth of the best cut's, they stay whole.
        if (piece && !measure_switch("FLATGFA_KEEP_CUTS")) {
            Vec<uint4> trial;
            const uint32_t n_split = cut(piece, &trial);
            if (n_split > (wb <= 12 ? kMaxShared : 0u)) {
                Vec<uint64_t> lens;
                for (const uint4 &w : whole) lens.push_back((uint64_t)w.y - w.x);
                std::sort(lens.begin(), lens.end(), std::greater<uint64_t>());
                std::priority_queue<uint64_t, Vec<uint64_t>, std::greater<uint64_t>> hands;
                for (uint32_t i = 0; i < fp->n_slots; ++i) hands.push(0);
                uint64_t longest = 0;
                for (const uint64_t n : lens) {
                    const uint64_t h = hands.top() + n + kTurnaround;
                    hands.pop();
                    hands.push(h);
                    longest = std::max(longest, h);
                }
                if (longest <= best + best / 10) piece = ~0ull >> 1;  // (longer than any path)
            }
        }
    }
    fp->n_shared = cut(piece ? piece : 32768, &items);
    std::stable_sort(items.begin(), items.end(), longer);
    plan_tick("range: the deal played through, items cut");
    // A wave-per-path list: the paths read from the graph's steps, then those read from their reversed copies; of
    // either kind the ones that need no claim lie next to the boundary, so that one stretch of the list names them
    // all: [claim][no claim | no claim, reversed][claim, reversed], each part longest first.
    const auto lay_out = [&](Vec<uint4> *fwd, Vec<uint4> *fwd_mono, Vec<uint4> *rev_mono, Vec<uint4> *rev,
                             uint32_t *n_rev, uint32_t *mono_lo, uint32_t *mono_n) {
        for (Vec<uint4> *v : {fwd, fwd_mono, rev_mono, rev}) std::stable_sort(v->begin(), v->end(), longer);
        *mono_lo = (uint32_t)fwd->size();
        *mono_n = (uint32_t)(fwd_mono->size() + rev_mono->size());
        *n_rev = (uint32_t)(rev_mono->size() + rev->size());
        for (Vec<uint4> *v : {fwd_mono, rev_mono, rev}) fwd->insert(fwd->end(), v->begin(), v->end());
    };
    lay_out(&short_items, &short_mono, &short_rev_mono, &short_rev, &fp->n_short_rev, &fp->short_mono_lo, &fp->short_mono_n);
    lay_out(&medium_items, &medium_mono, &medium_rev_mono, &medium_rev, &fp->n_medium_rev, &fp->medium_mono_lo, &fp->medium_mono_n);
    fp->tiny_mono_lo = (uint32_t)tiny_items.size();
    fp->tiny_mono_n = (uint32_t)tiny_mono.size();
    tiny_items.insert(tiny_items.end(), tiny_mono.begin(), tiny_mono.end());
    fp->n_items = (uint32_t)items.size();
    fp->n_short = (uint32_t)short_items.size();
    fp->n_medium = (uint32_t)medium_items.size();
    fp->n_tiny = (uint32_t)tiny_items.size();
    {
        const auto steps_of = [](const Vec<uint4> &v) {
            uint64_t n = 0;
            for (const uint4 &d : v) n += d.y - d.x;
            return n;
        };
        fp->class_steps[0] = steps_of(items);
        fp->class_steps[1] = steps_of(short_items);
        fp->class_steps[2] = steps_of(medium_items);
        fp->class_steps[3] = steps_of(tiny_items);
    }
    if (items.empty() && short_items.empty() && medium_items.empty() && tiny_items.empty()) return true;
    fp->max_back = std::min<uint32_t>(fp->n_short, kMaxHandBack);
    fp->exact_short = !short_any;  // the run counts the lists were made from are exact: nothing is handed back
    fp->dstride = fp->n_items + fp->max_back + 1;
    // The directory (one cursor pair per item and window) must stay small next to the steps.
    if ((uint64_t)fp->dstride * n_win * 8 > std::max<uint64_t>(64ull << 20, g.n_steps * 2)) {
        *fp = FastPlan();
        return true;
    }
    // Pass 2 runs one workgroup per window -- or, when the graph has fewer windows than half the
    // CUs, several that share the window's paths and sub-buckets and add their counts up (1000
    // paths over 100 k segments: 25 workgroups took 0.48 ms where 250 take 0.06).
    fp->acc_parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({16, fp->n_cus / n_win, (g.n_steps / n_win + (32u << 10) - 1) >> 15}));  // a workgroup per 32 k steps: a wave's walk is a chain of dependent round trips, a microsecond per 64 records
    if (const char *f = test_hook("FLATGFA_ACC_PARTS")) fp->acc_parts = std::max(1u, std::min(64u, (uint32_t)strtoul(f, nullptr, 10)));
    const uint32_t acc_waves = fp->acc_parts * kAccWaves;
    // Tagged calls (records say whose they are; see kTagShift): every workgroup's items -- the handed-back
    // ones included -- must have tags of their own next to the split paths', and pass 2 needs LDS for a
    // bitset per split path (none to spare with 8192-segment windows; a window shared by several
    // workgroups cannot share bitsets).  FLATGFA_TAGGED=0 keeps the directory (tests, measurements).
    {
        const uint32_t grid = (fp->n_short || fp->n_medium || fp->n_tiny) ? fp->n_slots : std::min<uint32_t>(fp->n_items, fp->n_slots);
        const uint64_t per_wg = grid ? ((uint64_t)fp->n_items + fp->max_back + grid - 1) / grid : 0;
        const char *t = test_hook("FLATGFA_TAGGED");
        const uint32_t shared_cap = wb <= 12 ? kMaxShared : 0u;
        const bool base_ok = !fp->dbg && !(t && t[0] == '0') && (fp->n_shared == 0 || fp->acc_parts == 1);
        const bool taggable = base_ok && fp->n_shared <= shared_cap;
        // A workgroup's private tags: what the split paths leave of the 512 (FLATGFA_TAG_LIMIT: fewer, tests).
        // k_scan deals the items out as its workgroups get to them, so it is not the mean that has to
        // fit but the most any workgroup takes: the deal is played through here (its first two items
        // are fixed, every further one goes to whoever is done first), with some room to spare --
        // a workgroup that does run out of tags stops taking items (k_scan) and the others go on.
        uint32_t limit = fp->n_shared + 1u < kTagCount ? kTagCount - 1u - fp->n_shared : 0u;  // (the highest tag says "no claim")
        if (const char *f = test_hook("FLATGFA_TAG_LIMIT")) limit = std::min<uint32_t>(limit, std::max(2u, (uint32_t)strtoul(f, nullptr, 10)));
        fp->tag_limit = limit;
        uint64_t most = per_wg;
        if (taggable && grid && per_wg <= limit && fp->n_items + fp->max_back > 2ull * grid && !test_hook("FLATGFA_TAG_MEAN_ONLY")) {
            constexpr uint64_t kTurn = 2048;  // steps' worth an item costs beyond its steps
            std::priority_queue<std::pair<uint64_t, uint32_t>, Vec<std::pair<uint64_t, uint32_t>>, std::greater<std::pair<uint64_t, uint32_t>>> hands;
            Vec<uint32_t> taken(grid, 0u);
            for (uint32_t i = 0; i < grid; ++i) {
                uint64_t h = 0;
                for (uint32_t k = 0; k < 2; ++k)
                    if (i + k * grid < fp->n_items) h += (uint64_t)(items[i + k * grid].y - items[i + k * grid].x) + kTurn, taken[i] += 1;
                hands.push({h, i});
            }
            for (uint64_t j = 2ull * grid; j < (uint64_t)fp->n_items + fp->max_back; ++j) {
                const uint64_t n = j < fp->n_items ? (uint64_t)(items[j].y - items[j].x) : kShortMax;
                auto [h, i] = hands.top();
                hands.pop();
                taken[i] += 1;
                hands.push({h + n + kTurn, i});
            }
            most = *std::max_element(taken.begin(), taken.end());
            most += most / 4;  // (the workgroups do not run at one speed)
        }
        fp->tagged = taggable && per_wg <= limit && most <= limit;
        // what fast_plan_create may do about a plan that is not: walk the paths in groups (fewer items, fewer
        // split paths per group), or take 4096-segment windows (pass 2 then has LDS for split paths' bitsets)
        fp->too_many_items = base_ok && !fp->tagged && fp->acc_parts == 1 && !fp->n_short && !fp->n_medium && !fp->n_tiny && (taggable || (wb <= 12 && fp->n_shared > shared_cap));
        fp->want_wb12 = base_ok && !fp->tagged && wb == 13 && fp->n_shared > 0 && fp->acc_parts == 1;
        if (!fp->tagged && n_win > kMaxWin) {  // so many windows only without cursor snapshots: the caller cuts smaller ranges
            const bool many = fp->too_many_items, w12 = fp->want_wb12;
            fast_plan_destroy(fp);
            fp->too_many_items = many;
            fp->want_wb12 = w12;
            return true;
        }
    }
    // Pass 2 walks k_scan's items grouped by path (the pieces of a split path share a bitset),
    // each of its waves a contiguous stretch of the list: paths are dealt to the waves longest
    // first, each to the wave with the least steps so far.
    {
        Vec<Vec<uint32_t>> by_path;  // item indices per path that has items
        Vec<uint64_t> path_steps;
        Vec<int64_t> slot_of(g.n_paths, -1);
        for (uint32_t j = 0; j < fp->n_items; ++j) {
            const uint32_t p = items[j].w;
            if (slot_of[p] < 0) {
                slot_of[p] = (int64_t)by_path.size();
                by_path.emplace_back();
                path_steps.push_back(0);
            }
            by_path[(size_t)slot_of[p]].push_back(j);
            path_steps[(size_t)slot_of[p]] += items[j].y - items[j].x;
        }
        Vec<uint32_t> order(by_path.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return path_steps[a] > path_steps[b]; });
        // A path with more than half a wave's even share of the steps would hold its wave up (four
        // paths of 25 M steps: four waves busy out of sixteen, pass 2 2.6 times slower).  Such
        // paths go to the window's workgroups whole -- to the one with the least so far -- and
        // their pieces to its sixteen waves in turn; the others are dealt to single waves as before.
        uint64_t total_steps = 0;
        for (uint64_t v : path_steps) total_steps += v;
        const uint64_t fat_min = total_steps / (2ull * acc_waves) + 1;
        Vec<Vec<uint32_t>> per_wave(acc_waves), fat_of_part(fp->acc_parts);
        Vec<uint64_t> load(acc_waves, 0), part_load(fp->acc_parts, 0);
        for (uint32_t gi : order) {
            if (path_steps[gi] < fat_min || by_path[gi].size() < kAccWaves / 2 || measure_switch("FLATGFA_NO_FAT_PATHS")) continue;  // (fewer pieces than half the waves: better one wave busy all the time than three)
            const uint32_t q = (uint32_t)(std::min_element(part_load.begin(), part_load.end()) - part_load.begin());
            part_load[q] += path_steps[gi];
            fat_of_part[q].push_back(gi);
        }
        for (uint32_t q = 0; q < fp->acc_parts; ++q)
            for (uint32_t wv = 0; wv < kAccWaves; ++wv) load[q * kAccWaves + wv] = part_load[q] / kAccWaves;
        std::vector<bool> is_fat(by_path.size(), false);
        for (const auto &v : fat_of_part)
            for (uint32_t gi : v) is_fat[gi] = true;
        for (uint32_t gi : order) {
            if (is_fat[gi]) continue;
            const uint32_t wv = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
            load[wv] += path_steps[gi] + 64;
            bool first = true;
            for (uint32_t j : by_path[gi]) {
                per_wave[wv].push_back(j | (first ? 0x80000000u : 0u));
                first = false;
            }
        }
        Vec<uint32_t> elist, wave_off(acc_waves + 1, 0);
        for (uint32_t wv = 0; wv < acc_waves; ++wv) {
            wave_off[wv] = (uint32_t)elist.size();
            elist.insert(elist.end(), per_wave[wv].begin(), per_wave[wv].end());
        }
        wave_off[acc_waves] = (uint32_t)elist.size();
        Vec<uint32_t> fat_off(fp->acc_parts + 1, 0), fat_woff;
        for (uint32_t q = 0; q < fp->acc_parts; ++q) {
            fat_off[q] = fp->n_fat;
            for (uint32_t gi : fat_of_part[q]) {
                const Vec<uint32_t> &its = by_path[gi];
                for (uint32_t wv = 0; wv < kAccWaves; ++wv) {
                    fat_woff.push_back((uint32_t)elist.size());
                    bool first = true;
                    for (size_t k = wv; k < its.size(); k += kAccWaves) {
                        elist.push_back(its[k] | (first ? 0x80000000u : 0u));
                        first = false;
                    }
                }
                fat_woff.push_back((uint32_t)elist.size());
                fp->n_fat += 1;
            }
        }
        fat_off[fp->acc_parts] = fp->n_fat;
        // k_scan leaves an item's cursors and sub-bucket at the item's place in this order
        Vec<uint32_t> perm(fp->n_items + 1, 0);
        for (size_t at = 0; at < elist.size(); ++at) perm[elist[at] & 0x7FFFFFFFu] = (uint32_t)at | (elist[at] & 0x80000000u);
        // (the five lists in one allocation and one copy, every one on a 256-byte boundary: an allocation and a copy each cost a
        // tenth of what the first answer costs)
        const auto padded = [](size_t words) { return (words + 63) & ~(size_t)63; };
        const size_t o_fat_off = 0, o_fat_woff = o_fat_off + padded(fat_off.size()), o_perm = o_fat_woff + padded(fat_woff.size() + 1),
                     o_elist = o_perm + padded(perm.size()), o_wave_off = o_elist + padded(elist.size() + 1), total_words = o_wave_off + padded(wave_off.size());
        Vec<uint32_t> slab(total_words, 0u);
        std::copy(fat_off.begin(), fat_off.end(), slab.begin() + (ptrdiff_t)o_fat_off);
        std::copy(fat_woff.begin(), fat_woff.end(), slab.begin() + (ptrdiff_t)o_fat_woff);
        std::copy(perm.begin(), perm.end(), slab.begin() + (ptrdiff_t)o_perm);
        std::copy(elist.begin(), elist.end(), slab.begin() + (ptrdiff_t)o_elist);
        std::copy(wave_off.begin(), wave_off.end(), slab.begin() + (ptrdiff_t)o_wave_off);
        FAST_TRY(hipMalloc(&fp->lists_slab, total_words * 4));
        FAST_TRY(plan_memcpy(fp->lists_slab, slab.data(), total_words * 4, hipMemcpyHostToDevice));
        fp->fat_off = fp->lists_slab + o_fat_off;
        fp->fat_woff = fp->lists_slab + o_fat_woff;
        fp->perm = fp->lists_slab + o_perm;
        fp->elist = fp->lists_slab + o_elist;
        fp->wave_off = fp->lists_slab + o_wave_off;
    }
    plan_tick("range: pass 2 lists made and uploaded");
    // Worst case is one record per step (plus one per block and window crossing) for k_scan and
    // one depth plus one uniq record per step for k_scan_short, spread evenly over the
    // sub-buckets; real graphs need a fraction of that (runs), skewed ones more, so the
    // capacity starts at the even share of N records + 25% and grows on demand (fast_plan_grow).
    const uint64_t slots = (uint64_t)n_win * fp->n_slots;
    uint64_t walked = 0;  // (the steps the paths span, not the pool: a plan over a few paths of a large graph needs little)
    for (uint32_t p = 0; p < g.n_paths; ++p) walked += he[p] - hb[p];
    uint64_t cap = (std::min<uint64_t>(g.n_steps, walked) + slots - 1) / slots;
    if (ranged) cap = (uint64_t)((double)cap * n_range / g.n_segs) + 1;  // a range sees its share of the runs
    cap = cap + cap / 4 + 256;
    if (const char *forced = test_hook("FLATGFA_BUCKET_CAP")) {  // tests: force the overflow route
        cap = strtoull(forced, nullptr, 10);
        fp->cap_forced = true;
    }
    // Packed buckets (below, once the items are on the device): where an even share for every sub-bucket would take
    // gigabytes -- whole-genome graphs, whose paths leave most sub-buckets of a window empty and a few deep --
    // the plan counts what every sub-bucket gets and lays them out back to back.  For tagged plans whose
    // records all come from k_scan; FLATGFA_PACKED=0|1 never / whenever possible (tests, measurements).
    const bool can_pack = fp->tagged && !fp->n_short && !fp->n_medium && !fp->n_tiny && !fp->dbg && !fp->cap_forced && n_win <= kMaxWinTagged && fp->acc_parts == 1 &&
                          scan_lds_bytes(fp->nwp, true, true) + 64 <= kLdsLimit;
    // (the even layout may take 2 GB for a graph's buckets: this plan's share when segment ranges and path groups make several of it)
    fp->can_pack = can_pack;  // (less what is found out below: pass 1 by partition keeps the even layout)
    const uint64_t even_limit = std::max<uint64_t>(128ull << 20, (2ull << 30) / std::max(1u, siblings));
    const uint64_t even_first = (slots + fp->n_slots) * std::max<uint64_t>(cap, 4) * 4;
    bool want_packed = can_pack && even_first > even_limit;
    // From half of that the counting call is asked first (a third of a query): it says how deep the deepest sub-bucket is, and
    // an even layout with headroom -- three times that for every sub-bucket -- that would pass the limit is not made at all.  (Paths
    // that run along the graph: 2000 contigs of 100 k steps on 4 M segments start at 1.25 GB, overflow in the sizing query and
    // grow to 4.8 GB, and the plan was then made a second time, packed: 11 ms to its first answer, now 3.)
    bool ask_first = can_pack && !want_packed && !t_prefer_packed && even_first > even_limit / 2;
    want_packed = want_packed || (can_pack && t_prefer_packed);  // (a plan whose even layout had to grow to gigabytes is made again, packed: flatgfa_dev_plan_create)
    if (const char *f = getenv("FLATGFA_PACKED")) {
        want_packed = can_pack && strtol(f, nullptr, 10) != 0;
        ask_first = false;
    }
    if (const char *f = test_hook("FLATGFA_PACKED_ASK")) ask_first = can_pack && !want_packed && strtol(f, nullptr, 10) != 0;  // (tests: small graphs ask too)
    want_packed = want_packed || ask_first;
    if (!want_packed) {
        const int rc = alloc_buckets(fp, std::max<uint64_t>(cap, 4));
        if (rc < 0) return false;
        if (rc == 0) return true;  // not eligible; the caller's destroy releases what was allocated
        fp->bucket_records = (slots + fp->n_slots) * (uint64_t)fp->cap;
    }
    FAST_TRY(hipMalloc(&fp->counts, slots * 4));
    FAST_TRY(hipMemset(fp->counts, 0, slots * 4));
    FAST_TRY(hipMalloc(&fp->taken, ((size_t)fp->n_slots + 1) * 4));
    FAST_TRY(hipMemset(fp->taken, 0xFF, (size_t)fp->n_slots * 4));  // (nothing known until a tagged k_scan has run)
    FAST_TRY(hipMemset(fp->taken + fp->n_slots, 0, 4));              // (the word behind them: this range's fullest sub-bucket beyond half the capacity)
    FAST_TRY(hipMalloc(&fp->counts0, slots * 4));
    FAST_TRY(hipMemset(fp->counts0, 0, slots * 4));
    FAST_TRY(hipMalloc(&fp->dir, (size_t)fp->dstride * n_win * sizeof(uint2)));
    FAST_TRY(hipMemset(fp->dir, 0, (size_t)fp->dstride * n_win * sizeof(uint2)));
    FAST_TRY(hipMalloc(&fp->islot, (size_t)fp->dstride * 4));
    FAST_TRY(hipMemset(fp->islot, 0, (size_t)fp->dstride * 4));
    plan_tick("range: buckets, cursors, directory allocated and cleared");
    FAST_TRY(hipMalloc(&fp->items, (items.size() + fp->max_back + 1) * sizeof(uint4)));
    // No path cut into pieces, and the counting kernel's facts at hand: an item is a whole path, and what k_item_dirs would find
    // out about it -- which way it mostly runs, whether it runs strictly one way, how many records it makes, how many of its
    // steps go against its grain -- is known (that kernel's read of the steps is then left out).
    const bool derived = !ext.empty() && fp->n_shared == 0 && !items.empty() && !measure_switch("FLATGFA_NO_ITEM_DIRS");
    unsigned long long derived_counts[2] = {0, 0};
    uint32_t derived_noclaim = 0;
    if (derived) {
        const char *nc0 = test_hook("FLATGFA_NO_CLAIM");
        const bool no_claim0 = !(nc0 && nc0[0] == '0');
        for (uint4 &it : items) {
            const uint32_t *x = &ext[6 * (size_t)it.w];
            const uint32_t n = it.y - it.x, pairs = n - 1u;  // (an item has at least one step)
            const uint32_t state = (x[0] == pairs ? 1u : 0u) | (x[1] == pairs ? 2u : 0u);
            it.z = (it.z & ~1u) | (x[3] > x[2] ? 1u : 0u);
            if (state && no_claim0) {
                it.z |= kItemNoClaim;
                derived_noclaim += 1;
            }
            derived_counts[0] += (unsigned long long)n - std::max(x[2], x[3]);
            derived_counts[1] += std::min(x[0], x[1]);
        }
    }
    if (!items.empty()) {
        FAST_TRY(plan_memcpy(fp->items, items.data(), items.size() * sizeof(uint4), hipMemcpyHostToDevice));
        if (!measure_switch("FLATGFA_NO_ITEM_DIRS")) {  // (measurement builds: every item taken as running upwards)
            unsigned long long *d_runs64 = nullptr, runs64 = 0, item_steps = 0, counted[2] = {0, 0};
            // (FLATGFA_NO_CLAIM=0: every item claims, monotone or not -- tests and measurements)
            const char *nc_env = test_hook("FLATGFA_NO_CLAIM");
            const bool no_claim = !(nc_env && nc_env[0] == '0');
            if (derived) {
                counted[0] = derived_counts[0];
                counted[1] = derived_counts[1];
                runs64 = counted[0];
                fp->n_noclaim = derived_noclaim;
            } else {
            FAST_TRY(hipMalloc(&d_runs64, 16));
            FAST_TRY(hipMemset(d_runs64, 0, 16));
            uint4 *d_mono = nullptr;
            if (fp->n_shared && no_claim) FAST_TRY(hipMalloc(&d_mono, (size_t)fp->n_items * sizeof(uint4)));
            hipLaunchKernelGGL(k_item_dirs, dim3(std::min<uint32_t>(fp->n_items, fp->n_cus * 8u)), dim3(256), 0, nullptr, g.steps,
                               reinterpret_cast<uint4 *>(fp->items), fp->n_items, d_runs64, d_mono);
            hipError_t e = plan_memcpy(counted, d_runs64, 16, hipMemcpyDeviceToHost);
            runs64 = counted[0];
            (void)hipFree(d_runs64);
            // The pieces of a split path: the path never meets a segment twice when every piece runs strictly one way,
            // all of them the same way, and each piece starts beyond (below) where the piece before it ended.
            Vec<uint4> dev_items;
            if (e == hipSuccess && (d_mono || !no_claim)) {
                dev_items.resize(items.size());
                e = plan_memcpy(dev_items.data(), fp->items, items.size() * sizeof(uint4), hipMemcpyDeviceToHost);
            }
            if (e == hipSuccess && !no_claim) {
                for (uint4 &it : dev_items) it.z &= ~kItemNoClaim;
                e = plan_memcpy(fp->items, dev_items.data(), items.size() * sizeof(uint4), hipMemcpyHostToDevice);
            } else if (e == hipSuccess && d_mono) {
                Vec<uint4> mono(items.size());
                e = plan_memcpy(mono.data(), d_mono, items.size() * sizeof(uint4), hipMemcpyDeviceToHost);
                if (e == hipSuccess) {
                    Vec<Vec<uint32_t>> pieces(fp->n_shared);
                    for (uint32_t j = 0; j < fp->n_items; ++j) {
                        const uint32_t sh = (dev_items[j].z & ~kItemNoClaim) >> 1;
                        if (sh) pieces[sh - 1].push_back(j);
                    }
                    bool any = false;
                    for (Vec<uint32_t> &pc : pieces) {
                        std::sort(pc.begin(), pc.end(), [&](uint32_t a, uint32_t b) { return dev_items[a].x < dev_items[b].x; });
                        uint32_t way = 3u;  // the ways all pieces so far can be read: 1 upwards, 2 downwards
                        for (size_t k = 0; k < pc.size() && way; ++k) {
                            way &= mono[pc[k]].x;
                            if (k) {
                                const uint32_t last = mono[pc[k - 1]].z, first = mono[pc[k]].y;
                                way &= (first > last ? 1u : 0u) | (first < last ? 2u : 0u);
                            }
                        }
                        if (way && !pc.empty()) {
                            for (uint32_t j : pc) dev_items[j].z |= kItemNoClaim;
                            any = true;
                        }
                    }
                    if (any) e = plan_memcpy(fp->items, dev_items.data(), items.size() * sizeof(uint4), hipMemcpyHostToDevice);
                }
            }
            if (d_mono) (void)hipFree(d_mono);
            FAST_TRY(e);
            if (no_claim) {  // how many there are (flatgfa_dev_plan_describe: no_claim_items)
                if (dev_items.empty()) {
                    dev_items.resize(items.size());
                    FAST_TRY(plan_memcpy(dev_items.data(), fp->items, items.size() * sizeof(uint4), hipMemcpyDeviceToHost));
                }
                for (const uint4 &it : dev_items) fp->n_noclaim += it.z >> 31;
            }
            }  // (!derived)
            for (const uint4 &it : items) item_steps += it.y - it.x;
            fp->item_steps = item_steps;
            // Items of paths that do not qualify as a whole: the stretches of them that lie in windows their path enters once
            // and walks one way (k_visit_bits, k_chunk_flags) -- k_scan gives the records of such blocks the no-claim tag too.
            // For one range over all segments, tagged; two bits per (path, window) and one per sixteen steps of scratch.
            // Three more reads of the steps, and not on the way to the first answer: the plan's owner runs them on a side
            // stream (fast_marks_start) and installs them between two later calls.  (Buckets laid out to the count too: a mark
            // rides on top of the ids of a whole block, and a block's first step starts a run in every build of k_scan -- the
            // marked build makes the records the counting call saw.)
            fp->marks_wanted = no_claim && fp->tagged && !ranged && fp->n_noclaim < fp->n_items && !test_hook("FLATGFA_NO_CLAIM_BLOCKS_OFF") &&
                               (((uint64_t)g.n_paths * n_win + 15) / 16 + 1) * 4 <= (256ull << 20);
            // ... and not worth looking for where the walks turn round all the time: a step against its item's grain spoils the
            // window it lies in, and with sixteen of them to a block of 1024 steps next to no block is left clean (the benchmark's
            // random walks: forty-five a block, one chunk in a thousand qualifies; a walk with a tandem repeat every 6400 steps: one
            // in six blocks has any).  FLATGFA_NO_CLAIM_BLOCKS_MIN (tests) asks for the marks regardless.
            if (fp->marks_wanted && counted[1] * 64 >= item_steps && !test_hook("FLATGFA_NO_CLAIM_BLOCKS_MIN")) fp->marks_wanted = false;
            fp->est_records = runs64;
            fp->narrow_emit = runs64 * 8 < item_steps;  // (long runs: see mode_wide)
            plan_tick("range: items uploaded, k_item_dirs");
            // more than three records for four steps: not worth looking for runs (k_scan_dense)
            const bool can = !fp->dbg && dense_lds_bytes(fp->nwp) + 64 <= kLdsLimit;
            // More than a record for two steps: k_scan_dense may be the better pass 1 -- when the
            // ids jump about; steps that stay in one window make its LDS atomics queue on one
            // address.  Sized for the dense form (one record per step is the most either makes),
            // then timed both ways by the plan's creator.
            fp->dense_maybe = can && runs64 * 2 > item_steps;
            fp->dense = fp->dense_maybe;
        }
        if (const char *f = test_hook("FLATGFA_DENSE")) {  // tests, measurements
            fp->dense = !fp->dbg && strtol(f, nullptr, 10) != 0 && dense_lds_bytes(fp->nwp) + 64 <= kLdsLimit;
            fp->dense_maybe = false;
        }
    }
    if (!short_items.empty()) {
        FAST_TRY(hipMalloc(&fp->short_items, short_items.size() * sizeof(uint4)));
        FAST_TRY(plan_memcpy(fp->short_items, short_items.data(), short_items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    if (!medium_items.empty()) {
        FAST_TRY(hipMalloc(&fp->medium_items, medium_items.size() * sizeof(uint4)));
        FAST_TRY(plan_memcpy(fp->medium_items, medium_items.data(), medium_items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    if (!tiny_items.empty()) {
        FAST_TRY(hipMalloc(&fp->tiny_items, tiny_items.size() * sizeof(uint4)));
        FAST_TRY(plan_memcpy(fp->tiny_items, tiny_items.data(), tiny_items.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    {
        Vec<uint32_t> other;
        for (const uint4 &it : short_items) other.push_back(it.w);
        for (const uint4 &it : medium_items) other.push_back(it.w);
        for (const uint4 &it : tiny_items) other.push_back(it.w);
        fp->n_other = (uint32_t)other.size();
        if (!other.empty()) {
            FAST_TRY(hipMalloc(&fp->other_ids, other.size() * 4));
            FAST_TRY(plan_memcpy(fp->other_ids, other.data(), other.size() * 4, hipMemcpyHostToDevice));
        }
    }
    // (the attribute belongs to the kernel, not to the plan: set once per device by the translation unit that holds it)
    if (!path_kernels_setup() || !scan_kernels_setup() || !accum_kernels_setup()) return false;
    FAST_TRY(hipMalloc(&fp->work_counter, 256));
    FAST_TRY(hipMemset(fp->work_counter, 0, 256));
    plan_tick("range: lists uploaded, kernel attributes");
    // Two pass-2 workgroups per window (k_accum_pair): tagged plans without split paths whose windows
    // do not fill the chip twice over anyway.  FLATGFA_ACC_PAIR=1 (measurements, tests).
    // In round 3 it paid where a window had 64 k records or more (the chromosome model 4 % faster, ten
    // thousand contigs 13 %); since the one-workgroup walk clears a plain sub-bucket's bitsets at once
    // and takes eight bitsets per wave where tags are many, it no longer does (same box: chromosome model
    // 95 against 79 us, ids without runs 175 against 149, ten thousand contigs 165 against 118) -- and the
    // table of sub-bucket starts (packed buckets) took the 4 KB of LDS that let two of its workgroups share
    // a CU: off by default.
    const bool pair_ok = fp->tagged && fp->n_shared == 0 && wb == 12 && fp->acc_parts == 1 && n_win <= 2 * fp->n_cus;
    fp->acc_pair = false;
#ifdef FGFA_MEASURE
    if (const char *f = test_hook("FLATGFA_ACC_PAIR")) fp->acc_pair = pair_ok && strtol(f, nullptr, 10) != 0;
#else
    (void)pair_ok;
#endif
    // Private bitsets per wave of the tagged walk: four, or eight where a sub-bucket holds the records of more
    // items than that (a 64-record step then spans more tags than four bitsets serve in one round: 3125 paths
    // of 32 k steps 77 -> 58 us, 16 000 paths of 100 k 0.81 -> 0.56 ms) and the LDS allows it: 4096-segment
    // windows, at most 64 split paths; one workgroup per window then (k_accum_pair has no room for them).
    // With four items or fewer per workgroup four are as good and cheaper to clear (cfg-L: +3 % with eight).
    {
        const uint32_t grid = std::min<uint32_t>(std::max<uint32_t>(fp->n_items, 1u), fp->n_slots);
        const bool can8 = fp->tagged && wb == 12 && fp->n_shared <= 64 && fp->acc_parts == 1;
        fp->acc_slots = can8 && (uint64_t)fp->n_items + fp->max_back > 4ull * grid ? 8u : kTagSlots;
        if (const char *f = test_hook("FLATGFA_ACC_SLOTS")) fp->acc_slots = can8 && strtol(f, nullptr, 10) == 8 ? 8u : kTagSlots;
        if (fp->acc_slots == 8) fp->acc_pair = false;
        if (const char *f = test_hook("FLATGFA_ACC_OWN")) fp->acc_own = strtol(f, nullptr, 10) != 0;  // tests, measurements
    }
    if (fp->acc_pair) {
        FAST_TRY(hipMalloc(&fp->pair_part, (size_t)n_win * 2 * 2 * (1u << wb) * 4));
        FAST_TRY(hipMalloc(&fp->pair_flag, (size_t)n_win * 4));
        FAST_TRY(hipMemset(fp->pair_flag, 0, (size_t)n_win * 4));
    }
    if (fp->dense || fp->dense_maybe) fp->can_pack = false;
    if (want_packed && (fp->dense || fp->dense_maybe)) {  // (pass 1 by partition keeps the even layout)
        want_packed = false;
        const int rc = alloc_buckets(fp, std::max<uint64_t>(cap, 4));
        if (rc < 0) return false;
        if (rc == 0) return true;
        fp->bucket_records = (slots + fp->n_slots) * (uint64_t)fp->cap;
    }
    if (want_packed) {
        // The counting call: k_scan alone, every sub-bucket without room (all records go to a sink), the
        // items dealt in the fixed order every later call uses.  Its cursors are the layout.
        const size_t row = (size_t)n_win + 1;
        uint32_t *d_status = nullptr;
        FAST_TRY(hipMalloc(&fp->pk_off, row * fp->n_slots * 4));
        FAST_TRY(hipMemset(fp->pk_off, 0, row * fp->n_slots * 4));
        FAST_TRY(hipMalloc(&fp->pk_base, (size_t)fp->n_slots * 8));
        FAST_TRY(hipMemset(fp->pk_base, 0, (size_t)fp->n_slots * 8));
        FAST_TRY(hipMalloc(&fp->buckets, 4096));
        FAST_TRY(hipMalloc(&d_status, 256));
        FAST_TRY(hipMemset(d_status, 0, 256));
        fp->packed = true;
        fp->cap = 0;
        fp->eligible = true;
        const uint32_t lds_even = fp->lds_bytes_scan;
        fp->lds_bytes_scan = scan_lds_bytes(fp->nwp, true, true);
        const int rc = run_range(*fp, g, nullptr, nullptr, d_status, nullptr, nullptr, true);
        Vec<uint32_t> cnt(slots);
        hipError_t e = rc == FLATGFA_OK ? hipDeviceSynchronize() : hipErrorUnknown;
        if (e == hipSuccess) e = plan_memcpy(cnt.data(), fp->counts, slots * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemset(fp->counts, 0, slots * 4);
        uint32_t st = 0;
        if (e == hipSuccess) e = plan_memcpy(&st, d_status, 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_status);
        fp->eligible = false;
        FAST_TRY(e);
        if (st & kStBounds) {  // (an id out of range: the atomic kernels report it; no layout to be had)
            fp->packed = false;
            fp->lds_bytes_scan = lds_even;
            return true;
        }
        plan_tick("range: packed buckets: the counting call");
        const bool countable = !(st & kStBackOverflow);  // (blocks without any runs do not fit a packed call's queues: the even layout)
        Vec<uint32_t> off(row * fp->n_slots);
        Vec<uint64_t> base(fp->n_slots);
        Vec<uint2> pk(slots);
        uint64_t total = 0, deepest = 0;
        bool fits = true;
        for (uint32_t gq = 0; gq < fp->n_slots; ++gq) {
            uint64_t o = 0;
            base[gq] = total;
            for (uint32_t wq = 0; wq < n_win; ++wq) {
                const uint64_t room = ((uint64_t)cnt[(size_t)wq * fp->n_slots + gq] + 3) & ~3ull;  // (sub-buckets start on 16 bytes: pass 2 reads four records at a time)
                off[gq * row + wq] = (uint32_t)o;
                pk[(size_t)wq * fp->n_slots + gq] = make_uint2((uint32_t)(total + o), (uint32_t)room);
                deepest = std::max(deepest, room);
                o += room;
            }
            off[gq * row + n_win] = (uint32_t)o;  // the region's sink
            o += 64;
            fits = fits && o < (1ull << 30);      // (a region's byte offsets take 32 bits)
            total += o;
        }
        fits = fits && countable && total < (1ull << 32);
        // (three times the deepest: a call deals the items as they come, so its fullest sub-bucket is not the counted one, and
        // what ends up more than half full is doubled for headroom -- flatgfa_dev_plan_create; twice the deepest was kept even,
        // doubled there, and made again packed after all)
        if (fits && ask_first && (slots + fp->n_slots) * std::max<uint64_t>(3 * deepest, 4) * 4 <= even_limit) {
            fits = false;  // the even layout can be had: kept (its k_scan deals the items as they come and keeps one table less in LDS)
            cap = std::max<uint64_t>(cap, 3 * deepest);
        }
        (void)hipFree(fp->buckets);
        fp->buckets = nullptr;
        if (!fits) {  // (not the case this layout is for: the even one, if it can be had)
            fp->packed = false;
            fp->lds_bytes_scan = lds_even;
            (void)hipFree(fp->pk_off);
            (void)hipFree(fp->pk_base);
            fp->pk_off = nullptr;
            fp->pk_base = nullptr;
            const int rc2 = alloc_buckets(fp, std::max<uint64_t>(cap, 4));
            if (rc2 < 0) return false;
            if (rc2 == 0) return true;
            fp->bucket_records = (slots + fp->n_slots) * (uint64_t)fp->cap;
        } else {
            FAST_TRY(hipMalloc(&fp->buckets, std::max<uint64_t>(total, 64) * 4));
            FAST_TRY(plan_memcpy(fp->pk_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
            FAST_TRY(plan_memcpy(fp->pk_base, base.data(), base.size() * 8, hipMemcpyHostToDevice));
            FAST_TRY(hipMalloc(&fp->pk, slots * sizeof(uint2)));
            FAST_TRY(plan_memcpy(fp->pk, pk.data(), slots * sizeof(uint2), hipMemcpyHostToDevice));
            fp->cap = (uint32_t)std::max<uint64_t>(deepest, 4);  // (what describe() reports: the deepest sub-bucket)
            fp->bucket_records = total;
        }
        plan_tick("range: packed buckets: laid out, allocated, tables uploaded");
    }
    fp->eligible = true;
    return true;
}

// The plans of all segment ranges for one set of path spans, appended to `plans`.  *all: every one
// of them is eligible; *many: one of them is kept from tagged calls by its number of items alone.
static bool append_ranges(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, uint32_t max_win, uint32_t force_wb,
                          std::vector<FastPlan> *plans, bool *all, bool *many, bool *want12 = nullptr, uint32_t n_groups = 1) {
    uint64_t max_range = (uint64_t)max_win << (force_wb ? force_wb : 13u);
    if (const char *f = test_hook("FLATGFA_RANGE_SEGS")) max_range = std::max<uint64_t>(8192, strtoull(f, nullptr, 10) & ~8191ull);  // tests
    const uint32_t n_ranges = (uint32_t)((g.n_segs + max_range - 1) / max_range);
    if (n_ranges > 64) {
        *all = false;
        return true;
    }
    const uint32_t per = (uint32_t)((((uint64_t)g.n_segs + n_ranges - 1) / n_ranges + 8191) & ~8191ull);
    for (uint32_t r = 0; r < n_ranges; ++r) {
        const uint32_t base = r * per;
        FastPlan q;
        if (!create_range(g, hb, he, &q, base, std::min<uint32_t>(per, g.n_segs - base), max_win, force_wb, n_ranges * n_groups)) {
            fast_plan_destroy(&q);
            return false;
        }
        *many = *many || q.too_many_items;
        if (want12) *want12 = *want12 || q.want_wb12;
        plans->push_back(q);
        if (!q.eligible) {  // all ranges or none
            *all = false;
            break;
        }
    }
    return true;
}

static void destroy_plans(std::vector<FastPlan> *plans) {
    for (FastPlan &q : *plans) fast_plan_destroy(&q);
    plans->clear();
}

static void adopt_plans(std::vector<FastPlan> *plans, FastPlan *fp) {
    *fp = (*plans)[0];
    fp->n_more = (uint32_t)plans->size() - 1;
    fp->more = fp->n_more ? new FastPlan[fp->n_more] : nullptr;
    for (uint32_t r = 0; r < fp->n_more; ++r) fp->more[r] = (*plans)[r + 1];
    plans->clear();
}

static bool fast_plan_create_impl(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, FastPlan *fp);
bool fast_plan_create(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, FastPlan *fp, uint32_t scan_workgroups, bool prefer_packed) {
    t_scan_workgroups = scan_workgroups;
    t_prefer_packed = prefer_packed;
    TempScope temporaries;  // (what the ranges' creation takes from the thread's arena is taken back when the next plan is made)
    const bool ok = fast_plan_create_impl(g, hb, he, fp);
    t_scan_workgroups = 0;
    t_prefer_packed = false;
    return ok;
}
static bool fast_plan_create_impl(const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, FastPlan *fp) {
    *fp = FastPlan();
    if (g.n_segs == 0 || g.n_paths == 0 || g.n_steps == 0) return true;
    if ((reinterpret_cast<uintptr_t>(g.steps) & 15u) != 0) return true;  // 16-byte step loads
    // One range while the graph fits 4096 windows of 8192 segments (32 M); beyond, ranges of equal
    // size, each a walk of the steps per call (64 M segments / 100 M steps: two walks).  So many
    // windows only for plans whose calls are tagged (k_scan then keeps one LDS table per window, not
    // two); when a range cannot be (FLATGFA_TAGGED=0, more split paths than pass 2 has bitsets
    // for), the ranges are cut again at 2048 windows.  FLATGFA_MAX_WINDOWS keeps the old cut-off (tests).
    //
    // A record's tag has nine bits: a k_scan workgroup can name 512 items (less the split paths).
    // When only that keeps a plan from being tagged -- 131 k paths or more on a graph too large for
    // the wave-per-path kernels -- the PATHS are walked in groups, each with plans of its own over
    // the same ranges: the first group's pass 2 stores its counts, the others' add theirs.  Every
    // step is still read once per range.  (FLATGFA_PATH_GROUPS=0: never; n: at least n groups, tests.)
    if (const char *off = test_hook("FLATGFA_MAX_WINDOWS")) {
        const uint32_t wb = g.n_segs <= 1024u * 4096u ? 12u : 13u;
        if ((((uint64_t)g.n_segs + (1u << wb) - 1) >> wb) > strtoul(off, nullptr, 10)) return true;
    }
    uint32_t want_groups = 1;
    bool groups_ok = true;
    if (const char *e = test_hook("FLATGFA_PATH_GROUPS")) {
        want_groups = (uint32_t)strtoul(e, nullptr, 10);
        groups_ok = want_groups != 0;
        want_groups = std::max(1u, std::min(64u, want_groups));
    }
    Vec<uint32_t> busy;  // the paths that have steps
    for (uint32_t p = 0; p < g.n_paths; ++p)
        if (he[p] > hb[p]) busy.push_back(p);
    // (max_win, force_wb) by preference: 4096 windows of the graph's own size; the same with 4096-segment
    // windows when the split paths of a larger graph ask for them (their bitsets need the LDS that
    // 8192-segment windows take: a graph of 16 M segments walked by ninety paths of ten million
    // steps would otherwise have no tagged plan, nor -- its buckets beyond 2^30 records -- any);
    // then 2048 windows, the untagged plans' limit.
    struct Try {
        uint32_t max_win, force_wb;
    };
    std::vector<Try> tries{{kMaxWinTagged, 0u}};
    bool tried_wb12 = false;
    for (size_t ti = 0; ti < tries.size(); ++ti) {
        const uint32_t max_win = tries[ti].max_win, force_wb = tries[ti].force_wb;
        std::vector<FastPlan> plans;
        bool all = true, many = false, want12 = false;
        if (want_groups == 1) {
            if (!append_ranges(g, hb, he, max_win, force_wb, &plans, &all, &many, &want12)) {
                destroy_plans(&plans);
                return false;
            }
            bool all_tagged = all;
            for (const FastPlan &q : plans) all_tagged = all_tagged && q.tagged;
            if (want12 && !all_tagged && !force_wb && !tried_wb12 && !test_hook("FLATGFA_WB") && !test_hook("FLATGFA_RANGE_SEGS")) {
                tried_wb12 = true;
                tries.push_back({kMaxWinTagged, 12u});  // next; if that does not yield a tagged plan either, this try comes again
                tries.push_back({max_win, 0u});
                destroy_plans(&plans);
                continue;
            }
            if (force_wb && !all_tagged && !many) {  // (the smaller windows were for tags' sake only)
                destroy_plans(&plans);
                continue;
            }
            if (all && !(many && groups_ok)) {
                // 4096-segment windows on a graph that would get 8192-segment ones, where pass 2 has use for the LDS they
                // free: a pass-1 workgroup that takes dozens of claiming items leaves every sub-bucket with dozens of
                // tags, and pass 2 hands its four private bitsets on all the time -- with the smaller windows it has
                // eight (16 000 paths of 100 k steps on 16 M segments: pass 2 2.14 -> 1.81 ms, pass 1 pays 0.1 for twice
                // the windows; with eight items per workgroup, or items that claim nothing, it is a loss: NOTES R5.12).
                const auto wants_small_windows = [&](const FastPlan &q) {
                    return q.eligible && q.tagged && q.wb == 13 && !q.n_more && g.n_segs <= (uint64_t)kMaxWinTagged << 12 && !test_hook("FLATGFA_WB") &&
                           !test_hook("FLATGFA_RANGE_SEGS") && q.n_items >= 16ull * q.n_slots && 2ull * q.n_noclaim < q.n_items && q.n_shared <= 64 &&
                           q.est_records / q.n_win >= 32768;
                };
                // ... and the other way round: 8192-segment windows on a graph of two to four million segments.  Pass 2 pays a fixed
                // part per window (its arrays cleared, 256 sub-buckets opened, the scan and the store: 15 us of a workgroup's time
                // whatever the records), and 4096-segment windows are two to four per CU there; with half as many, twice as wide,
                // cfg-L's walks on 4 M segments take 0.186 -> 0.162 ms (`k_accum` 85 -> 67 us, `k_scan` 97 -> 92: NOTES R6.9).  Only where
                // pass 2 needs nothing the wide windows have no LDS for -- eight bitsets per wave (more than four items per
                // pass-1 workgroup: 2000 contigs on 4 M segments lose 17 % of pass 2 that way), split paths' shared bitsets, the
                // wave-per-path kernels' records (they know 4096-segment windows only).
                const auto wants_wide_windows = [&](const FastPlan &q) {
                    return q.eligible && q.tagged && q.wb == 12 && !q.n_more && !q.packed && !test_hook("FLATGFA_WB") && !test_hook("FLATGFA_RANGE_SEGS") &&
                           q.n_win >= 2u * q.n_cus && q.acc_slots == kTagSlots && q.acc_parts == 1 && q.n_shared == 0 && !q.n_short && !q.n_medium && !q.n_tiny &&
                           (uint64_t)q.n_items <= 4ull * q.n_slots && !q.dense && !q.dense_maybe;
                };
                if (!force_wb && plans.size() == 1 && wants_wide_windows(plans[0])) {
                    std::vector<FastPlan> alt;
                    bool a_all = true, a_many = false;
                    if (!append_ranges(g, hb, he, kMaxWinTagged, 13u, &alt, &a_all, &a_many)) {
                        destroy_plans(&alt);
                        destroy_plans(&plans);
                        return false;
                    }
                    if (a_all && !a_many && alt.size() == 1 && alt[0].eligible && alt[0].tagged && alt[0].n_shared == 0 && !alt[0].packed) {
                        destroy_plans(&plans);
                        plans.swap(alt);
                    } else {
                        destroy_plans(&alt);
                    }
                }
                if (!force_wb && plans.size() == 1 && wants_small_windows(plans[0])) {
                    std::vector<FastPlan> alt;
                    bool a_all = true, a_many = false;
                    if (!append_ranges(g, hb, he, kMaxWinTagged, 12u, &alt, &a_all, &a_many)) {
                        destroy_plans(&alt);
                        destroy_plans(&plans);
                        return false;
                    }
                    if (a_all && !a_many && alt.size() == 1 && alt[0].eligible && alt[0].tagged && alt[0].acc_slots == 8) {
                        destroy_plans(&plans);
                        plans.swap(alt);
                    } else {
                        destroy_plans(&alt);
                    }
                }
                adopt_plans(&plans, fp);
                return true;
            }
        } else {
            many = true;
        }
        if (many && groups_ok && busy.size() >= 2) {
            // an even share of the paths per group; more groups while some group still has too many items (long paths are cut into pieces)
            uint32_t hint_items = (uint32_t)busy.size(), hint_slots = 256;
            if (!plans.empty() && plans[0].n_slots) {
                hint_slots = plans[0].n_slots;
                hint_items = std::max<uint32_t>(hint_items, plans[0].n_items);
            }
            destroy_plans(&plans);
            uint32_t n_groups = std::max<uint32_t>(want_groups, (uint32_t)((hint_items + 384ull * hint_slots - 1) / (384ull * hint_slots)));
            n_groups = std::max(n_groups, 2u);
            for (; n_groups <= 64 && n_groups <= busy.size(); n_groups *= 2) {
                bool g_all = true, g_many = false, hip_ok = true;
                Vec<uint32_t> hbk(g.n_paths), hek(g.n_paths);
                for (uint32_t k = 0; k < n_groups && g_all && hip_ok; ++k) {
                    const size_t lo = busy.size() * k / n_groups, hi = busy.size() * (k + 1) / n_groups;
                    for (uint32_t p = 0; p < g.n_paths; ++p) hbk[p] = hek[p] = hb[p];  // (a path outside the group: no steps)
                    for (size_t i = lo; i < hi; ++i) hek[busy[i]] = he[busy[i]];
                    const size_t first = plans.size();
                    hip_ok = append_ranges(g, hbk.data(), hek.data(), max_win, force_wb, &plans, &g_all, &g_many, nullptr, n_groups);
                    for (size_t i = first; i < plans.size(); ++i) {
                        plans[i].accumulate = k > 0;
                        g_all = g_all && plans[i].tagged;  // (a group is only worth it tagged)
                    }
                }
                if (!hip_ok) {
                    destroy_plans(&plans);
                    return false;
                }
                if (g_all && !plans.empty()) {
                    adopt_plans(&plans, fp);
                    fp->n_groups = n_groups;
                    return true;
                }
                destroy_plans(&plans);
                if (!g_many) break;  // (something else stands in the way)
            }
            if (force_wb) continue;  // (the smaller windows were for tags' sake only)
            // no luck: the plan the whole path set gets
            all = true;
            many = false;
            if (!append_ranges(g, hb, he, max_win, force_wb, &plans, &all, &many)) {
                destroy_plans(&plans);
                return false;
            }
            if (all) {
                adopt_plans(&plans, fp);
                return true;
            }
        }
        destroy_plans(&plans);
        if (ti + 1 == tries.size() && max_win == kMaxWinTagged && !force_wb) {
            const uint64_t max_range = test_hook("FLATGFA_RANGE_SEGS") ? 0 : (uint64_t)max_win << 13;
            const uint64_t n_ranges = max_range ? (g.n_segs + max_range - 1) / max_range : 1;
            if (((uint64_t)g.n_segs + 8191) / 8192 > kMaxWin * n_ranges) tries.push_back({kMaxWin, 0u});  // (else the smaller cut-off would make the same ranges)
        }
    }
    return true;
}

#ifndef FGFA_SKIP_FLAG_CLEAR
#define FGFA_SKIP_FLAG_CLEAR 0  /* a test's build: the marks of overlapping spans not taken away again (tests/test_gpu_depth.py::test_no_claim_marks_where_spans_overlap must then fail) */
#endif

static void marks_release(MarksJob *job) {
    for (void *p : {(void *)job->vis, (void *)job->pbeg, (void *)job->chunks, (void *)job->flags, (void *)job->cnt})
        if (p) (void)hipFree(p);
    if (job->done) (void)hipEventDestroy(job->done);
    *job = MarksJob();
}

bool fast_marks_start(const FastPlan &fp, const flatgfa_dev_graph_t &g, const uint32_t *hb, hipStream_t side, MarksJob *job) {
    *job = MarksJob();
    if (!fp.marks_wanted || !fp.items || !fp.n_items) return true;
    const uint64_t vis_words = ((uint64_t)g.n_paths * fp.n_win + 15) / 16 + 1, flag_words = g.n_steps / 512 + 4;
    hipError_t e = hipMalloc(&job->vis, vis_words * 4);
    if (e == hipSuccess) e = hipMemsetAsync(job->vis, 0, vis_words * 4, side);
    if (e == hipSuccess) e = hipMalloc(&job->pbeg, (size_t)g.n_paths * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(job->pbeg, hb, (size_t)g.n_paths * 4, hipMemcpyHostToDevice, side);
    if (e == hipSuccess) e = hipMalloc(&job->cnt, 32);
    if (e == hipSuccess) e = hipMemsetAsync(job->cnt, 0, 32, side);
    if (e == hipSuccess) e = hipMalloc(&job->chunks, flag_words * 4);
    if (e == hipSuccess) e = hipMemsetAsync(job->chunks, 0, flag_words * 4, side);
    if (e == hipSuccess) e = hipMalloc(&job->flags, flag_words * 4);
    if (e == hipSuccess) e = hipMemsetAsync(job->flags, 0, flag_words * 4, side);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&job->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error(std::string("no-claim marks: ") + hipGetErrorString(e));
        marks_release(job);
        return false;
    }
    const dim3 grid(std::min<uint32_t>(fp.n_items, fp.n_cus * 8u));
    const uint4 *items = reinterpret_cast<const uint4 *>(fp.items);
    hipLaunchKernelGGL(k_visit_bits, grid, dim3(256), 0, side, g.steps, items, fp.n_items, job->pbeg, fp.wb, fp.n_win, job->vis);
    for (int clear = 0; clear < (FGFA_SKIP_FLAG_CLEAR ? 1 : 2); ++clear)
        hipLaunchKernelGGL(k_chunk_flags, grid, dim3(256), 0, side, g.steps, items, fp.n_items, clear != 0, fp.wb, fp.n_win, job->vis, job->chunks, job->cnt);
    // (cnt[2]: the chunks of claiming items' blocks that qualify as wholes; where spans overlap, before another path took a mark away)
    for (int clear = 0; clear < (FGFA_SKIP_FLAG_CLEAR ? 1 : 2); ++clear)
        hipLaunchKernelGGL(k_block_flags, grid, dim3(256), 0, side, items, fp.n_items, g.n_steps, clear != 0, job->chunks, job->flags, job->cnt + 2);
    e = hipEventRecord(job->done, side);
    if (e != hipSuccess || hipGetLastError() != hipSuccess) {
        set_error("no-claim marks: launch failed");
        (void)hipStreamSynchronize(side);
        marks_release(job);
        return false;
    }
    job->active = true;
    return true;
}

bool fast_marks_ready(const MarksJob &job) {
    if (!job.active) return true;
    const hipError_t e = hipEventQuery(job.done);
    if (e != hipSuccess) (void)hipGetLastError();  // ("not ready" must not be what a later check of the launches finds)
    return e == hipSuccess;
}

void fast_marks_finish(FastPlan *fp, MarksJob *job) {
    if (!job->active) return;
    unsigned long long cnt[4] = {0, 0, 0, 0};
    hipError_t e = hipEventSynchronize(job->done);
    if (e == hipSuccess) e = plan_memcpy(cnt, job->cnt, 32, hipMemcpyDeviceToHost);
    const unsigned long long flagged = cnt[2];
    // (worth it from half of the chunks: pass 2's build with the no-claim test costs the claiming records 4 %, k_scan's with the
    // marks 1-10 % (short items most), and the claims are two fifths of pass 2 -- contigs of ten blocks with a third of
    // their chunks marked lost 7 % of the call; cfg-L's random walks have one chunk in a thousand that qualifies)
    uint64_t min_pct = 50;
    if (const char *f = test_hook("FLATGFA_NO_CLAIM_BLOCKS_MIN")) min_pct = strtoull(f, nullptr, 10);  // tests, measurements
    if (e == hipSuccess && flagged != 0 && flagged * 1600 >= fp->item_steps * min_pct && !fp->cflags) {
        fp->cflags = job->flags;
        job->flags = nullptr;
        fp->n_flag_chunks = flagged;
    }
    if (e != hipSuccess) (void)hipGetLastError();
    marks_release(job);
}

// ---- FLATGFA_CHECK_NO_CLAIM=1: what a plan takes for granted about the step values, looked at again before a call ----
// Every listed stretch of steps (x .. y of `steps`; `first` = where its path starts: pbeg[w], or x itself) must walk the
// segment ids strictly one way, the steps before it (down to `first`) included: a local test on three consecutive steps.
__global__ __launch_bounds__(256) void k_check_mono(const uint32_t *__restrict__ steps, const uint4 *__restrict__ list, uint32_t n, const uint32_t *__restrict__ pbeg,
                                                     uint32_t need_flag, uint32_t *__restrict__ status) {
    for (uint32_t j = blockIdx.x; j < n; j += gridDim.x) {
        const uint4 it = list[j];
        if (need_flag && !(it.z & need_flag)) continue;
        const uint64_t first = pbeg ? pbeg[it.w] : it.x;
        bool bad = false;
        for (uint64_t t = (uint64_t)it.x + threadIdx.x; t < it.y; t += 256) {
            if (t < first + 1) continue;
            const uint32_t a = steps[t] >> 1, b = steps[t - 1] >> 1;
            bad = bad || a == b;
            if (t >= first + 2) {
                const uint32_t c = steps[t - 2] >> 1;
                bad = bad || b == c || ((a > b) != (b > c));
            }
        }
        if (bad) atomicOr(status, kStStale);
    }
}
// a reversed copy (x .. y of `rev`) against its path's steps as they are now
__global__ __launch_bounds__(256) void k_check_rev(const uint32_t *__restrict__ steps, const uint32_t *__restrict__ rev, const uint4 *__restrict__ list, uint32_t n,
                                                    const uint32_t *__restrict__ pend, uint32_t *__restrict__ status) {
    for (uint32_t j = blockIdx.x; j < n; j += gridDim.x) {
        const uint4 it = list[j];
        const uint64_t e = pend[it.w];
        bool bad = false;
        for (uint32_t i = threadIdx.x; i < it.y - it.x; i += 256) bad = bad || rev[it.x + i] != steps[e - 1u - i];
        if (bad) atomicOr(status, kStStale);
    }
}
// a mark the plan holds that the step values no longer earn
__global__ __launch_bounds__(256) void k_check_flags(const uint32_t *__restrict__ held, const uint32_t *__restrict__ fresh, uint64_t n_words, uint32_t *__restrict__ status) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256) bad = bad || (held[i] & ~fresh[i]) != 0u;
    if (bad) atomicOr(status, kStStale);
}

static int check_range(const FastPlan &fp, const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *d_pbeg, const uint32_t *d_pend, uint32_t *status, hipStream_t stream) {
    if (!fp.eligible) return FLATGFA_OK;
    const uint32_t grid = fp.n_cus * 8u;
    if (fp.n_items && fp.n_noclaim)
        hipLaunchKernelGGL(k_check_mono, dim3(std::min(fp.n_items, grid)), dim3(256), 0, stream, g.steps, reinterpret_cast<const uint4 *>(fp.items), fp.n_items, d_pbeg, kItemNoClaim, status);
    struct L { const void *list; uint32_t n, n_rev, mono_lo, mono_n; };
    for (const L &l : {L{fp.short_items, fp.n_short, fp.n_short_rev, fp.short_mono_lo, fp.short_mono_n}, L{fp.medium_items, fp.n_medium, fp.n_medium_rev, fp.medium_mono_lo, fp.medium_mono_n},
                       L{fp.tiny_items, fp.n_tiny, 0u, fp.tiny_mono_lo, fp.tiny_mono_n}}) {
        if (!l.n) continue;
        const uint4 *list = reinterpret_cast<const uint4 *>(l.list);
        const uint32_t n_fwd = l.n - l.n_rev;
        // the paths taken as strictly one way: those read from the graph's steps, then those read from their reversed copies
        const uint32_t fwd_mono = l.mono_lo < n_fwd ? std::min(l.mono_n, n_fwd - l.mono_lo) : 0u, rev_mono = l.mono_n - fwd_mono;
        if (fwd_mono) hipLaunchKernelGGL(k_check_mono, dim3(std::min(fwd_mono, grid)), dim3(256), 0, stream, g.steps, list + l.mono_lo, fwd_mono, (const uint32_t *)nullptr, 0u, status);
        if (rev_mono) hipLaunchKernelGGL(k_check_mono, dim3(std::min(rev_mono, grid)), dim3(256), 0, stream, (const uint32_t *)fp.rev_steps, list + l.mono_lo + fwd_mono, rev_mono, (const uint32_t *)nullptr, 0u, status);
        if (l.n_rev) hipLaunchKernelGGL(k_check_rev, dim3(std::min(l.n_rev, grid)), dim3(256), 0, stream, g.steps, (const uint32_t *)fp.rev_steps, list + n_fwd, l.n_rev, d_pend, status);
    }
    if (fp.cflags) {  // the marks, made again from the steps as they are
        FastPlan again = fp;
        again.marks_wanted = true;
        MarksJob job;
        if (!fast_marks_start(again, g, hb, stream, &job)) return FLATGFA_ERR_HIP;
        if (job.active) {
            const uint64_t flag_words = g.n_steps / 512 + 4;
            hipLaunchKernelGGL(k_check_flags, dim3(256), dim3(256), 0, stream, (const uint32_t *)fp.cflags, (const uint32_t *)job.flags, flag_words, status);
            (void)hipStreamSynchronize(stream);
            marks_release(&job);
        }
    }
    return hipGetLastError() == hipSuccess ? FLATGFA_OK : FLATGFA_ERR_HIP;
}

int fast_check_plan_facts(const FastPlan &fp, const flatgfa_dev_graph_t &g, const uint32_t *hb, const uint32_t *he, uint32_t *status, hipStream_t stream) {
    if (!fp.eligible || !g.n_paths) return FLATGFA_OK;
    uint32_t *d_spans = nullptr;
    if (hipMalloc(&d_spans, (size_t)g.n_paths * 8) != hipSuccess) { (void)hipGetLastError(); set_error("check: hipMalloc"); return FLATGFA_ERR_HIP; }
    int rc = FLATGFA_OK;
    if (hipMemcpyAsync(d_spans, hb, (size_t)g.n_paths * 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(d_spans + g.n_paths, he, (size_t)g.n_paths * 4, hipMemcpyHostToDevice, stream) != hipSuccess) rc = FLATGFA_ERR_HIP;
    // (path groups: a group's plan was made with spans of its own -- empty ones for the other groups' paths --, but its items and
    // lists only name paths of its own, whose spans are the graph's)
    if (rc == FLATGFA_OK) rc = check_range(fp, g, hb, d_spans, d_spans + g.n_paths, status, stream);
    for (uint32_t r = 0; r < fp.n_more && rc == FLATGFA_OK; ++r) rc = check_range(fp.more[r], g, hb, d_spans, d_spans + g.n_paths, status, stream);
    (void)hipStreamSynchronize(stream);
    (void)hipFree(d_spans);
    return rc;
}

// Scratch for path sums riding on seg_depth: one {sum len, sum depth * len} per (window, item).
// False when that would be out of proportion (then the caller walks the steps a second time).
bool fast_plan_want_path_sums(FastPlan *fp) {
    if (fp->packed) return false;  // (its calls are tagged; the fused sums ride on the directory)
    if (!fp->eligible || fp->wb != 12 || fp->acc_parts > 1 || fp->n_more || fp->n_win > kMaxWin) return false;  // (the fused form needs a window's final depth in one workgroup, and the directory: k_scan's untagged build)
    if (((uint64_t)fp->n_win + 1) * fp->n_slots * fp->cap >= (1ull << 30)) return false;  // (... which knows 32-bit bucket offsets only)
    if (fp->psum_part) return true;
    const uint64_t bytes = (uint64_t)fp->n_win * fp->dstride * 16;
    if (bytes > (256ull << 20)) return false;
    if (hipMalloc(&fp->psum_part, std::max<uint64_t>(bytes, 16)) != hipSuccess) {
        fp->psum_part = nullptr;
        (void)hipGetLastError();
        return false;
    }
    return true;
}

// After a call that ran out of sub-bucket room: four times the capacity, if that is possible (twice, ahead of need).
static bool grow_range(FastPlan *fp, uint32_t factor) {
    const uint32_t before = fp->cap;
    if (alloc_buckets(fp, (uint64_t)before * factor) <= 0) return false;
    return fp->cap > before;  // else the slot arithmetic allows no more
}

bool fast_plan_grow(FastPlan *fp, bool ahead_of_need) {
    if (!fp->eligible || fp->cap_forced) return false;
    if (fp->packed || std::any_of(fp->more, fp->more + fp->n_more, [](const FastPlan &q) { return q.packed; })) {
        // packed buckets have the room the counted call needed and no more: a call that makes other
        // records (the steps changed behind the plan) is completed through the atomic kernels
        if (!ahead_of_need) fp->eligible = false;
        return false;
    }
    const uint32_t factor = ahead_of_need ? 2u : 4u;  // (ahead of need: what was more than half full is then at most half full)
    // Ahead of need only the ranges (or path groups) whose own word says so are given more room: their
    // bucket arrays may be gigabytes each, and one hot window is no reason to double them all.
    const auto wants = [&](FastPlan *q) {
        if (!ahead_of_need || !q->taken) return true;
        uint32_t v = 0;
        if (plan_memcpy(&v, q->taken + q->n_slots, 4, hipMemcpyDeviceToHost) != hipSuccess) return true;
        if (v) (void)hipMemset(q->taken + q->n_slots, 0, 4);
        return v > (q->cap >> 1);
    };
    bool ok = wants(fp) ? grow_range(fp, factor) : true;
    for (uint32_t r = 0; r < fp->n_more && (ok || ahead_of_need); ++r)
        if (wants(&fp->more[r])) ok = grow_range(&fp->more[r], factor) && ok;  // (after an overflow: the status word does not say which range ran out)
    if (!ok && !ahead_of_need) fp->eligible = false;  // the atomic kernels take over
    return ok;
}

void fast_plan_destroy(FastPlan *fp) {
    for (uint32_t r = 0; r < fp->n_more; ++r) fast_plan_destroy(&fp->more[r]);
    delete[] fp->more;
    // (perm, elist, wave_off, fat_off and fat_woff lie in lists_slab)
    if (fp->buckets && !fp->packed && fp->cap) {
        bucket_cache_give(fp->buckets, ((uint64_t)fp->n_win + 1) * fp->n_slots * (uint64_t)fp->cap * 4);
        fp->buckets = nullptr;
    }
    for (void *p : {(void *)fp->counts, (void *)fp->counts0, (void *)fp->buckets, (void *)fp->dir, (void *)fp->islot, (void *)fp->lists_slab,
                    (void *)fp->items, (void *)fp->short_items,
                    (void *)fp->medium_items, (void *)fp->tiny_items, (void *)fp->rev_steps, (void *)fp->work_counter, (void *)fp->other_ids, (void *)fp->psum_part,
                    (void *)fp->pair_part, (void *)fp->pair_flag, (void *)fp->taken, (void *)fp->pk_off, (void *)fp->pk_base, (void *)fp->pk, (void *)fp->cflags})
        if (p) (void)hipFree(p);
    *fp = FastPlan();
}

// One range of the graph: the outputs are the range's own stretch of the result vectors.
static int run_range(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                     uint32_t *status, hipStream_t stream, const PathSums *ps, bool count_only) {
    if (ps && (uniq_out || fp.wb != 12 || !g.seg_len || !fp.psum_part || fp.n_range != g.n_segs)) {
        set_error("fast_seg_depth: path sums ride on seg_depth with 4096-segment windows only");
        return FLATGFA_ERR_ARG;
    }
    const uint32_t stride = fp.n_slots * fp.cap;
    const bool has_pre = fp.n_short || fp.n_medium || fp.n_tiny;
    // one persistent workgroup per CU; k_scan may be handed short paths back, so it gets a full grid when there are any
    // (and whenever the wave-per-path kernels ran: it saves their cursors for pass 2)
    // -- unless it has no items of its own and nothing can come back (the run counts of the lists are
    // exact): then the launch is left out, every record is one of the wave-per-path kernels', and a
    // path that does not fit after all (steps changed behind the plan) raises kStBackOverflow.
    const bool scan_skip = has_pre && fp.n_items == 0 && fp.exact_short && !fp.dbg && !test_hook("FLATGFA_SCAN_ALWAYS");
    const uint32_t grid = scan_skip ? 0u : has_pre ? fp.n_slots : std::min<uint32_t>(fp.n_items, fp.n_slots);
    ScanArgs sa;
    sa.zero_a = sa.zero_b = nullptr;
    sa.zero_c = sa.zero_d = nullptr;
    sa.n_zero64 = 0;
    sa.path_begin = sa.path_end = nullptr;
    sa.rev_steps = nullptr;
    sa.n_fwd = ~0u;
    sa.mono_lo = sa.mono_n = 0u;
    sa.steps = g.steps;
    sa.n_steps = g.n_steps;
    sa.items = reinterpret_cast<uint4 *>(fp.items);
    sa.short_items = reinterpret_cast<const uint4 *>(fp.short_items);
    sa.n_short = fp.n_short;
    sa.n_items = fp.n_items;
    sa.n_segs = fp.n_range;
    sa.seg_base = fp.seg_base;
    sa.n_total = g.n_segs;
    sa.ranged = (fp.seg_base != 0 || fp.n_range != g.n_segs) ? 1u : 0u;
    sa.n_win = fp.n_win;
    sa.n_slots = fp.n_slots;
    sa.wb = fp.wb;
    sa.nwp = fp.nwp;
    sa.has_pre = has_pre ? 1u : 0u;
    sa.max_back = scan_skip ? 0u : fp.max_back;
    sa.work_counter = fp.work_counter;
    sa.counts = fp.counts;
    sa.counts0 = fp.counts0;
    sa.buckets = fp.buckets;
    sa.dir = reinterpret_cast<uint2 *>(fp.dir);
    sa.islot = fp.islot;
    sa.perm = fp.perm;
    sa.dstride = fp.dstride;
    sa.cap = fp.cap;
    sa.stride = stride;
    sa.sink = fp.n_win * stride;
    sa.big = ((uint64_t)fp.n_win + 1) * stride >= (1ull << 30) ? 1u : 0u;
    sa.status = status;
    sa.dbg = fp.dbg;
    // Tagged: k_scan's records name their items, pass 2 walks whole sub-buckets.  Path sums ride on
    // the directory walk (an item's records have to be found again once the window's depth is final).
    const bool tagged = fp.tagged && !ps;
    sa.tagged = tagged ? 1u : 0u;
    sa.tag_limit = std::max(2u, fp.tag_limit);
    sa.taken = fp.taken;
    sa.mall_steps = fp.mall_steps;
    sa.cflags = tagged ? fp.cflags : nullptr;
    sa.pk_off = fp.pk_off;
    sa.pk_base = fp.pk_base;
    if (fp.packed && !tagged) { set_error("fast_seg_depth: a plan with packed buckets runs tagged calls only"); return FLATGFA_ERR_ARG; }
    sa.tprof = nullptr;
    if (test_hook("FLATGFA_SCAN_TIME") && hipMalloc(&sa.tprof, kTprofRow * 8 * (size_t)fp.n_slots) == hipSuccess) (void)hipMemset(sa.tprof, 0, kTprofRow * 8 * (size_t)fp.n_slots);
    AccArgs aa{fp.n_range, fp.n_win, fp.n_slots, fp.cap, fp.counts, fp.counts0, scan_skip ? 2u : has_pre ? 1u : 0u, fp.buckets,
               reinterpret_cast<const uint2 *>(fp.dir), fp.islot, fp.dstride, fp.elist, fp.wave_off, fp.n_items,
               fp.work_counter, scan_skip ? 0u : fp.max_back, depth_out, uniq_out, status, fp.dbg,
               reinterpret_cast<const uint4 *>(fp.items), g.seg_len, ps ? reinterpret_cast<ulonglong2 *>(fp.psum_part) : nullptr,
               fp.fat_off, fp.fat_woff, fp.acc_parts, tagged ? fp.n_shared : 0u, nullptr, fp.pair_part, fp.pair_flag, fp.accumulate ? 1u : 0u, test_hook("FLATGFA_NO_PLAIN") ? nullptr : fp.taken, fp.taken ? fp.taken + fp.n_slots : nullptr,
               fp.packed ? reinterpret_cast<const uint2 *>(fp.pk) : nullptr};  // (FLATGFA_NO_PLAIN: measurements)
#ifdef FGFA_MEASURE
    if (const char *sk = test_hook("FLATGFA_ACC_SKIP")) aa.dbg = (uint32_t)strtoul(sk, nullptr, 10);  // (diagnostic: pass 2 without its revisit counts 128 / depth 256 / claims 64 / words behind the first 1024)
#endif
    const size_t tprof_words = (size_t)fp.n_win * fp.acc_parts * kAccWaves * 16;
    if (test_hook("FLATGFA_ACC_TIME") && uniq_out && hipMalloc(&aa.tprof, tprof_words * 4) != hipSuccess) aa.tprof = nullptr;
    // The wave-per-path kernels: the paths read from the graph's steps, then those read from their
    // reversed copies (a handed-back one is walked by k_scan from the graph's own steps).
    if (fp.n_short && hipMemsetAsync(fp.work_counter, 0, 4, stream) != hipSuccess) return FLATGFA_ERR_HIP;
    if (fp.n_tiny) {  // paths a wave holds whole (at most 128 steps)
        ScanArgs sk = sa;
        sk.short_items = reinterpret_cast<const uint4 *>(fp.tiny_items);
        sk.n_short = fp.n_tiny;
        sk.mono_lo = fp.tiny_mono_lo;
        sk.mono_n = fp.tiny_mono_n;
        const uint32_t kgrid = std::min<uint32_t>((fp.n_tiny + kWaves - 1) / kWaves, fp.n_slots);
        ProfScope pscope(uniq_out ? "k_scan_tiny<uniq>" : "k_scan_tiny<depth>", stream);
        launch_scan_tiny(fp, sk, uniq_out != nullptr, kgrid, stream);
    }
    for (int medium = 0; medium < 2; ++medium) {
        const uint32_t n_all = medium ? fp.n_medium : fp.n_short, n_rev = medium ? fp.n_medium_rev : fp.n_short_rev;
        const uint4 *list = reinterpret_cast<const uint4 *>(medium ? fp.medium_items : fp.short_items);
        {
            const uint32_t n = n_all;
            if (!n) continue;
            ScanArgs sk = sa;
            sk.short_items = list;
            sk.n_short = n;
            sk.n_fwd = n_all - n_rev;  // (the reversed ones lie behind the others in the list)
            sk.mono_lo = medium ? fp.medium_mono_lo : fp.short_mono_lo;
            sk.mono_n = medium ? fp.medium_mono_n : fp.short_mono_n;
            sk.rev_steps = fp.rev_steps;
            sk.path_begin = g.path_begin;
            sk.path_end = g.path_end;
            const uint32_t per_wg = medium ? (kMediumPaired ? kMediumWaves / 2 : kMediumWaves) : kShortWaves;  // paths a workgroup walks at a time
            const uint32_t kgrid = std::min<uint32_t>((n + per_wg - 1) / per_wg, fp.n_slots);
            ProfScope pscope(medium ? (uniq_out ? "k_scan_medium<uniq>" : "k_scan_medium<depth>") : (uniq_out ? "k_scan_short<uniq>" : "k_scan_short<depth>"), stream);
            launch_scan_short(fp, sk, medium != 0, uniq_out != nullptr, kgrid, stream);
        }
    }
    if (ps && ps->clear) {  // the sums k_path_reduce adds to start at zero: k_scan's first act, or two memsets when it does not run
        if (grid) {
            sa.zero_c = (unsigned long long *)ps->len_out;
            sa.zero_d = (unsigned long long *)ps->weighted_out;
            sa.n_zero64 = g.n_paths;
        } else {
            ProfScope pscope("memset_path_sums", stream);
            if (hipMemsetAsync(ps->len_out, 0, (size_t)g.n_paths * 8, stream) != hipSuccess) return FLATGFA_ERR_HIP;
            if (hipMemsetAsync(ps->weighted_out, 0, (size_t)g.n_paths * 8, stream) != hipSuccess) return FLATGFA_ERR_HIP;
        }
    }
    if (grid) {
        if (fp.acc_parts > 1 && !fp.accumulate && !count_only) {  // (a later group of paths adds to what is there)
            sa.zero_a = depth_out;
            sa.zero_b = uniq_out;
        }
        ProfScope pscope(fp.dense ? "k_scan_dense" : "k_scan", stream);
        const int rc = launch_scan(fp, sa, tagged, grid, stream);
        if (rc != FLATGFA_OK) return rc;
    }
    if (count_only) return hipGetLastError() == hipSuccess ? FLATGFA_OK : FLATGFA_ERR_HIP;  // (the counting call of a packed plan: pass 1 alone)
    if (fp.acc_parts > 1 && !grid && !fp.accumulate) {  // the window's workgroups add to the outputs: cleared by k_scan, or here when it does not run
        ProfScope pscope("memset_outputs", stream);
        if (hipMemsetAsync(depth_out, 0, (size_t)fp.n_range * 4, stream) != hipSuccess) return FLATGFA_ERR_HIP;
        if (uniq_out && hipMemsetAsync(uniq_out, 0, (size_t)fp.n_range * 4, stream) != hipSuccess) return FLATGFA_ERR_HIP;
    }
    {
        ProfScope pscope(uniq_out ? "k_accum<uniq>" : (ps ? "k_accum<depth+paths>" : "k_accum<depth>"), stream);
        launch_accum(fp, aa, uniq_out != nullptr, tagged, ps != nullptr, stream);
    }
    if (ps && fp.n_items) {
        ProfScope pscope("k_path_reduce", stream);
        launch_path_reduce(fp, (unsigned long long *)ps->len_out, (unsigned long long *)ps->weighted_out, stream);
    }
    if (hipGetLastError() != hipSuccess) {
        set_error("fast_seg_depth: kernel launch failed");
        return FLATGFA_ERR_HIP;
    }
    if (sa.tprof) {  // diagnostic: when the workgroups of k_scan start, when their first wave runs out of work, when they end
        constexpr size_t kRow = kTprofRow;
        std::vector<unsigned long long> raw(kRow * (size_t)fp.n_slots);
        (void)hipStreamSynchronize(stream);
        (void)plan_memcpy(raw.data(), sa.tprof, raw.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(sa.tprof);
        unsigned long long t0 = ~0ull;
        for (uint32_t i = 0; i < grid; ++i) t0 = std::min(t0, raw[kRow * i]);
        std::vector<double> st, fw, lw, en;
        for (uint32_t i = 0; i < grid; ++i) {
            st.push_back((raw[kRow * i] - t0) / 100.0);
            en.push_back((raw[kRow * i + 1] - t0) / 100.0);
            unsigned long long a = ~0ull, b = 0;
            for (size_t k = 0; k < kWaves; ++k) {
                a = std::min(a, raw[kRow * i + 4 + k]);
                b = std::max(b, raw[kRow * i + 4 + k]);
            }
            fw.push_back((a - t0) / 100.0);
            lw.push_back((b - t0) / 100.0);
        }
        const auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        fprintf(stderr, "k_scan%s workgroups (us since the first one started): start p50 %.1f max %.1f | first wave out of work p5 %.1f p50 %.1f p95 %.1f | last wave p5 %.1f p50 %.1f p95 %.1f max %.1f | end p50 %.1f max %.1f\n",
                tagged ? " [tagged]" : "", pct(st, 0.5), pct(st, 1.0), pct(fw, 0.05), pct(fw, 0.5), pct(fw, 0.95), pct(lw, 0.05), pct(lw, 0.5), pct(lw, 0.95), pct(lw, 1.0), pct(en, 0.5), pct(en, 1.0));
        // FLATGFA_SCAN_TIME=<file>: one line per workgroup and call -- call, workgroup, XCC, HW_ID, items, start, first / last wave out of work, end (us)
        static int call_no = 0;
        const char *where = test_hook("FLATGFA_SCAN_TIME");
        if (where && strchr(where, '/')) {
            if (FILE *f = fopen(where, "a")) {
                for (uint32_t i = 0; i < grid; ++i)
                    fprintf(f, "%d,%u,%u,0x%08x,%u,%.2f,%.2f,%.2f,%.2f\n", call_no, i, (unsigned)(raw[kRow * i + 2] >> 32) & 15u, (unsigned)raw[kRow * i + 2],
                            (unsigned)raw[kRow * i + 3], st[i], fw[i], lw[i], en[i]);
                fclose(f);
            }
        }
        call_no += 1;
    }
    if (aa.tprof) {  // diagnostic: where the waves of pass 2 spend their time
        std::vector<uint32_t> raw(tprof_words);
        (void)hipStreamSynchronize(stream);
        (void)plan_memcpy(raw.data(), aa.tprof, tprof_words * 4, hipMemcpyDeviceToHost);
        (void)hipFree(aa.tprof);
        const size_t waves = tprof_words / 16;
        double sum[16] = {}, mx[16] = {};
        for (size_t w = 0; w < waves; ++w)
            for (int k = 0; k < 16; ++k) {
                sum[k] += raw[w * 16 + k];
                mx[k] = std::max<double>(mx[k], raw[w * 16 + k]);
            }
        static const char *names[5] = {"setup", "flat", "walk", "barrier", "scan+store"};
        fprintf(stderr, "k_accum%s per wave, us avg (max):", tagged ? " [tagged]" : "");
        for (int k = 0; k < 5; ++k) fprintf(stderr, "  %s %.2f (%.2f)", names[k], sum[k] / waves / 100.0, mx[k] / 100.0);
        fprintf(stderr, "  | per wave avg (max): steps %.1f (%.0f) rounds %.1f flushes %.1f (%.0f)\n", sum[8] / waves, mx[8], sum[9] / waves, sum[10] / waves, mx[10]);
    }
    if (fp.dbg & kDbgTime) {  // diagnostic: where the waves of k_scan spend their cycles
        unsigned long long acc[8] = {};
        (void)hipStreamSynchronize(stream);
        (void)plan_memcpy(acc, status + 8, sizeof acc, hipMemcpyDeviceToHost);
        (void)hipMemset(status + 8, 0, sizeof acc);
        const double waves = (double)grid * kWaves;
        fprintf(stderr, "k_scan cycles per wave: wait_block %.0f  epoch_wait %.0f  passA+B %.0f  drain %.0f  other %.0f  item switch %.0f\n",
                acc[0] / waves, acc[1] / waves, acc[2] / waves, acc[3] / waves, acc[4] / waves, acc[5] / waves);
    }
    return FLATGFA_OK;
}

int fast_seg_depth(const FastPlan &fp, const flatgfa_dev_graph_t &g, uint32_t *depth_out, uint32_t *uniq_out,
                   uint32_t *status, hipStream_t stream, const PathSums *ps) {
    int rc = run_range(fp, g, depth_out, uniq_out, status, stream, ps, false);
    for (uint32_t r = 0; r < fp.n_more && rc == FLATGFA_OK; ++r) {
        const FastPlan &q = fp.more[r];
        rc = run_range(q, g, depth_out + q.seg_base, uniq_out ? uniq_out + q.seg_base : nullptr, status, stream, nullptr, false);
    }
    return rc;
}

}  // namespace fgfa_dev

