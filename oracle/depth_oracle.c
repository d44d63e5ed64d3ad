/*
 * depth_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the reference's node-depth /
 * path-depth algorithms (cucapra/pollen, flatgfa/src/ops/depth.rs).  It exists
 * so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
 * check the HIP product path bit-for-bit.  Nothing under pollen_amd/ links,
 * loads or calls this file; the product has no CPU fallback.
 *
 * Parity pinning: this restatement is checked (tests/test_oracle.py) against
 *   - the known answers in slow_odgi/README.md:144-178 and
 *     flatgfa-sh/README.md:31-36,51-59,267-270 (via stand-in fixtures),
 *   - golden vectors produced by importing the reference's own Python
 *     implementation (slow_odgi/slow_odgi/depth.py:6-16) in the authoring
 *     container (tests/golden/make_golden.py).
 * The Rust reference itself cannot be built here (no cargo/rustc), so there is
 * no oracle/_ref for this path.
 *
 * Inputs are the reference's own packed array-of-struct pools, byte-for-byte:
 *   Path    (flatgfa/src/flatgfa.rs:99-112): 24 B, align 1:
 *             name{start,end}@0, steps{start,end}@8, overlaps{start,end}@16
 *   Segment (flatgfa/src/flatgfa.rs:71-82): 24 B, align 1:
 *             name:u64@0, seq{start,end}@8, optional{start,end}@16
 *   Handle  (flatgfa/src/flatgfa.rs:186-209): u32, segment = h>>1, orient = h&1
 * Outputs mirror Vec<usize> (u64) and Vec<f64>.
 */
#define _POSIX_C_SOURCE 200809L /* pthread barriers (all-cores baseline only) */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PATH_STRIDE 24
#define SEG_STRIDE 24

static inline uint32_t rd32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4); /* pools are align-1 (repr(packed)); never cast */
    return v;
}

static inline uint64_t rd64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

/* Segment::len(), flatgfa.rs:86-88 + pool.rs:105-107 */
static inline uint64_t seg_len(const uint8_t *segs, uint64_t id) {
    const uint8_t *s = segs + id * SEG_STRIDE;
    return (uint64_t)(rd32(s + 12) - rd32(s + 8));
}

/*
 * seg_depth_with_uniq, ops/depth.rs:15-39.
 * One reusable bitset, cleared per path (depth.rs:23,26), test/set per step
 * (depth.rs:30,33).  Returns 0, or -1 when a span or a segment id is out of
 * bounds (where the reference would panic on the slice index).
 */
int oracle_seg_depth_with_uniq(const uint8_t *paths, uint64_t n_paths,
                               const uint32_t *steps, uint64_t n_steps,
                               uint64_t n_segs, uint64_t *depths,
                               uint64_t *uniq_depths) {
    uint64_t words = (n_segs + 63) / 64;
    uint64_t *seen = (uint64_t *)malloc((words ? words : 1) * sizeof(uint64_t));
    if (!seen) return -2;
    memset(depths, 0, n_segs * sizeof(uint64_t));
    memset(uniq_depths, 0, n_segs * sizeof(uint64_t));
    for (uint64_t p = 0; p < n_paths; ++p) {
        const uint8_t *path = paths + p * PATH_STRIDE;
        uint64_t start = rd32(path + 8), end = rd32(path + 12);
        if (start > end || end > n_steps) { free(seen); return -1; }
        memset(seen, 0, words * sizeof(uint64_t)); /* seen.clear() */
        for (uint64_t i = start; i < end; ++i) {
            uint32_t h;
            memcpy(&h, (const uint8_t *)steps + i * 4, 4);
            uint64_t seg_id = h >> 1; /* Handle::segment() */
            if (seg_id >= n_segs) { free(seen); return -1; }
            depths[seg_id] += 1;
            uint64_t bit = 1ull << (seg_id & 63);
            if (!(seen[seg_id >> 6] & bit)) {
                uniq_depths[seg_id] += 1;
                seen[seg_id >> 6] |= bit;
            }
        }
    }
    free(seen);
    return 0;
}

/* The same loop, path-parallel on `n_threads` host threads (the "all host
 * cores" baseline).  Not in the reference -- its loop is single-threaded -- and never the
 * checker: every thread runs depth.rs:25-36 over the paths p = t, t + T, ... into private
 * 32-bit vectors with its own `seen` bitset; after a barrier every thread adds up one slice of
 * the segments over all the private vectors. */
typedef struct mt_job {
    const uint8_t *paths; uint64_t n_paths; const uint32_t *steps; uint64_t n_steps, n_segs;
    uint32_t *depths, *uniq; uint64_t t, T; int rc;
    struct mt_job *all; pthread_barrier_t *bar; uint64_t *out_d, *out_u;
} mt_job;
static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    uint64_t words = (j->n_segs + 63) / 64;
    uint64_t *seen = (uint64_t *)malloc((words ? words : 1) * sizeof(uint64_t));
    if (!seen) j->rc = -2;
    for (uint64_t p = j->t; !j->rc && p < j->n_paths; p += j->T) {
        const uint8_t *path = j->paths + p * PATH_STRIDE;
        uint64_t start = rd32(path + 8), end = rd32(path + 12);
        if (start > end || end > j->n_steps) { j->rc = -1; break; }
        memset(seen, 0, words * sizeof(uint64_t));
        for (uint64_t i = start; i < end; ++i) {
            uint32_t h;
            memcpy(&h, (const uint8_t *)j->steps + i * 4, 4);
            uint64_t seg_id = h >> 1;
            if (seg_id >= j->n_segs) { j->rc = -1; break; }
            j->depths[seg_id] += 1;
            uint64_t bit = 1ull << (seg_id & 63);
            if (!(seen[seg_id >> 6] & bit)) {
                j->uniq[seg_id] += 1;
                seen[seg_id >> 6] |= bit;
            }
        }
    }
    free(seen);
    pthread_barrier_wait(j->bar);
    uint64_t lo = j->n_segs * j->t / j->T, hi = j->n_segs * (j->t + 1) / j->T;
    for (uint64_t s = lo; s < hi; ++s) {
        uint64_t d = 0, u = 0;
        for (uint64_t k = 0; k < j->T; ++k) { d += j->all[k].depths[s]; u += j->all[k].uniq[s]; }
        j->out_d[s] = d;
        j->out_u[s] = u;
    }
    return NULL;
}
int oracle_seg_depth_with_uniq_mt(const uint8_t *paths, uint64_t n_paths, const uint32_t *steps,
                                  uint64_t n_steps, uint64_t n_segs, uint64_t *depths,
                                  uint64_t *uniq_depths, uint32_t n_threads) {
    if (n_threads == 0 || n_threads > 1024) return -3;
    mt_job *jobs = (mt_job *)calloc(n_threads, sizeof(mt_job));
    pthread_t *tid = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
    pthread_barrier_t bar;
    if (!jobs || !tid) { free(jobs); free(tid); return -2; }
    int rc = 0;
    for (uint32_t t = 0; t < n_threads; ++t) {
        mt_job *j = &jobs[t];
        j->paths = paths; j->n_paths = n_paths; j->steps = steps; j->n_steps = n_steps; j->n_segs = n_segs;
        j->t = t; j->T = n_threads; j->all = jobs; j->bar = &bar; j->out_d = depths; j->out_u = uniq_depths;
        j->depths = (uint32_t *)calloc(n_segs ? n_segs : 1, 4);
        j->uniq = (uint32_t *)calloc(n_segs ? n_segs : 1, 4);
        if (!j->depths || !j->uniq) rc = -2;
    }
    if (!rc) {
        pthread_barrier_init(&bar, NULL, n_threads);
        uint32_t started = 0;
        for (; started < n_threads; ++started)
            if (pthread_create(&tid[started], NULL, mt_worker, &jobs[started])) break;
        if (started < n_threads) {
            /* cannot run short-handed (the barrier counts n_threads): give the missing ones a turn inline is
             * not possible either, so report the failure after the started ones are released */
            rc = -2;
            for (uint32_t t = started; t < n_threads; ++t) pthread_create(&tid[t], NULL, mt_worker, &jobs[t]);
        }
        for (uint32_t t = 0; t < n_threads; ++t) pthread_join(tid[t], NULL);
        pthread_barrier_destroy(&bar);
        for (uint32_t t = 0; t < n_threads; ++t)
            if (!rc && jobs[t].rc) rc = jobs[t].rc;
    }
    for (uint32_t t = 0; t < n_threads; ++t) { free(jobs[t].depths); free(jobs[t].uniq); }
    free(jobs); free(tid);
    return rc;
}

/* seg_depth, ops/depth.rs:45-56 */
int oracle_seg_depth(const uint8_t *paths, uint64_t n_paths,
                     const uint32_t *steps, uint64_t n_steps, uint64_t n_segs,
                     uint64_t *depths) {
    memset(depths, 0, n_segs * sizeof(uint64_t));
    for (uint64_t p = 0; p < n_paths; ++p) {
        const uint8_t *path = paths + p * PATH_STRIDE;
        uint64_t start = rd32(path + 8), end = rd32(path + 12);
        if (start > end || end > n_steps) return -1;
        for (uint64_t i = start; i < end; ++i) {
            uint32_t h;
            memcpy(&h, (const uint8_t *)steps + i * 4, 4);
            uint64_t seg_id = h >> 1;
            if (seg_id >= n_segs) return -1;
            depths[seg_id] += 1;
        }
    }
    return 0;
}

/*
 * path_depth, ops/depth.rs:88-111 with measure_path, depth.rs:116-131.
 * Pass 1 over ALL paths; pass 2 over the requested path ids in the order
 * given.  usize sums wrap like release-mode Rust; the single f64 division is
 * (depth as f64) / (length as f64) -- 0/0 gives NaN exactly as in Rust.
 */
int oracle_path_depth(const uint8_t *paths, uint64_t n_paths,
                      const uint32_t *steps, uint64_t n_steps,
                      const uint8_t *segs, uint64_t n_segs,
                      const uint32_t *path_ids, uint64_t n_ids,
                      uint64_t *path_lengths, double *path_depths) {
    uint64_t *seg_depths = (uint64_t *)malloc((n_segs ? n_segs : 1) * sizeof(uint64_t));
    if (!seg_depths) return -2;
    int rc = oracle_seg_depth(paths, n_paths, steps, n_steps, n_segs, seg_depths);
    if (rc) { free(seg_depths); return rc; }
    for (uint64_t k = 0; k < n_ids; ++k) {
        if (path_ids[k] >= n_paths) { free(seg_depths); return -1; }
        const uint8_t *path = paths + (uint64_t)path_ids[k] * PATH_STRIDE;
        uint64_t start = rd32(path + 8), end = rd32(path + 12);
        uint64_t depth = 0, length = 0;
        for (uint64_t i = start; i < end; ++i) {
            uint32_t h;
            memcpy(&h, (const uint8_t *)steps + i * 4, 4);
            uint64_t seg_id = h >> 1;
            uint64_t len = seg_len(segs, seg_id);
            depth += seg_depths[seg_id] * len;
            length += len;
        }
        path_lengths[k] = length;
        path_depths[k] = (double)depth / (double)length;
    }
    free(seg_depths);
    return 0;
}

/*
 * format_float, ops/depth.rs:192-197: format!("{:.digits$}") then strip
 * trailing '0's, then strip trailing '.'s.  glibc's %.*f is correctly rounded
 * on the exact binary value (round-half-even), as Rust's is.  Rust spells the
 * non-finite values "NaN", "inf", "-inf"; trim_end_matches leaves them alone.
 * Returns the number of bytes written (no NUL counted).
 */
int oracle_format_float(double x, int digits, char *out, int cap) {
    char buf[512];
    int n;
    if (x != x) {
        n = snprintf(buf, sizeof buf, "NaN");
    } else if (x == 1.0 / 0.0) {
        n = snprintf(buf, sizeof buf, "inf");
    } else if (x == -1.0 / 0.0) {
        n = snprintf(buf, sizeof buf, "-inf");
    } else {
        n = snprintf(buf, sizeof buf, "%.*f", digits, x);
    }
    while (n > 0 && buf[n - 1] == '0') --n;
    while (n > 0 && buf[n - 1] == '.') --n;
    if (n > cap) n = cap;
    memcpy(out, buf, (size_t)n);
    return n;
}

/* A growable byte buffer for the emitters. */
typedef struct { char *p; size_t n, cap; } obuf;

static int ob_put(obuf *b, const char *s, size_t n) {
    if (b->n + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        while (nc < b->n + n) nc *= 2;
        char *np = (char *)realloc(b->p, nc);
        if (!np) return -1;
        b->p = np;
        b->cap = nc;
    }
    memcpy(b->p + b->n, s, n);
    b->n += n;
    return 0;
}

/*
 * SegDepth::emit, ops/depth.rs:67-82: header, then one line per segment in
 * pool order: "{seg.name as u32}\t{depth}\t{uniq}\n".
 * Returns a malloc'd buffer (caller frees with oracle_free) and its length.
 */
char *oracle_emit_seg_depth(const uint8_t *segs, uint64_t n_segs,
                            const uint64_t *depths, const uint64_t *uniq,
                            uint64_t *out_len) {
    obuf b = {0, 0, 0};
    const char *hdr = "#node.id\tdepth\tdepth.uniq\n";
    ob_put(&b, hdr, strlen(hdr));
    char line[96];
    for (uint64_t i = 0; i < n_segs; ++i) {
        uint32_t name = (uint32_t)rd64(segs + i * SEG_STRIDE); /* as u32 */
        int n = snprintf(line, sizeof line, "%u\t%llu\t%llu\n", name,
                         (unsigned long long)depths[i],
                         (unsigned long long)uniq[i]);
        ob_put(&b, line, (size_t)n);
    }
    *out_len = b.n;
    return b.p ? b.p : (char *)calloc(1, 1);
}

/*
 * PathDepth::emit, ops/depth.rs:143-160: header, then per requested path
 * "{name}\t0\t{length}\t{format_float(depth, 2)}\n".
 */
char *oracle_emit_path_depth(const uint8_t *paths, const uint8_t *name_data,
                             const uint32_t *path_ids, uint64_t n_ids,
                             const uint64_t *lengths, const double *depths,
                             uint64_t *out_len) {
    obuf b = {0, 0, 0};
    const char *hdr = "#path\tstart\tend\tmean.depth\n";
    ob_put(&b, hdr, strlen(hdr));
    char num[600];
    for (uint64_t k = 0; k < n_ids; ++k) {
        const uint8_t *path = paths + (uint64_t)path_ids[k] * PATH_STRIDE;
        uint32_t ns = rd32(path), ne = rd32(path + 4);
        ob_put(&b, (const char *)name_data + ns, ne - ns);
        int n = snprintf(num, sizeof num, "\t0\t%llu\t", (unsigned long long)lengths[k]);
        ob_put(&b, num, (size_t)n);
        n = oracle_format_float(depths[k], 2, num, (int)sizeof num);
        ob_put(&b, num, (size_t)n);
        ob_put(&b, "\n", 1);
    }
    *out_len = b.n;
    return b.p ? b.p : (char *)calloc(1, 1);
}

void oracle_free(void *p) { free(p); }
