/*
 * overlap_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of slow_odgi's path-pair overlap query (cucapra/pollen,
 * slow_odgi/slow_odgi/overlap.py:6-32): path q "touches" query path ip when they are
 * different paths and their SETS OF ORIENTED HANDLES intersect
 * (set(path.segments) holds Handle(name, orientation), mygfa/mygfa/gfa.py:130-138).
 * The reference has no Rust implementation of this query; slow_odgi is the reference.
 * Pinned by the *.overlap.tsv golden vectors (tests/golden/make_golden.py) and by the
 * worked example in slow_odgi/README.md (y = {1+,3-} touches x but not z = {3+,4+}).
 * Also restates the window / interval depth arithmetic of
 * flatgfa/src/ops/window_depth.rs:84-147,176-211 (f64 accumulation in reference order).
 *
 * Same input convention as depth_oracle.c: the reference's packed AoS pools.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PATH_STRIDE 24
#define SEG_STRIDE 24

static inline uint32_t rd32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

/*
 * out[k * n_paths + j] = 1 iff path j touches query path query_ids[k].
 * One bitset of 2*n_segs bits (one per oriented handle) per path.
 */
int oracle_path_touches(const uint8_t *paths, uint64_t n_paths, const uint32_t *steps, uint64_t n_steps,
                        uint64_t n_segs, const uint32_t *query_ids, uint64_t n_q, uint8_t *out) {
    const uint64_t words = (2 * n_segs + 63) / 64;
    uint64_t *bits = (uint64_t *)calloc((size_t)(n_paths ? n_paths : 1) * (words ? words : 1), 8);
    if (!bits) return -2;
    for (uint64_t p = 0; p < n_paths; ++p) {
        const uint8_t *path = paths + p * PATH_STRIDE;
        uint64_t start = rd32(path + 8), end = rd32(path + 12);
        if (start > end || end > n_steps) { free(bits); return -1; }
        for (uint64_t i = start; i < end; ++i) {
            uint32_t h;
            memcpy(&h, (const uint8_t *)steps + i * 4, 4);
            if ((uint64_t)(h >> 1) >= n_segs) { free(bits); return -1; }
            bits[p * words + (h >> 6)] |= 1ull << (h & 63); /* the handle bits ARE (segment, orientation) */
        }
    }
    for (uint64_t k = 0; k < n_q; ++k) {
        uint64_t ip = query_ids[k];
        if (ip >= n_paths) { free(bits); return -1; }
        for (uint64_t j = 0; j < n_paths; ++j) {
            uint8_t t = 0;
            if (j != ip) { /* overlap.py:10-11: a path does not touch itself */
                for (uint64_t w = 0; w < words; ++w)
                    if (bits[ip * words + w] & bits[j * words + w]) { t = 1; break; }
            }
            out[k * n_paths + j] = t;
        }
    }
    free(bits);
    return 0;
}

typedef struct { char *p; size_t n, cap; } obuf;
static void ob_put(obuf *b, const char *s, size_t n) {
    if (b->n + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        while (nc < b->n + n) nc *= 2;
        b->p = (char *)realloc(b->p, nc);
        b->cap = nc;
    }
    memcpy(b->p + b->n, s, n);
    b->n += n;
}

/*
 * overlap.py:17-32: the header is printed once, before the first touching pair; each line is
 * "{ip}\t0\t{len(pathseq[ip])}\t{q}".  path_len[k] = length in base pairs of query path k
 * (mygfa/preprocess.py pathseq: the concatenated sequence of its steps).
 */
char *oracle_emit_overlap(const uint8_t *paths, const uint8_t *name_data, uint64_t n_paths,
                          const uint32_t *query_ids, uint64_t n_q, const uint64_t *path_len, const uint8_t *touch,
                          uint64_t *out_len) {
    obuf b = {0, 0, 0};
    int header = 0;
    char num[64];
    for (uint64_t k = 0; k < n_q; ++k) {
        const uint8_t *ip = paths + (uint64_t)query_ids[k] * PATH_STRIDE;
        for (uint64_t j = 0; j < n_paths; ++j) {
            if (!touch[k * n_paths + j]) continue;
            if (!header) {
                const char *h = "#path\tstart\tend\tpath.touched\n";
                ob_put(&b, h, strlen(h));
                header = 1;
            }
            const uint8_t *q = paths + j * PATH_STRIDE;
            ob_put(&b, (const char *)name_data + rd32(ip), rd32(ip + 4) - rd32(ip));
            int n = snprintf(num, sizeof num, "\t0\t%llu\t", (unsigned long long)path_len[k]);
            ob_put(&b, num, (size_t)n);
            ob_put(&b, (const char *)name_data + rd32(q), rd32(q + 4) - rd32(q));
            ob_put(&b, "\n", 1);
        }
    }
    *out_len = b.n;
    return b.p ? b.p : (char *)calloc(1, 1);
}

/* ------------------------------------------------------------------------------------------
 * Window / interval depth (flatgfa/src/ops/window_depth.rs).
 *
 * weighted_depths (:84-103): per step, range = [pos, pos + seg.len), depth = (seg_depth*len) as f64.
 * assign_depths (:116-147): walk segments and windows together; for each overlapping pair add
 *     (seg.depth * ((end-start) as f64 / (seg_len) as f64)) / (window_len as f64)
 * in exactly this order of operations.  `win_start/win_end` are the intervals (sorted along the
 * path); returns 0 or -1 on out-of-range input.
 */
int oracle_interval_depth(const uint8_t *paths, uint64_t n_paths, const uint32_t *steps, uint64_t n_steps,
                          const uint8_t *segs, uint64_t n_segs, const uint64_t *seg_depths, uint32_t path_id,
                          const uint64_t *win_start, const uint64_t *win_end, uint64_t n_win, double *out) {
    if (path_id >= n_paths) return -1;
    const uint8_t *path = paths + (uint64_t)path_id * PATH_STRIDE;
    uint64_t start = rd32(path + 8), end = rd32(path + 12);
    if (start > end || end > n_steps) return -1;
    for (uint64_t i = 0; i < n_win; ++i) out[i] = 0.0;
    uint64_t cur = 0, pos = 0;
    for (uint64_t i = start; i < end; ++i) {
        uint32_t h;
        memcpy(&h, (const uint8_t *)steps + i * 4, 4);
        uint64_t seg = h >> 1;
        if (seg >= n_segs) return -1;
        const uint8_t *s = segs + seg * SEG_STRIDE;
        uint64_t len = (uint64_t)(rd32(s + 12) - rd32(s + 8));
        uint64_t r0 = pos, r1 = pos + len;
        pos = r1;
        double sdepth = (double)(seg_depths[seg] * len); /* `total as f64`, :99-100 */
        while (cur < n_win) {
            uint64_t w0 = win_start[cur], w1 = win_end[cur];
            uint64_t o0 = w0 > r0 ? w0 : r0, o1 = w1 < r1 ? w1 : r1;
            if (o1 > o0) {
                double amt = (double)(o1 - o0) / (double)(r1 - r0);
                out[cur] += (sdepth * amt) / (double)(w1 - w0);
            }
            if (w1 > r1) break;
            cur += 1;
        }
    }
    return 0;
}
