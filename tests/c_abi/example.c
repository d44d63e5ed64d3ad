/* A C consumer of libflatgfa.so, the way a user of the reference's flatgfa-c would write one
 * (flatgfa-c/example/example.c shows the reference's own): parse a GFA file, walk every path and
 * every step through the C ABI of include/flatgfa.h Part 1, then -- unless "--no-depth" is given --
 * ask for node depth through Part 2, print the `fgfa depth -d` table, and ask again with the
 * graph sharded two ways (flatgfa_sharded_*).
 *
 *   cc -Iinclude tests/c_abi/example.c -Lpollen_amd/lib -lflatgfa -Wl,-rpath,$PWD/pollen_amd/lib -o example
 *   ./example graph.gfa [--no-depth]
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flatgfa.h"

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s file.gfa [--no-depth]\n", argv[0]);
        return 2;
    }
    const int want_depth = !(argc > 2 && strcmp(argv[2], "--no-depth") == 0);
    flatgfa_t g = flatgfa_parse(argv[1]);
    if (!g) {
        fprintf(stderr, "parse failed: %s\n", flatgfa_last_error());
        return 1;
    }
    const uint32_t n_segs = flatgfa_get_segment_count(g), n_paths = flatgfa_path_count(g);
    printf("segments %" PRIu32 " paths %" PRIu32 "\n", n_segs, n_paths);
    for (uint32_t p = 0; p < n_paths; ++p) {
        const flatgfa_string_t name = flatgfa_get_path_name(g, p);
        const uint32_t n_steps = flatgfa_get_path_step_count(g, p);
        printf("path %.*s: %" PRIu32 " steps\n", name.len, (const char *)name.data, n_steps);
        for (uint32_t s = 0; s < n_steps; ++s) { /* bounded by the path's step count */
            flatgfa_handle_t h;
            if (!flatgfa_get_step(g, p, s, &h)) {
                fprintf(stderr, "step %" PRIu32 " of path %" PRIu32 " is out of range\n", s, p);
                return 1;
            }
            const flatgfa_string_t seq = flatgfa_get_seq(g, h.segment_id);
            printf("  %c %.*s\n", h.is_forward ? '+' : '-', seq.len, (const char *)seq.data);
        }
    }
    /* the in-band sentinels of the reference (flatgfa-c/src/lib.rs:96-98, 133-135, 158-165) */
    flatgfa_handle_t h;
    if (flatgfa_get_seq(g, n_segs).data != NULL || flatgfa_get_path_step_count(g, n_paths) != UINT32_MAX ||
        flatgfa_get_step(g, n_paths, 0, &h)) {
        fprintf(stderr, "out-of-range accessors did not return their sentinels\n");
        return 1;
    }
    if (want_depth) {
        uint64_t *depth = calloc(n_segs ? n_segs : 1, sizeof *depth), *uniq = calloc(n_segs ? n_segs : 1, sizeof *uniq);
        const int rc = flatgfa_seg_depth(g, depth, uniq);
        if (rc != FLATGFA_OK) {
            fprintf(stderr, "flatgfa_seg_depth: %d (%s)\n", rc, flatgfa_last_error());
            return rc == FLATGFA_ERR_NO_DEVICE ? 3 : 1;
        }
        char *text = NULL;
        size_t len = 0;
        if (flatgfa_depth_table(g, &text, &len) != FLATGFA_OK) return 1;
        /* the table and the vectors must agree: line k+1 ends in "\t<depth>\t<uniq>\n" */
        fwrite(text, 1, len, stdout);
        uint64_t total = 0;
        for (uint32_t i = 0; i < n_segs; ++i) total += depth[i];
        printf("total depth %" PRIu64 "\n", total);
        flatgfa_free_text(text);
        /* the same query with the graph sharded over two shards by this process (here both on device 0;
         * on a node with several GPUs: flatgfa_sharded_create(g, NULL, n_gpus, 0) and RCCL carries the reduce) */
        const int devices[2] = {0, 0};
        flatgfa_sharded_t *sh = flatgfa_sharded_create(g, devices, 2, 0);
        if (!sh) {
            fprintf(stderr, "flatgfa_sharded_create: %s\n", flatgfa_last_error());
            return 1;
        }
        uint64_t *depth2 = calloc(n_segs ? n_segs : 1, sizeof *depth2), *uniq2 = calloc(n_segs ? n_segs : 1, sizeof *uniq2);
        if (flatgfa_sharded_seg_depth(sh, depth2, uniq2) != FLATGFA_OK) {
            fprintf(stderr, "flatgfa_sharded_seg_depth: %s\n", flatgfa_last_error());
            return 1;
        }
        uint32_t n_cut = 0;
        flatgfa_sharded_layout(sh, 0, NULL, NULL, NULL, NULL, NULL, &n_cut, NULL);
        const int same = memcmp(depth, depth2, n_segs * sizeof *depth) == 0 && memcmp(uniq, uniq2, n_segs * sizeof *uniq) == 0;
        printf("sharded two ways: %s\n", same ? "same vectors" : "DIFFERENT");
        flatgfa_sharded_free(sh);
        free(depth2);
        free(uniq2);
        free(depth);
        free(uniq);
        if (!same) return 1;
    }
    flatgfa_free(g);
    flatgfa_free(NULL); /* null-safe, lib.rs:72-76 */
    return 0;
}
