/*
 * flatgfa.h -- C ABI of the MI355X-native FlatGFA depth engine (libflatgfa.so).
 *
 * Part 1 is a drop-in for the reference's `flatgfa-c` crate (cucapra/pollen,
 * flatgfa-c/src/lib.rs; its header is cbindgen-generated, flatgfa-c/build.rs:6-12):
 * identical type names, function names, signatures, ownership and in-band error
 * sentinels.  Part 2 is additive: the reference exposes its depth queries only
 * as Rust functions (flatgfa/src/ops/depth.rs), so the entry points a binding
 * for them would need are declared here, each citing the Rust item it replaces.
 * Part 3 is the device-level surface used when the caller already owns HBM
 * buffers (e.g. a torch tensor's data_ptr) and a HIP stream.
 *
 * All depth entry points run on the GPU through hand-written HIP kernels for
 * gfx950.  There is no CPU fallback: without a usable HIP device they return
 * FLATGFA_ERR_NO_DEVICE.
 */
#ifndef FLATGFA_H
#define FLATGFA_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libflatgfa.so is built with -fvisibility=hidden; only this header's functions are exported. */
#pragma GCC visibility push(default)

/* ------------------------------------------------------------------------ */
/* Part 1 -- the flatgfa-c surface                                          */
/* ------------------------------------------------------------------------ */

/* flatgfa-c/src/lib.rs:16,31 -- opaque store, handed out as a raw pointer. */
typedef struct CStore CStore;
typedef CStore *flatgfa_t;

/* flatgfa-c/src/lib.rs:36-40 -- borrowed, NOT NUL-terminated. */
typedef struct flatgfa_string_t {
    const uint8_t *data;
    int len;
} flatgfa_string_t;

/* flatgfa-c/src/lib.rs:140-144 */
typedef struct flatgfa_handle_t {
    uint32_t segment_id;
    bool is_forward;
} flatgfa_handle_t;

/* lib.rs:63 -- parse a GFA text file.  Where the reference aborts (missing or
 * malformed file) this returns NULL and sets flatgfa_last_error(). */
flatgfa_t flatgfa_parse(const char *filename);
/* lib.rs:72 -- NULL-safe; also releases any device buffers the handle owns. */
void flatgfa_free(flatgfa_t gfa);
/* lib.rs:80 */
uint32_t flatgfa_get_segment_count(flatgfa_t gfa);
/* lib.rs:92 -- {NULL,0} when segment_id is out of range. */
flatgfa_string_t flatgfa_get_seq(flatgfa_t gfa, uint32_t segment_id);
/* lib.rs:105 */
uint32_t flatgfa_path_count(flatgfa_t gfa);
/* lib.rs:118 -- {NULL,0} when path_index is out of range. */
flatgfa_string_t flatgfa_get_path_name(flatgfa_t gfa, uint32_t path_index);
/* lib.rs:130 -- UINT32_MAX when path_index is out of range. */
uint32_t flatgfa_get_path_step_count(flatgfa_t gfa, uint32_t path_index);
/* lib.rs:149-154 -- false when either index is out of range. */
bool flatgfa_get_step(flatgfa_t gfa, uintptr_t path_index, uintptr_t step_index, flatgfa_handle_t *out);

/* ------------------------------------------------------------------------ */
/* Part 2 -- additive: loaders, writers, and the depth queries              */
/* ------------------------------------------------------------------------ */

enum {
    FLATGFA_OK = 0,
    FLATGFA_ERR_ARG = -1,       /* NULL handle / bad argument */
    FLATGFA_ERR_BOUNDS = -2,    /* a span or segment id is out of range (the reference panics) */
    FLATGFA_ERR_NO_DEVICE = -3, /* no usable HIP device; there is no CPU fallback */
    FLATGFA_ERR_HIP = -4,       /* a HIP runtime call failed; see flatgfa_last_error() */
    FLATGFA_ERR_IO = -5,
    FLATGFA_ERR_TOO_LARGE = -6, /* more than 2^32-1 steps */
    FLATGFA_ERR_PARSE = -7,     /* GFA text the reference's parser panics on (flatgfa_translate_prealloc; the parse calls that return a handle return NULL) */
    FLATGFA_ERR_STALE_PLAN = -8 /* FLATGFA_CHECK_NO_CLAIM=1 only: the step values changed behind a device plan (flatgfa_dev_plan_steps_changed) */
};

/* Thread-local description of the last failure in this thread ("" if none). */
const char *flatgfa_last_error(void);

/* Parser::parse_mem on a caller buffer (flatgfa/src/parse.rs:77; flatgfa-py/src/lib.rs parse_bytes). */
flatgfa_t flatgfa_parse_bytes(const uint8_t *data, size_t len);
/* Parser::parse_stream semantics (parse.rs:24-74; what `fgfa` uses for stdin, cli/main.rs:110-113):
 * an unterminated last line is kept and links are added before paths. */
flatgfa_t flatgfa_parse_stream_bytes(const uint8_t *data, size_t len);
/* file::view on a memory-mapped `.flatgfa` file, zero-copy (flatgfa/src/file.rs:185; cli/main.rs:99-100). */
flatgfa_t flatgfa_load(const char *flatgfa_filename);
/* file::dump (flatgfa/src/file.rs:290; cli/main.rs:197-201). */
int flatgfa_write_flatgfa(flatgfa_t gfa, const char *filename);
/* The preallocated ("in-place") container of `fgfa -m -p FACTOR -o OUT [-I GFA]`
 * (cli/main.rs:216-248, file.rs:255-272): every pool's region is `capacity` items long, `len` of
 * them in use.  With the GFA text the graph was parsed from, the capacities are the reference's
 * estimates from it (parse.rs:176-216, file.rs:136-158); with gfa_text == NULL, file.rs:117-132's
 * guess from `factor`.  FLATGFA_ERR_BOUNDS where a pool does not fit its capacity (the reference's
 * fixed-capacity store panics there).  flatgfa_load reads such files as it reads any other. */
int flatgfa_write_flatgfa_prealloc(flatgfa_t gfa, const char *filename, const uint8_t *gfa_text, size_t text_len, uint32_t factor);
/* prealloc_translate itself (cli/main.rs:216-248): GFA text -> preallocated container with no graph
 * in between.  The output file is created at the size its capacities add up to and mapped
 * (memfile::map_new_file, memfile.rs:24-33), file::init (file.rs:255-272) writes the empty table of
 * contents, the parser pushes straight into the file's regions (Parser::for_slice, parse.rs:170-174)
 * and the table's lengths are set at the end (Toc::for_fixed_store).  from_stream == 0: the text of
 * `-I GFA` (capacities estimated from it, Parser::parse_mem); != 0: the text came from stdin
 * (capacities guessed from `factor`, Parser::parse_stream).  Byte for byte the file that
 * flatgfa_parse_bytes + flatgfa_write_flatgfa_prealloc leave.  FLATGFA_ERR_BOUNDS where a pool
 * does not fit (the reference panics with the file as it then is: capacities in the table, every
 * length 0 -- so it is left here), FLATGFA_ERR_PARSE where the text does not parse. */
int flatgfa_translate_prealloc(const uint8_t *gfa_text, size_t text_len, int from_stream, const char *filename, uint32_t factor);
/* GFA text (flatgfa/src/print.rs:99-153).  *text is malloc'd; release with flatgfa_free_text. */
int flatgfa_print_gfa(flatgfa_t gfa, char **text, size_t *len);
void flatgfa_free_text(char *text);
/* Deterministic synthetic graph; model 0 = pangenome walk, 1 = uniform,
 * 2 = chromosome (paths walk along the graph, every other one downwards; one step in a hundred
 * jumps anywhere), 3 = haplotype (the same without those jumps: one step in 1600 skips up to 1087
 * segments), 4 = repeats (a haplotype walk that, one step in 6400, goes 16 .. 271 segments back and
 * walks them again). */
flatgfa_t flatgfa_synth(uint64_t seed, uint32_t n_segs, uint32_t n_paths, uint32_t steps_per_path, int model,
                        bool with_seq);

/* Raw pool access in `.flatgfa` order (file.rs:14-27): 0 header, 1 segs, 2 paths, 3 links,
 * 4 steps, 5 seq_data, 6 overlaps, 7 alignment, 8 name_data, 9 optional_data, 10 line_order.
 * *data borrows from the handle and may be unaligned (the reference's PODs are repr(packed)). */
int flatgfa_pool(flatgfa_t gfa, int pool_index, const void **data, uint64_t *len, uint64_t *elem_size);
/* FlatGFA::find_path (flatgfa.rs:387): first path with this name, or -1. */
int64_t flatgfa_find_path(flatgfa_t gfa, const uint8_t *name, size_t len);

/* Number of visible HIP devices (0 if none). */
int flatgfa_device_count(void);
/* Start the HIP runtime on `device` ahead of need: the runtime's own start-up (a tenth of a second
 * and more on a cold process), the pinned staging buffers of flatgfa_to_device, the first copy and
 * the first kernel launch of the process.  Meant to be called from a helper thread while the caller
 * maps or parses its graph (`fgfa` does); everything it does would otherwise happen inside the
 * first flatgfa_to_device.  Returns 0, or FLATGFA_ERR_NO_DEVICE / FLATGFA_ERR_HIP. */
int flatgfa_warm_device(int device);
/* Host memory policy of the PROCESS (glibc's malloc; a no-op elsewhere): with `on`, freed host memory stays with the process --
 * nothing is unmapped, the heap is not trimmed -- instead of going back to the system.  Why a GPU library has this: on the
 * amdgpu/KFD driver a process that unmaps host memory has its GPU queues quiesced and restored by a delayed work item, and
 * the next kernel launch or copy then waits 10-30 ms (in steps of the kernel's timer tick; a second now and then) where it
 * would take 0.1 -- measured: a first query 20 ms instead of 0.9, a plan over a million paths 49 ms instead of 12
 * (profiles/NOTES.md R6.6c).  The library keeps its own large temporaries; what the host application frees around its
 * queries -- a parsed GFA, result vectors, a table -- is the application's to keep, or this switch's.  Not applied by the
 * library itself: `fgfa` and the Python package call it at start-up (FLATGFA_KEEP_HOST_MEMORY=0 in the environment keeps
 * them from it).  Returns 0. */
int flatgfa_keep_host_memory(int on);
/* Copy the graph's structure-of-arrays image (steps, path spans, segment lengths) into the
 * HBM of `device` and keep it resident until flatgfa_free.  Depth calls do this lazily on
 * device 0 if it has not been done. */
int flatgfa_to_device(flatgfa_t gfa, int device);
/* What that took on this handle, in milliseconds of host time: the host-to-device copies (SoA
 * conversion and allocation included) and the creation of the depth plan (scratch, item lists,
 * the timing of kernel variants).  An error before the graph is resident. */
int flatgfa_residency_ms(flatgfa_t gfa, double *h2d_ms, double *plan_ms);

/* seg_depth_with_uniq (ops/depth.rs:15-39) when uniq_out != NULL, seg_depth (depth.rs:45-56)
 * when it is NULL.  Outputs are indexed by segment id, one uint64_t (Rust usize) each. */
int flatgfa_seg_depth(flatgfa_t gfa, uint64_t *depth_out, uint64_t *uniq_out);
/* path_depth + measure_path (ops/depth.rs:88-131) for the given path ids, in order. */
int flatgfa_path_depth(flatgfa_t gfa, const uint32_t *path_ids, uint32_t n_ids, uint64_t *length_out,
                       double *mean_depth_out);
/* The bytes `fgfa depth -d` prints: SegDepth::emit (ops/depth.rs:67-82; cli/cmds.rs:237-245). */
int flatgfa_depth_table(flatgfa_t gfa, char **text, size_t *len);
/* The bytes `fgfa depth [-r NAME]...` prints: PathDepth::emit (ops/depth.rs:143-160;
 * cli/cmds.rs:256-284).  path_ids == NULL means all paths, in order. */
int flatgfa_path_depth_table(flatgfa_t gfa, const uint32_t *path_ids, uint32_t n_ids, char **text, size_t *len);
/* PathDepth::as_bed (ops/depth.rs:173-183): the same paths as a BED store -- one entry
 * {path name, 0, path length in base pairs} each, the depths dropped -- rendered as three-column
 * BED text (`name\t0\tlength\n`).  path_ids == NULL means all paths, in order. */
int flatgfa_path_depth_bed(flatgfa_t gfa, const uint32_t *path_ids, uint32_t n_ids, char **text, size_t *len);
/* odgi-style node depth restricted to a subset of paths (`odgi depth -d -s FILE`,
 * slow_odgi/slow_odgi/depth.py:12; tests/turnt.toml:31-44).  The reference's Rust `-d` ignores
 * `-r`; this closes that gap.  path_ids may repeat: each occurrence counts as its own path. */
int flatgfa_seg_depth_subset(flatgfa_t gfa, const uint32_t *path_ids, uint32_t n_ids, uint64_t *depth_out,
                             uint64_t *uniq_out);
/* Path-pair overlap (slow_odgi/slow_odgi/overlap.py:6-32; `odgi overlap -R FILE`):
 * touch_out[k * path_count + j] = 1 iff path j touches query path query_ids[k]. */
int flatgfa_path_overlaps(flatgfa_t gfa, const uint32_t *query_ids, uint32_t n_q, uint8_t *touch_out);
/* The bytes `slow_odgi overlap --paths FILE` prints for these query paths (overlap.py:17-32). */
int flatgfa_overlap_table(flatgfa_t gfa, const uint32_t *query_ids, uint32_t n_q, char **text, size_t *len);
/* interval_depth (flatgfa/src/ops/window_depth.rs:176-180): mean depth of each [start,end) interval
 * (base pairs along path `path_index`, sorted) -- node depth from the GPU, the f64 accumulation of
 * assign_depths (:116-147) on the host in the reference's order. */
int flatgfa_interval_depth(flatgfa_t gfa, uint32_t path_index, const uint64_t *starts, const uint64_t *ends,
                           uint64_t n_intervals, double *depth_out);
/* The bytes `fgfa window-depth PATH SIZE` prints (window_depth.rs:183-200, cli/cmds.rs:488-496). */
int flatgfa_window_depth_table(flatgfa_t gfa, uint32_t path_index, uint64_t window, char **text, size_t *len);
/* The bytes `fgfa depth -b FILE.bed` prints (window_depth.rs:203-211, cli/cmds.rs:246-255); the BED
 * text is parsed as flatbed.rs:125-158 does. */
int flatgfa_bed_depth_table(flatgfa_t gfa, const uint8_t *bed, size_t bed_len, char **text, size_t *len);
/* format_float (ops/depth.rs:192-197); returns bytes written (no NUL). */
int flatgfa_format_float(double x, int digits, char *out, int cap);

/* ---- one graph sharded over the GPUs of a node by ONE process ----
 * The reference has no counterpart (it is single-threaded, flatgfa/src/ops/depth.rs:15-39); the
 * semantics are those of the functions above, the sharding is invisible in the results.
 * The graph's steps are cut into n_shards contiguous stretches of near-equal size -- at path
 * boundaries where one lies within an eighth of a shard's share of the even cut, inside a path
 * where none does (fewer paths than shards; a path longer than a shard's share) -- and shard i is
 * made resident on HIP device devices[i] (NULL: shard i on device i mod the device count), with a
 * depth plan, a stream and a host thread of its own.  A query is the local HIP kernels on every
 * shard, ONE all-reduce of the fused [depth | uniq | per-cut-path touch] u32 vector over RCCL
 * (ncclAllReduce, ncclUint32, ncclSum; communicators from ncclCommInitAll), and a fix-up of
 * unique depth for the paths that were cut (depth.rs:30-34 counts a path once per segment however
 * many of its pieces touch it).  librccl.so is loaded on the first create with more than one
 * shard.  Devices may repeat: shards that share a device exchange by device-side adds instead (so
 * does FLATGFA_SHARD_NO_RCCL).  The handle borrows `gfa`, which must outlive it.
 * flags: */
enum {
    FLATGFA_SHARD_WHOLE_PATHS = 1, /* never cut inside a path (shards may then be uneven, or empty) */
    FLATGFA_SHARD_NO_RCCL = 2      /* exchange by peer copies and device-side adds */
};
typedef struct flatgfa_sharded flatgfa_sharded_t;
flatgfa_sharded_t *flatgfa_sharded_create(flatgfa_t gfa, const int *devices, int n_shards, unsigned flags);
void flatgfa_sharded_free(flatgfa_sharded_t *sh);
/* What shard `shard` holds (any out pointer may be NULL): its device, its stretch of the steps
 * pool, the first path it has steps of, how many paths or pieces of paths it walks, how many paths
 * the whole handle cuts, and whether the exchange is RCCL's. */
int flatgfa_sharded_layout(flatgfa_sharded_t *sh, int shard, int *device, uint64_t *step_begin, uint64_t *step_end,
                           uint32_t *first_path, uint32_t *n_pieces, uint32_t *n_split_paths, int *uses_rccl);
/* Bytes every shard contributes to the one collective of a call: 4 per segment for node depth alone;
 * with unique depth 8, plus 4 per segment for every 32 / bits(n_shards) paths that were cut (their
 * touch counters share words: with eight shards, 12 bytes per segment whatever was cut). */
uint64_t flatgfa_sharded_collective_bytes(flatgfa_sharded_t *sh, int with_uniq);
/* seg_depth_with_uniq (depth.rs:15-39) / seg_depth (:45-56) over all shards: results as
 * flatgfa_seg_depth's. */
int flatgfa_sharded_seg_depth(flatgfa_sharded_t *sh, uint64_t *depth_out, uint64_t *uniq_out);
/* path_depth (depth.rs:88-131) for the given path ids: every shard measures its paths (and
 * pieces) against the reduced node depth; the host adds the pieces of a cut path up and divides. */
int flatgfa_sharded_path_depth(flatgfa_sharded_t *sh, const uint32_t *path_ids, uint32_t n_ids, uint64_t *length_out,
                               double *mean_depth_out);
/* The same query in three steps, for callers that time it or overlap it with other work:
 * enqueue (local kernels + collective on every shard's stream; returns without waiting), sync
 * (waits for every shard; FLATGFA_ERR_BOUNDS as flatgfa_dev_status), fetch (the reduced vectors
 * as held by shard `shard` -- every shard holds them). */
int flatgfa_sharded_enqueue(flatgfa_sharded_t *sh, int with_uniq);
int flatgfa_sharded_sync(flatgfa_sharded_t *sh);
int flatgfa_sharded_fetch(flatgfa_sharded_t *sh, int shard, uint64_t *depth_out, uint64_t *uniq_out);
/* How many ranks the handle's exchange really spans: every shard contributes a one to an all-reduce over the
 * handle's communicator (RCCL), or -- shards that exchange by adds -- to the same copies and adds its vectors go
 * through, and every shard reads back its own copy of the sum (FLATGFA_ERR_HIP if they disagree).  Equals
 * n_shards on a handle that works; negative = an error code. */
int flatgfa_sharded_ranks_seen(flatgfa_sharded_t *sh);
/* Where flatgfa_sharded_create would cut a graph whose paths, in path order, have path_steps[p] steps (host only: no
 * device is touched).  cuts_out[r], r = 0 .. n_shards, counts path steps along the path order: shard r walks
 * [cuts_out[r], cuts_out[r + 1]).  Cut r lies at the path boundary nearest to the even cut (total * r / n_shards) when
 * that is within an eighth of a shard's share of it -- or always, with FLATGFA_SHARD_WHOLE_PATHS -- and inside the path
 * otherwise.  This is flatgfa_sharded_create's rule for a graph whose paths' step spans lie in path order in the steps pool
 * (every path's steps behind those of the path before it: what the parser emits, flatgfa/src/parse.rs:149-159); where they do
 * not -- the types allow arbitrary spans -- flatgfa_sharded_create cuts at path boundaries only, as with
 * FLATGFA_SHARD_WHOLE_PATHS. */
int flatgfa_shard_cuts(const uint64_t *path_steps, uint32_t n_paths, int n_shards, unsigned flags, uint64_t *cuts_out);

/* ------------------------------------------------------------------------ */
/* Part 3 -- device-level entry points (caller-owned HBM buffers)           */
/* ------------------------------------------------------------------------ */

/* The graph image the kernels read, all pointers in device memory:
 *   steps       u32[n_steps]   Handle bits, (segment << 1) | orient   (flatgfa.rs:186-209)
 *   path_begin  u32[n_paths]   Path.steps.start                        (flatgfa.rs:106)
 *   path_end    u32[n_paths]   Path.steps.end
 *   seg_len     u32[n_segs]    Segment::len() = seq.end - seq.start    (flatgfa.rs:86-88); may be
 *                              NULL for node depth                                              */
typedef struct flatgfa_dev_graph_t {
    const uint32_t *steps;
    uint64_t n_steps;
    const uint32_t *path_begin;
    const uint32_t *path_end;
    uint32_t n_paths;
    uint32_t n_segs;
    const uint32_t *seg_len;
} flatgfa_dev_graph_t;

/* A prepared depth query over one resident graph image: owns the launch plan (how paths are cut
 * into work items) and the scratch HBM the kernels need, on the device that is current when it
 * is created.  `host_path_begin/host_path_end` are host copies of the span arrays (P entries each);
 * pass NULL to have them copied back from the device.  Returns NULL on failure
 * (flatgfa_last_error()); spans that are reversed or exceed n_steps are rejected here, where the
 * reference would panic on the slice index (pool.rs:341-347).  The plan is laid out for the path
 * spans and the step values it was created with (how many runs each path has decides which
 * kernel walks it; a path that walks the segment ids strictly one way is counted without the
 * per-path "seen" set, and so are the stretches of a path in windows it enters once and walks one way,
 * flatgfa_dev_plan_describe: no_claim_items / no_claim_paths / no_claim_chunks): neither may change while it
 * lives.  Every call still checks the step values against n_segs and its scratch against what the
 * steps need, and reports an error -- or completes the call through the simple kernels -- rather
 * than a wrong answer when they no longer fit; what it does not re-check per step is that a path
 * the plan found strictly monotone still is (a compare per step: 12-18 % of the step scan on a
 * graph of such paths, profiles/NOTES.md R5.8), that a no-claim mark still holds, or that the
 * reversed copy of a short downward path still mirrors it.  A caller that DOES change step values
 * under a live plan says so with flatgfa_dev_plan_steps_changed (below), which makes the plan
 * again from the steps as they are; FLATGFA_CHECK_NO_CLAIM=1 in the environment (read when a plan
 * is made; a debugging aid, three more reads of the steps per call) re-derives those facts before
 * every call and has flatgfa_dev_status return FLATGFA_ERR_STALE_PLAN where one no longer holds.
 * Creation runs the query a few times into scratch outputs: once to size the record buckets for
 * this graph (so that no later call runs out of room), and, up to 8 M steps, to time the bucketed
 * kernels against the atomic ones and keep the faster.  A graph beyond 16 M segments is walked in
 * ranges of at most 16 M (one pass over the steps per range and call).  What is not needed for the
 * first answer -- the per-block no-claim marks, three more reads of the steps -- is made on a
 * side stream by the first call behind the plan's creation and used from the first call after it
 * is there (flatgfa_dev_plan_describe waits for it; a caller that only wants the first answer never
 * pays for it).
 * Calls on one plan must not overlap in time. */
typedef struct flatgfa_dev_plan flatgfa_dev_plan_t;
flatgfa_dev_plan_t *flatgfa_dev_plan_create(const flatgfa_dev_graph_t *g, const uint32_t *host_path_begin,
                                            const uint32_t *host_path_end);
/* The same, and the first answer with it: the reference's consumers ask ONE depth query per graph
 * (flatgfa/src/cli/cmds.rs:234-285, flatgfa-sh/src/eval/instr.rs:27-72; bench/config.toml:29-32 times
 * a process per query), and the query that sizes the plan's scratch is a whole query -- here it
 * writes the caller's buffers.  depth_out u32[n_segs] (required) and uniq_out u32[n_segs] (or
 * NULL: node depth alone) are device memory and hold seg_depth_with_uniq (ops/depth.rs:15-39) of
 * the graph when the function returns (the device is synchronized).  *first_status (may be NULL):
 * FLATGFA_OK, or FLATGFA_ERR_BOUNDS when a step named a segment id out of range (the counts are
 * then those of the other steps, as after flatgfa_dev_seg_depth + flatgfa_dev_status). */
flatgfa_dev_plan_t *flatgfa_dev_plan_create_first(const flatgfa_dev_graph_t *g, const uint32_t *host_path_begin,
                                                  const uint32_t *host_path_end, uint32_t *depth_out, uint32_t *uniq_out,
                                                  int *first_status);
void flatgfa_dev_plan_destroy(flatgfa_dev_plan_t *plan);
/* A plan that goes leaves its record buckets (the one large device allocation it had: half a gigabyte at cfg-L) with the library for
 * the next plan of the device to take, because allocating and freeing gigabytes of device memory in a row is what the driver does worst
 * (a plan made beside five that stay: 5 ms, or every other time 80-300 ms; with the array kept 4.8 ms every time).  One array per
 * device, the largest seen, at most 8 GB.  This gives them back to the device; call it when no plan will be made for a while. */
void flatgfa_dev_release_scratch(void);
/* The step values (not the spans) of the plan's graph image were changed by the caller: waits for
 * `stream` (the plan's), then makes the plan's launch plan and scratch again from the steps as they
 * are -- which kernel walks which path, the reversed copies, which paths and blocks need no claim,
 * the record buckets' sizes, the overlap query's bitmaps -- at the cost of creating it.  The
 * reference's pools are immutable while a query runs (flatgfa/src/ops/depth.rs:30-34 reads them
 * as they are); this is how a caller who owns the HBM buffers keeps that true of a plan. */
int flatgfa_dev_plan_steps_changed(flatgfa_dev_plan_t *plan, void *stream);

/* Node depth on device: seg_depth_with_uniq (ops/depth.rs:15-39) when uniq_out != NULL, seg_depth
 * (depth.rs:45-56) when NULL.  depth_out / uniq_out are u32[n_segs] in device memory.  Enqueues on
 * `stream` (a hipStream_t; NULL = the default stream) and returns without synchronizing. */
int flatgfa_dev_seg_depth(flatgfa_dev_plan_t *plan, uint32_t *depth_out, uint32_t *uniq_out, void *stream);
/* Per-path sums on device given depth u32[n_segs] (measure_path, ops/depth.rs:116-131):
 * length_out[k] = sum seg_len, weighted_out[k] = sum depth*seg_len over the steps of path
 * path_ids[k] (u32[n_ids], device memory), both u64 wrapping like Rust usize.  The single f64
 * division per path is left to the host. */
int flatgfa_dev_path_sums(flatgfa_dev_plan_t *plan, const uint32_t *path_ids, uint32_t n_ids, const uint32_t *depth,
                          uint64_t *length_out, uint64_t *weighted_out, void *stream);
/* path_depth for ALL paths of the plan in one go (ops/depth.rs:88-111 with every path requested --
 * what `fgfa depth` prints, cmds.rs:256-262): node depth into depth_out u32[n_segs], and per path
 * the two integer sums of measure_path into length_out / weighted_out u64[n_paths] (device
 * memory), indexed by path.  The sums of the paths the step-scan kernel walks are formed in the
 * same pass that accumulates node depth -- a run of segments contributes two differences of
 * window-local prefix sums -- so the steps are read once. */
int flatgfa_dev_path_depth_all(flatgfa_dev_plan_t *plan, uint32_t *depth_out, uint64_t *length_out,
                               uint64_t *weighted_out, void *stream);
/* Path-pair overlap on device (slow_odgi/slow_odgi/overlap.py:6-14): touch_out[k * n_paths + j] = 1
 * iff path j is a different path from path_ids[k] and the two share at least one ORIENTED handle.
 * query_ids u32[n_q] and touch_out u8[n_q * n_paths] are device memory.  A coarse bitmap per path
 * (one bit per 2048 handles) is built on the first call and kept with the plan; exact handle
 * bitsets are built for the query paths only, per call -- memory follows the queries, not the
 * number of paths. */
int flatgfa_dev_path_overlaps(flatgfa_dev_plan_t *plan, const uint32_t *query_ids, uint32_t n_q, uint8_t *touch_out,
                              void *stream);
/* Synchronizes `stream`, then returns FLATGFA_OK, or FLATGFA_ERR_BOUNDS if any kernel since the
 * last call saw a segment id >= n_segs or a path id >= n_paths.  Plans size their scratch for the
 * graph when they are created; should a node-depth call nevertheless have run out of scratch room
 * (the step values changed behind the plan's back), this call runs it again on a larger plan
 * before it returns, into the same output buffers -- which therefore must not have been
 * modified in between.  Only the last call can be completed that way: if several node-depth calls
 * were enqueued since the last status and one of them ran out of room, this returns
 * FLATGFA_ERR_HIP (call status after every call where step values may change behind a plan).
 * A plan belongs to ONE stream: every call of the plan, and this one, must be enqueued on the same
 * stream (this call may release and reallocate the plan's scratch once that stream is idle). */
int flatgfa_dev_status(flatgfa_dev_plan_t *plan, void *stream);

/* Calls in flight.  A plan's calls run one after the other on its one stream, and a call is two kernels with
 * opposite needs: pass 1 is bound by the memory system, pass 2 by instruction issue, each wants the whole chip, and
 * each leaves compute units idle at its end.  A pipeline is `calls_in_flight` plans of ONE resident graph (they share
 * the graph image and its claim on the Infinity Cache; each has its own record scratch) on as many internal streams,
 * taken in turn, and each lane's pass 1 runs on fewer persistent workgroups than there are compute units (half of
 * them with three calls in flight or more on graphs of up to 2^28 steps, eleven sixteenths otherwise), so that the
 * other calls' kernels share the chip with it all the time, not only at its tail (1 M segments / 100 M steps: 0.133 ms
 * per call one at a time, 0.114 with two in flight, 0.107 with three; a fourth loses).  A lane's plan alone would be
 * slower than flatgfa_dev_plan_create's.  Every call is a whole query into the caller's buffers,
 * which must not be reused before the call that wrote them is known to be done (join, or status).
 *   flatgfa_dev_pipeline_seg_depth   as flatgfa_dev_seg_depth, on the pipeline's next lane; returns without waiting.
 *                                    The call first waits for everything enqueued so far on `after_stream` (the
 *                                    stream that filled the graph image or consumed the buffers' previous contents;
 *                                    NULL = the default stream), or for nothing when after_stream is (void *)-1.
 *   flatgfa_dev_pipeline_path_depth_all   as flatgfa_dev_path_depth_all (what `fgfa depth` prints), likewise.
 *   flatgfa_dev_pipeline_join        makes `stream` wait for every call enqueued so far (events; no host wait).
 *   flatgfa_dev_pipeline_status      waits for every lane; FLATGFA_ERR_BOUNDS etc. as flatgfa_dev_status. */
typedef struct flatgfa_dev_pipeline flatgfa_dev_pipeline_t;
flatgfa_dev_pipeline_t *flatgfa_dev_pipeline_create(const flatgfa_dev_graph_t *g, const uint32_t *host_path_begin,
                                                    const uint32_t *host_path_end, int calls_in_flight);
void flatgfa_dev_pipeline_destroy(flatgfa_dev_pipeline_t *p);
int flatgfa_dev_pipeline_seg_depth(flatgfa_dev_pipeline_t *p, uint32_t *depth_out, uint32_t *uniq_out, void *after_stream);
int flatgfa_dev_pipeline_path_depth_all(flatgfa_dev_pipeline_t *p, uint32_t *depth_out, uint64_t *length_out, uint64_t *weighted_out,
                                        void *after_stream);
int flatgfa_dev_pipeline_join(flatgfa_dev_pipeline_t *p, void *stream);
int flatgfa_dev_pipeline_status(flatgfa_dev_pipeline_t *p);
/* flatgfa_dev_plan_steps_changed for every lane (waits for all of them first). */
int flatgfa_dev_pipeline_steps_changed(flatgfa_dev_pipeline_t *p);
int flatgfa_dev_pipeline_describe(flatgfa_dev_pipeline_t *p, char *out, int cap);

/* Which kernels this plan's calls run -- the choices made when it was created, some of them by
 * timing on the graph -- as a line of `key=value` words; returns the bytes written (NUL excluded). */
int flatgfa_dev_plan_describe(flatgfa_dev_plan_t *plan, char *out, int cap);

/* Kernel-level timing for bench.py: when enabled, every kernel the library launches is bracketed
 * by HIP events on its own stream; flatgfa_dev_profile_read synchronizes and returns, for up to
 * `cap` kernels since the last read, the kernel name and elapsed milliseconds. */
void flatgfa_dev_profile_enable(int on);
int flatgfa_dev_profile_read(const char **names, float *ms, int cap);
/* What such an event pair reads around a kernel that does nothing, launched with `n_workgroups`
 * workgroups of 1024 threads and `lds_bytes` of dynamic LDS (median of `reps`; < 0 on error):
 * the part of a profiled kernel's time that rocprofv3's dispatch duration does not contain. */
float flatgfa_dev_profile_overhead_ms(int n_workgroups, int lds_bytes, int reps, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* FLATGFA_H */
