// `fgfa` -- the slice of the reference CLI that sits on the depth path
// (cucapra/pollen flatgfa/src/cli/main.rs:9-55, cmds.rs:16-97,217-285), built on the C ABI:
//
//   fgfa [-i FILE.flatgfa | -I FILE.gfa] [-o OUT.flatgfa] [-O OUT.gfa] [COMMAND]
//   COMMAND: toc [-b] | paths | stats -S | depth [-d] [-r NAME]... [-b FILE.bed] [-s PATHS]
//            | window-depth PATH SIZE | overlap --paths FILE
//
// With no -i/-I the GFA text is read from stdin; with no COMMAND the graph is written out
// (-o binary, -O text, otherwise text on stdout).  `depth` output is byte-identical to the
// reference's and is computed on the GPU.  Everything else in the reference CLI is out of scope.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/flatgfa.h"

static inline const char *test_hook(const char *name) { return getenv(name); }  // (a test hook, not a user-facing switch: device_common.hpp)

static int die(const char *what) {
    fprintf(stderr, "fgfa: %s: %s\n", what, flatgfa_last_error());
    return 1;
}

// One name per line (blank lines skipped), as slow_odgi's parse_paths does.
static bool read_names(const char *file, std::vector<std::string> *out) {
    FILE *f = fopen(file, "rb");
    if (!f) return false;
    std::string cur;
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (c == '\n') {
            if (!cur.empty()) out->push_back(cur);
            cur.clear();
        } else if (c != '\r') {
            cur.push_back((char)c);
        }
    }
    if (!cur.empty()) out->push_back(cur);
    fclose(f);
    return true;
}

static void write_all(const char *p, size_t n) {
    while (n) {
        ssize_t w = write(STDOUT_FILENO, p, n);
        if (w <= 0) exit(1);
        p += w;
        n -= (size_t)w;
    }
}

int main(int argc, char **argv) {
    const char *in_flat = nullptr, *in_gfa = nullptr, *out_flat = nullptr, *out_gfa = nullptr, *prealloc = nullptr;
    bool mutate = false;  // -m: with -o, write the preallocated container (cli/main.rs:88-96); with -i, open for mutation (read the same way here)
    int i = 1;
    for (; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char **dst) {
            if (i + 1 >= argc) { fprintf(stderr, "fgfa: %s needs a value\n", a.c_str()); exit(2); }
            *dst = argv[++i];
        };
        if (a == "-i") need(&in_flat);
        else if (a == "-I") need(&in_gfa);
        else if (a == "-o") need(&out_flat);
        else if (a == "-O") need(&out_gfa);
        else if (a == "-m") mutate = true;
        else if (a == "-p") need(&prealloc);
        else break;
    }
    std::string cmd = i < argc ? argv[i++] : "";

    // The special case the reference's main starts with (cli/main.rs:61-66): `-m -o OUT` with no command and
    // no -i is prealloc_translate -- GFA text (the mapped -I file, or stdin) parsed straight into the mapped,
    // preallocated output; no graph is built.
    if (mutate && cmd.empty() && !in_flat && out_flat) {
        const uint32_t factor = prealloc ? (uint32_t)strtoul(prealloc, nullptr, 10) : 32u;
        int rc;
        if (in_gfa) {
            const int fd = open(in_gfa, O_RDONLY);
            struct stat sb;
            if (fd < 0 || fstat(fd, &sb) != 0) { fprintf(stderr, "fgfa: cannot open %s\n", in_gfa); return 1; }
            void *m = sb.st_size ? mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
            close(fd);
            if (m == MAP_FAILED) { fprintf(stderr, "fgfa: cannot map %s\n", in_gfa); return 1; }
            rc = flatgfa_translate_prealloc((const uint8_t *)m, (size_t)sb.st_size, 0, out_flat, factor);
            if (m) munmap(m, (size_t)sb.st_size);
        } else {
            std::string buf;
            char tmp[1 << 16];
            ssize_t r;
            while ((r = read(STDIN_FILENO, tmp, sizeof tmp)) > 0) buf.append(tmp, (size_t)r);
            rc = flatgfa_translate_prealloc((const uint8_t *)buf.data(), buf.size(), 1, out_flat, factor);
        }
        return rc ? die("write") : 0;
    }

    // A one-shot query is mostly start-up: the HIP runtime takes a tenth of a second and more to come
    // up in a cold process, the staging buffers, the first copy and the first launch another thirty
    // milliseconds.  All of that starts now, on a thread of its own, while this one maps or parses
    // the graph (and, for a mapped file, has the kernel map the step pool's pages in).
    const bool wants_device = cmd == "depth" || cmd == "window-depth" || cmd == "overlap";
    // (`fgfa` only ever uses device 0: on a node with several GPUs the runtime need not bring the others up.  Set
    // before the first HIP call; a caller's or scheduler's own choice of visible devices -- by any of the variables the
    // HIP runtime honours: CUDA_VISIBLE_DEVICES and GPU_DEVICE_ORDINAL index into what ROCr exposes, so narrowing
    // ROCr to device 0 under them would hide the assigned GPU -- is left alone.)
    bool pinned = false;
    for (const char *v : {"ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"}) pinned = pinned || getenv(v) != nullptr;
    if (wants_device && !pinned) (void)setenv("ROCR_VISIBLE_DEVICES", "0", 0);
    // (freed host memory stays with the process: an unmapped buffer costs the next launch or copy 10-30 ms on this driver, flatgfa.h)
    if (const char *k = getenv("FLATGFA_KEEP_HOST_MEMORY"); !(k && k[0] == '0')) (void)flatgfa_keep_host_memory(1);
    std::thread warm;
    if (wants_device && !getenv("FLATGFA_NO_WARM")) warm = std::thread([] { (void)flatgfa_warm_device(0); });

    flatgfa_t g;
    if (in_flat) {
        g = flatgfa_load(in_flat);
    } else if (in_gfa) {
        g = flatgfa_parse(in_gfa);
    } else {
        std::string buf;
        char tmp[1 << 16];
        ssize_t r;
        while ((r = read(STDIN_FILENO, tmp, sizeof tmp)) > 0) buf.append(tmp, (size_t)r);
        g = flatgfa_parse_stream_bytes((const uint8_t *)buf.data(), buf.size());
    }
    if (!g) {
        if (warm.joinable()) warm.join();
        return die("cannot load graph");
    }
#ifdef MADV_POPULATE_READ
    if (wants_device && in_flat) {  // (a freshly mapped file: one call instead of a fault per page inside the upload's copies)
        const void *steps = nullptr;
        uint64_t n = 0, es = 0;
        if (flatgfa_pool(g, 4, &steps, &n, &es) == 0 && n) {
            const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE), a0 = (uintptr_t)steps & ~(page - 1);
            (void)madvise((void *)a0, ((uintptr_t)steps + n * es) - a0, MADV_POPULATE_READ);
        }
    }
#endif
    if (warm.joinable()) warm.join();

    int rc = 0;
    if (cmd.empty()) {
        if (out_flat) {
            if (flatgfa_write_flatgfa(g, out_flat)) rc = die("write");
        } else {
            char *text = nullptr;
            size_t n = 0;
            if (flatgfa_print_gfa(g, &text, &n)) {
                rc = die("print");
            } else if (out_gfa) {
                FILE *f = fopen(out_gfa, "wb");
                if (!f || fwrite(text, 1, n, f) != n || fclose(f)) { fprintf(stderr, "fgfa: cannot write %s\n", out_gfa); rc = 1; }
            } else {
                write_all(text, n);
            }
            flatgfa_free_text(text);
        }
    } else if (cmd == "toc") {
        bool bytes = i < argc && !strcmp(argv[i], "-b");
        static const char *names[11] = {"header", "segs", "paths", "links", "steps", "seq_data",
                                        "overlaps", "alignment", "name_data", "optional_data", "line_order"};
        for (int k = 0; k < 11; ++k) {
            uint64_t len = 0, es = 0;
            flatgfa_pool(g, k, nullptr, &len, &es);
            printf("%s: %llu\n", names[k], (unsigned long long)(bytes ? len * es : len));
        }
    } else if (cmd == "paths") {
        uint32_t n = flatgfa_path_count(g);
        for (uint32_t k = 0; k < n; ++k) {
            flatgfa_string_t s = flatgfa_get_path_name(g, k);
            printf("%.*s\n", s.len, (const char *)s.data);
        }
    } else if (cmd == "stats") {
        if (i < argc && !strcmp(argv[i], "-S")) {
            uint64_t len[11];
            for (int k = 0; k < 11; ++k) flatgfa_pool(g, k, nullptr, &len[k], nullptr);
            printf("#length\tnodes\tedges\tpaths\tsteps\n%llu\t%llu\t%llu\t%llu\t%llu\n", (unsigned long long)len[5],
                   (unsigned long long)len[1], (unsigned long long)len[3], (unsigned long long)len[2],
                   (unsigned long long)len[4]);
        }
    } else if (cmd == "depth") {
        bool seg_depth = false;
        std::vector<std::string> names;
        const char *bed = nullptr, *subset = nullptr;
        for (; i < argc; ++i) {
            std::string a = argv[i];
            if (a == "-d" || a == "--graph-depth-table") seg_depth = true;
            else if (a == "-r" && i + 1 < argc) names.push_back(argv[++i]);
            else if ((a == "-b" || a == "--bed-input") && i + 1 < argc) bed = argv[++i];
            else if ((a == "-s" || a == "--subset-paths") && i + 1 < argc) subset = argv[++i];  // odgi depth -d -s
            else { fprintf(stderr, "fgfa: depth: unknown option %s\n", a.c_str()); flatgfa_free(g); return 2; }
        }
        char *text = nullptr;
        size_t n = 0;
        if (seg_depth && subset) {
            // node depth over the listed paths only (slow_odgi depth --paths, depth.py:12)
            std::vector<std::string> want;
            if (!read_names(subset, &want)) { fprintf(stderr, "fgfa: cannot read %s\n", subset); flatgfa_free(g); return 1; }
            std::vector<uint32_t> ids;
            for (auto &nm : want) {
                int64_t id = flatgfa_find_path(g, (const uint8_t *)nm.data(), nm.size());
                if (id >= 0) ids.push_back((uint32_t)id);
            }
            uint32_t S = flatgfa_get_segment_count(g);
            std::vector<uint64_t> d(S), u(S);
            uint32_t dummy = 0;
            rc = flatgfa_seg_depth_subset(g, ids.empty() ? &dummy : ids.data(), (uint32_t)ids.size(), d.data(), u.data());
            if (!rc) {
                std::string out = "#node.id\tdepth\tdepth.uniq\n";
                const void *segs;
                uint64_t ns;
                flatgfa_pool(g, 1, &segs, &ns, nullptr);
                char line[96];
                for (uint32_t k = 0; k < S; ++k) {
                    uint64_t name;
                    memcpy(&name, (const char *)segs + (size_t)k * 24, 8);
                    out.append(line, (size_t)snprintf(line, sizeof line, "%u\t%llu\t%llu\n", (uint32_t)name,
                                                      (unsigned long long)d[k], (unsigned long long)u[k]));
                }
                write_all(out.data(), out.size());
            }
        } else if (seg_depth) {
            rc = flatgfa_depth_table(g, &text, &n);
        } else if (bed) {
            std::string btext;
            FILE *bf = fopen(bed, "rb");
            if (!bf) { fprintf(stderr, "fgfa: cannot open %s\n", bed); flatgfa_free(g); return 1; }
            char tmp[1 << 16];
            size_t r;
            while ((r = fread(tmp, 1, sizeof tmp, bf)) > 0) btext.append(tmp, r);
            fclose(bf);
            rc = flatgfa_bed_depth_table(g, (const uint8_t *)btext.data(), btext.size(), &text, &n);
        } else if (names.empty()) {
            rc = flatgfa_path_depth_table(g, nullptr, 0, &text, &n);
        } else {
            // cmds.rs:270-274: names that do not resolve are silently dropped
            std::vector<uint32_t> ids;
            for (auto &nm : names) {
                int64_t id = flatgfa_find_path(g, (const uint8_t *)nm.data(), nm.size());
                if (id >= 0) ids.push_back((uint32_t)id);
            }
            uint32_t dummy = 0;
            rc = flatgfa_path_depth_table(g, ids.empty() ? &dummy : ids.data(), (uint32_t)ids.size(), &text, &n);
        }
        if (rc) rc = die("depth");
        else write_all(text, n);
        flatgfa_free_text(text);
    } else if (cmd == "window-depth") {
        // cli/cmds.rs:477-496: fgfa window-depth PATH SIZE
        if (i + 1 >= argc) { fprintf(stderr, "usage: fgfa window-depth PATH SIZE\n"); flatgfa_free(g); return 2; }
        std::string pname = argv[i];
        int64_t id = flatgfa_find_path(g, (const uint8_t *)pname.data(), pname.size());
        char *text = nullptr;
        size_t n = 0;
        if (id < 0) { fprintf(stderr, "fgfa: path not found\n"); rc = 1; }
        else if (flatgfa_window_depth_table(g, (uint32_t)id, strtoull(argv[i + 1], nullptr, 10), &text, &n)) rc = die("window-depth");
        else write_all(text, n);
        flatgfa_free_text(text);
    } else if (cmd == "overlap") {
        // slow_odgi overlap --paths FILE (slow_odgi/__main__.py:93-101)
        if (i + 1 >= argc || strcmp(argv[i], "--paths")) { fprintf(stderr, "usage: fgfa overlap --paths FILE\n"); flatgfa_free(g); return 2; }
        std::vector<std::string> want;
        if (!read_names(argv[i + 1], &want)) { fprintf(stderr, "fgfa: cannot read %s\n", argv[i + 1]); flatgfa_free(g); return 1; }
        std::vector<uint32_t> ids;
        for (auto &nm : want) {
            int64_t id = flatgfa_find_path(g, (const uint8_t *)nm.data(), nm.size());
            if (id < 0) { fprintf(stderr, "fgfa: overlap: path %s is not in the graph\n", nm.c_str()); flatgfa_free(g); return 1; }
            ids.push_back((uint32_t)id);
        }
        char *text = nullptr;
        size_t n = 0;
        uint32_t dummy = 0;
        if (flatgfa_overlap_table(g, ids.empty() ? &dummy : ids.data(), (uint32_t)ids.size(), &text, &n)) rc = die("overlap");
        else write_all(text, n);
        flatgfa_free_text(text);
    } else {
        fprintf(stderr, "fgfa: command '%s' is outside the depth path this build covers\n", cmd.c_str());
        rc = 2;
    }
    // (no tear-down: the process is over, and unloading the HIP runtime takes longer than the query did)
    fflush(stdout);
    fflush(stderr);
    if (!test_hook("FLATGFA_SLOW_EXIT")) _exit(rc);
    flatgfa_free(g);
    return rc;
}
