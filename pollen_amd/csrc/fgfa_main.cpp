// `fgfa` -- the slice of the reference CLI that sits on the depth path
// (cucapra/pollen flatgfa/src/cli/main.rs:9-55, cmds.rs:16-97,217-285), built on the C ABI:
//
//   fgfa [-i FILE.flatgfa | -I FILE.gfa] [-o OUT.flatgfa] [-O OUT.gfa] [COMMAND]
//   COMMAND: toc [-b] | paths | stats -S | depth [-d] [-r NAME]...
//
// With no -i/-I the GFA text is read from stdin; with no COMMAND the graph is written out
// (-o binary, -O text, otherwise text on stdout).  `depth` output is byte-identical to the
// reference's and is computed on the GPU.  Everything else in the reference CLI is out of scope.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/flatgfa.h"

static int die(const char *what) {
    fprintf(stderr, "fgfa: %s: %s\n", what, flatgfa_last_error());
    return 1;
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
    const char *in_flat = nullptr, *in_gfa = nullptr, *out_flat = nullptr, *out_gfa = nullptr;
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
        else if (a == "-m" || a == "-p") { fprintf(stderr, "fgfa: %s (in-place mutation) is out of scope\n", a.c_str()); return 2; }
        else break;
    }
    std::string cmd = i < argc ? argv[i++] : "";

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
    if (!g) return die("cannot load graph");

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
        for (; i < argc; ++i) {
            std::string a = argv[i];
            if (a == "-d" || a == "--graph-depth-table") seg_depth = true;
            else if (a == "-r" && i + 1 < argc) names.push_back(argv[++i]);
            else if (a == "-b" || a == "--bed-input") { fprintf(stderr, "fgfa: depth -b is not built yet\n"); flatgfa_free(g); return 2; }
            else { fprintf(stderr, "fgfa: depth: unknown option %s\n", a.c_str()); flatgfa_free(g); return 2; }
        }
        char *text = nullptr;
        size_t n = 0;
        if (seg_depth) {
            rc = flatgfa_depth_table(g, &text, &n);
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
    } else {
        fprintf(stderr, "fgfa: command '%s' is outside the depth path this build covers\n", cmd.c_str());
        rc = 2;
    }
    flatgfa_free(g);
    return rc;
}
