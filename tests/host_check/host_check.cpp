// host_check -- TEST INFRASTRUCTURE: drives the host side of the product (pollen_amd/csrc/flatgfa_core.cpp,
// synth.cpp: the GFA parser with its threaded step-list parse, the .flatgfa view / dump / prealloc
// container, the GFA printer, the BED parser and the interval walk, the table emitters with their
// formatter threads) so that it can run under AddressSanitizer + UBSan and under ThreadSanitizer
// (`make -C pollen_amd/csrc asan tsan`; tests/test_host_sanitized.py).  The reference gets this
// from safe Rust (its only `unsafe` is the mmap, flatgfa/src/memfile.rs:9,21,30); this C++ reads
// unaligned, attacker-shaped images (file.rs:163-213) and must show it the hard way.
//
//   host_check FILE.gfa ...     prints one digest line per phase; the digests of a sanitized build must
//                               equal those of the plain build (the test compares them).
// No depth is computed here (the product has no CPU depth path): the emitters get made-up counts.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../pollen_amd/csrc/flatgfa_core.hpp"

using namespace fgfa;

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}
static uint64_t rng(uint64_t &s) {  // splitmix64
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static bool slurp(const char *path, std::string *out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, r);
    fclose(f);
    return true;
}

// Everything that can be done with one parsed graph; returns a digest of all the bytes produced.
static uint64_t exercise(const Store &st, uint64_t h) {
    const View v = st.view();
    std::string err, text;
    if (print_gfa(v, &text, &err)) h = fnv(h, text.data(), text.size());
    else h = fnv(h, err.data(), err.size());
    // .flatgfa round trip (file.rs:290-313 -> 185-213), from an image at an odd address: the pools of a file are unaligned
    const size_t n = flatgfa_file_size(v);
    std::vector<uint8_t> img(n + 3);
    dump_flatgfa(v, img.data() + 3);
    View back;
    if (view_flatgfa(img.data() + 3, n, &back, &err)) {
        std::string text2;
        if (print_gfa(back, &text2, &err)) h = fnv(h, text2.data(), text2.size());
        h = fnv(h, &n, sizeof n);
        validate_spans(back, &err);
        h = fnv(h, "v", validate_step_ids(back) ? 1 : 0);
        // the tables, with counts that are a function of the segment index
        std::vector<uint64_t> d(back.segs.len), u(back.segs.len);
        for (size_t i = 0; i < d.size(); ++i) d[i] = i * 7 % 13, u[i] = i % 5;
        std::string tab;
        emit_seg_depth(back, d.data(), u.data(), &tab);
        h = fnv(h, tab.data(), tab.size());
        std::vector<uint32_t> ids(back.paths.len);
        std::vector<uint64_t> lens(back.paths.len);
        std::vector<double> means(back.paths.len);
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = (uint32_t)i, lens[i] = 100 + i, means[i] = (double)(i * 37 % 101) / 8.0;
        tab.clear();
        emit_path_depth(back, ids.data(), ids.size(), lens.data(), means.data(), &tab);
        h = fnv(h, tab.data(), tab.size());
        // windows over the first path (window_depth.rs:22-57, 84-147), when the step ids are sound
        if (back.paths.len && validate_spans(back, &err) && validate_step_ids(back)) {
            const uint64_t plen = path_length(back, 0);
            Bed bed;
            const uint8_t nm[1] = {'w'};
            make_windows(nm, 1, 0, plen, plen / 3 + 1, &bed);
            std::vector<double> out(bed.entries.size());
            interval_depth(back, d.data(), 0, bed.entries.data(), bed.entries.size(), out.data());
            std::string itab;
            emit_interval_depth(bed, out.data(), &itab);
            h = fnv(h, itab.data(), itab.size());
        }
    } else {
        h = fnv(h, err.data(), err.size());
    }
    // the preallocated container (file.rs:117-158, parse.rs:176-216), estimated from the printed text
    uint64_t cap[11];
    if (!text.empty() && estimate_toc((const uint8_t *)text.data(), text.size(), cap, &err)) {
        size_t total = 0;
        if (prealloc_file_size(v, cap, &total, &err) && total < (64u << 20)) {
            std::vector<uint8_t> pre(total + 1);
            dump_flatgfa_prealloc(v, cap, pre.data() + 1);
            View pv;
            if (view_flatgfa(pre.data() + 1, total, &pv, &err)) h = fnv(h, &pv.steps.len, sizeof pv.steps.len);
        }
    }
    return h;
}

int main(int argc, char **argv) {
    uint64_t all = 0xcbf29ce484222325ull;
    std::vector<std::string> texts;
    for (int i = 1; i < argc; ++i) {
        std::string t;
        if (!slurp(argv[i], &t)) {
            fprintf(stderr, "host_check: cannot read %s\n", argv[i]);
            return 2;
        }
        texts.push_back(t);
    }
    // 1. the fixtures themselves, both parser modes
    uint64_t h = 0xcbf29ce484222325ull;
    for (const std::string &t : texts)
        for (int stream = 0; stream < 2; ++stream) {
            Store st;
            std::string err;
            if (parse_gfa((const uint8_t *)t.data(), t.size(), &st, &err, stream != 0)) h = exercise(st, h);
            else h = fnv(h, err.data(), err.size());
        }
    printf("fixtures %016llx\n", (unsigned long long)h);
    all = fnv(all, &h, 8);
    // 2. mutated texts: flipped bytes, cut tails, doubled lines, spliced fixtures -- the parser must reject or
    //    accept them, never read or write out of bounds
    h = 0xcbf29ce484222325ull;
    uint64_t seed = 42;
    size_t accepted = 0, rejected = 0, pre_fit = 0, pre_full = 0;
    for (int round = 0; round < 600 && !texts.empty(); ++round) {
        std::string t = texts[rng(seed) % texts.size()];
        if (t.size() > 20000) t.resize(20000 + rng(seed) % 64);
        const int kind = (int)(rng(seed) % 5);
        if (kind == 0 && !t.empty()) {
            for (int k = 0; k < 3; ++k) t[rng(seed) % t.size()] = (char)rng(seed);
        } else if (kind == 1 && !t.empty()) {
            t.resize(rng(seed) % t.size());
        } else if (kind == 2 && !t.empty()) {
            const size_t a = rng(seed) % t.size();
            t.insert(a, t.substr(a, rng(seed) % 200));
        } else if (kind == 3) {
            const std::string &o = texts[rng(seed) % texts.size()];
            t += o.substr(0, std::min<size_t>(o.size(), rng(seed) % 3000));
        } else if (!t.empty()) {
            static const char pool[] = "SPLH\t\n+-,*0123456789ACGTMIDN";
            for (int k = 0; k < 8; ++k) t[rng(seed) % t.size()] = pool[rng(seed) % (sizeof pool - 1)];
        }
        Store st;
        std::string err;
        const bool heap_ok = parse_gfa((const uint8_t *)t.data(), t.size(), &st, &err, (round & 1) != 0);
        if (heap_ok) {
            ++accepted;
            h = exercise(st, h);
        } else {
            ++rejected;
            h = fnv(h, err.data(), err.size());
        }
        // the same text parsed straight into a preallocated image (file.rs:255-272), one byte off any alignment: the
        // image a heap parse + dump_flatgfa_prealloc leaves, or the push that does not fit, or the same parse error
        uint64_t cap[11];
        std::string perr;
        size_t total = 0;
        if (round % 3 == 0) guess_toc(1 + rng(seed) % 3, cap);
        else if (!estimate_toc((const uint8_t *)t.data(), t.size(), cap, &perr)) continue;
        if (!toc_file_size(cap, &total, &perr) || total > (64u << 20)) continue;
        std::vector<uint8_t> img(total + 1);
        const bool pre_ok = parse_gfa_prealloc((const uint8_t *)t.data(), t.size(), (round & 1) != 0, cap, img.data() + 1, &perr);
        if (pre_ok) {
            ++pre_fit;
            size_t t2 = 0;
            std::vector<uint8_t> want(total + 1);
            if (!heap_ok || !prealloc_file_size(st.view(), cap, &t2, &perr) || t2 != total) {
                fprintf(stderr, "host_check: round %d: the preallocated parse accepted what the heap parse did not\n", round);
                return 3;
            }
            dump_flatgfa_prealloc(st.view(), cap, want.data() + 1);
            if (memcmp(want.data() + 1, img.data() + 1, total) != 0) {
                fprintf(stderr, "host_check: round %d: the preallocated parse left a different image\n", round);
                return 3;
            }
        } else {
            const bool capacity = perr.rfind("preallocated flatgfa:", 0) == 0;
            pre_full += capacity ? 1 : 0;
            if (!capacity && (heap_ok || perr != err)) {
                fprintf(stderr, "host_check: round %d: heap parse %s, preallocated parse says '%s'\n", round, heap_ok ? "ok" : err.c_str(), perr.c_str());
                return 3;
            }
        }
    }
    printf("mutated_texts %016llx accepted=%zu rejected=%zu preallocated: %zu fit, %zu over capacity\n", (unsigned long long)h, accepted, rejected, pre_fit, pre_full);
    all = fnv(all, &h, 8);
    // 3. damaged .flatgfa images: the table of contents and the spans are attacker-shaped (file.rs:185-213)
    h = 0xcbf29ce484222325ull;
    size_t img_ok = 0, img_bad = 0;
    if (!texts.empty()) {
        size_t pick = 0;  // the largest fixture below 100 KB: most of its image is pools, not table of contents
        for (size_t i = 0; i < texts.size(); ++i)
            if (texts[i].size() < 100000 && (texts[pick].size() >= 100000 || texts[i].size() > texts[pick].size())) pick = i;
        Store st;
        std::string err;
        if (parse_gfa((const uint8_t *)texts[pick].data(), texts[pick].size(), &st, &err)) {
            const View v = st.view();
            const size_t n = flatgfa_file_size(v);
            std::vector<uint8_t> good(n);
            dump_flatgfa(v, good.data());
            for (int round = 0; round < 400; ++round) {
                std::vector<uint8_t> img = good;
                const int kind = (int)(rng(seed) % 4);
                if (kind == 0) {  // a length or capacity of the table of contents
                    const size_t at = 8 + 8 * (rng(seed) % 22);
                    uint64_t val = rng(seed) >> (rng(seed) % 64);
                    memcpy(img.data() + at, &val, 8);
                } else if (kind == 1) {
                    img.resize(rng(seed) % (n + 1));
                } else if (kind == 2 && n > 184) {  // a path or segment record
                    for (int k = 0; k < 4; ++k) img[184 + rng(seed) % (n - 184)] = (uint8_t)rng(seed);
                } else {
                    img[rng(seed) % 8] ^= 1;  // the magic
                }
                View back;
                if (view_flatgfa(img.data(), img.size(), &back, &err)) {
                    ++img_ok;
                    std::string text;
                    const bool sound = validate_spans(back, &err) && validate_step_ids(back);
                    h = fnv(h, "s", sound ? 1 : 0);
                    if (sound && print_gfa(back, &text, &err)) h = fnv(h, text.data(), text.size());
                } else {
                    ++img_bad;
                    h = fnv(h, err.data(), err.size());
                }
            }
        }
    }
    printf("damaged_images %016llx viewed=%zu refused=%zu\n", (unsigned long long)h, img_ok, img_bad);
    all = fnv(all, &h, 8);
    // 4. BED text and number formatting
    h = 0xcbf29ce484222325ull;
    {
        const char *beds[] = {"p0\t0\t10\np1\t5\t6\n", "p0\t3\n", "", "x\t1\t2\textra\n", "p0\t18446744073709551615\t0\n", "\n\n"};
        for (const char *b : beds) {
            Bed bed;
            std::string err;
            const bool ok = parse_bed((const uint8_t *)b, strlen(b), &bed, &err);
            h = fnv(h, ok ? "1" : "0", 1);
            h = fnv(h, err.data(), err.size());
            h = fnv(h, bed.name_data.data(), bed.name_data.size());
        }
        const double xs[] = {0.0, 1.9, 2.0, 0.125, 0.375, 2.675, 1e300, -0.0, 1.0 / 3.0, 0.0 / 0.0, 1.0 / 0.0, -1.0 / 0.0, 0.005, 0.015, 1234567.891};
        for (double x : xs)
            for (int dg = 0; dg < 4; ++dg) {
                const std::string s = format_float(x, dg);
                h = fnv(h, s.data(), s.size());
            }
    }
    printf("bed_and_floats %016llx\n", (unsigned long long)h);
    all = fnv(all, &h, 8);
    // 5. the threaded paths: a synthetic graph's text (step lists long enough for the parallel parse) parsed back,
    //    and a table wide enough for every formatter thread
    h = 0xcbf29ce484222325ull;
    {
        Store st;
        synth_store(7, 60000, 40, 6000, 0, true, &st);
        std::string text, err;
        if (print_gfa(st.view(), &text, &err)) {
            Store back;
            if (parse_gfa((const uint8_t *)text.data(), text.size(), &back, &err)) {
                const bool same = back.steps.size() == st.steps.size() && !memcmp(back.steps.data(), st.steps.data(), st.steps.size() * sizeof(Handle)) &&
                                  back.paths.size() == st.paths.size();
                h = fnv(h, same ? "same" : "DIFF", 4);
                std::vector<uint64_t> d(back.segs.size()), u(back.segs.size());
                for (size_t i = 0; i < d.size(); ++i) d[i] = i * 2654435761ull % 100000, u[i] = i % 977;
                std::string tab;
                emit_seg_depth(back.view(), d.data(), u.data(), &tab);
                h = fnv(h, tab.data(), tab.size());
            } else {
                h = fnv(h, err.data(), err.size());
            }
        }
    }
    {   // the table straight from 32-bit counts (flatgfa_depth_table), enough segments for all of its threads: the same bytes
        Store big;
        synth_store(9, 70000, 4, 100, 0, false, &big);
        std::vector<uint32_t> d32(big.segs.size()), u32(big.segs.size());
        std::vector<uint64_t> d(big.segs.size()), u(big.segs.size());
        for (size_t i = 0; i < d.size(); ++i) d[i] = d32[i] = (uint32_t)(i * 2654435761ull % 4000000000ull), u[i] = u32[i] = (uint32_t)(i % 977);
        std::string tab;
        emit_seg_depth(big.view(), d.data(), u.data(), &tab);
        size_t len = 0;
        char *m = emit_seg_depth_u32_malloc(big.view(), d32.data(), u32.data(), &len);
        const bool same = m && len == tab.size() && !memcmp(m, tab.data(), len) && m[len] == 0;
        free(m);
        if (!same) { fprintf(stderr, "emit_seg_depth_u32_malloc differs from emit_seg_depth\n"); return 1; }
        h = fnv(h, tab.data(), tab.size());
    }
    printf("threads %016llx\n", (unsigned long long)h);
    all = fnv(all, &h, 8);
    printf("all %016llx\n", (unsigned long long)all);
    return 0;
}
