// Host-side FlatGFA: pools, GFA text parser, .flatgfa container, emitters.
// See flatgfa_core.hpp for the reference citations.
#include "flatgfa_core.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <atomic>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>

namespace fgfa {

const size_t kPoolElemSize[11] = {1, sizeof(Segment), sizeof(Path), sizeof(Link), sizeof(Handle), 1,
                                  sizeof(Span), 4, 1, 1, 1};
const char *const kPoolName[11] = {"header", "segs", "paths", "links", "steps", "seq_data",
                                   "overlaps", "alignment", "name_data", "optional_data", "line_order"};

size_t View::pool_len(int ix) const {
    switch (ix) {
        case pHeader: return header.len;
        case pSegs: return segs.len;
        case pPaths: return paths.len;
        case pLinks: return links.len;
        case pSteps: return steps.len;
        case pSeqData: return seq_data.len;
        case pOverlaps: return overlaps.len;
        case pAlignment: return alignment.len;
        case pNameData: return name_data.len;
        case pOptionalData: return optional_data.len;
        case pLineOrder: return line_order.len;
    }
    return 0;
}

const void *View::pool_data(int ix) const {
    switch (ix) {
        case pHeader: return header.data;
        case pSegs: return segs.data;
        case pPaths: return paths.data;
        case pLinks: return links.data;
        case pSteps: return steps.data;
        case pSeqData: return seq_data.data;
        case pOverlaps: return overlaps.data;
        case pAlignment: return alignment.data;
        case pNameData: return name_data.data;
        case pOptionalData: return optional_data.data;
        case pLineOrder: return line_order.data;
    }
    return nullptr;
}

int64_t View::find_path(const uint8_t *name, size_t n) const {
    for (size_t i = 0; i < paths.len; ++i) {
        Span s = paths[i].name;
        if (s.len() == n && (n == 0 || memcmp(name_data.data + s.start, name, n) == 0)) return (int64_t)i;
    }
    return -1;
}

template <class T>
static Pool<T> pool_of(const std::vector<T> &v) {
    Pool<T> p;
    p.data = v.data();
    p.len = v.size();
    return p;
}

View Store::view() const {
    View v;
    v.header = pool_of(header);
    v.segs = pool_of(segs);
    v.paths = pool_of(paths);
    v.links = pool_of(links);
    v.steps = pool_of(steps);
    v.seq_data = pool_of(seq_data);
    v.overlaps = pool_of(overlaps);
    v.alignment = {reinterpret_cast<const AlignOp *>(alignment.data()), alignment.size()};
    v.name_data = pool_of(name_data);
    v.optional_data = pool_of(optional_data);
    v.line_order = pool_of(line_order);
    return v;
}

// ---------------------------------------------------------------- NameMap ---

void NameMap::insert(uint64_t name, uint32_t id) {
    // `name - 1` wraps for name == 0, as release-mode Rust does (namemap.rs:20).
    uint64_t nm1 = name - 1;
    if (nm1 == sequential_max_ && nm1 == id) {
        sequential_max_ += 1;
    } else {
        others_[name] = id;
    }
}

bool NameMap::get(uint64_t name, uint32_t *id) const {
    if (name <= sequential_max_) {
        *id = (uint32_t)(name - 1);
        return true;
    }
    auto it = others_.find(name);
    if (it == others_.end()) return false;
    *id = it->second;
    return true;
}

// ----------------------------------------------------------------- parser ---

namespace {

struct Cursor {
    const uint8_t *p;
    const uint8_t *e;
    bool empty() const { return p == e; }
    size_t size() const { return (size_t)(e - p); }
};

// gfaline.rs:153-158 (atoi FromRadix10: leading digits, wrapping arithmetic)
bool parse_num(Cursor *c, uint64_t *out) {
    const uint8_t *s = c->p;
    uint64_t v = 0;
    while (s < c->e && *s >= '0' && *s <= '9') v = v * 10 + (uint64_t)(*s++ - '0');
    if (s == c->p) return false;
    c->p = s;
    *out = v;
    return true;
}

bool parse_byte(Cursor *c, uint8_t b) {
    if (c->empty() || *c->p != b) return false;
    c->p++;
    return true;
}

// gfaline.rs:128-142
Cursor parse_field(Cursor *c) {
    const uint8_t *t = (const uint8_t *)memchr(c->p, '\t', c->size());
    Cursor f{c->p, t ? t : c->e};
    c->p = t ? t + 1 : c->e;
    return f;
}

bool parse_orient(Cursor *c, uint32_t *orient) {
    if (c->empty()) return false;
    if (*c->p == '+') *orient = 0;
    else if (*c->p == '-') *orient = 1;
    else return false;
    c->p++;
    return true;
}

// gfaline.rs:174-198; ops are appended to `out`.
bool parse_align(Cursor *c, std::vector<uint32_t> *out, const char **why) {  // (a scratch list: the ops reach the store once their line has parsed)
    while (!c->empty() && *c->p >= '0' && *c->p <= '9') {
        uint64_t len;
        parse_num(c, &len);
        len &= 0xFFFFFFFFull;  // parse_num::<u32>
        if (c->empty()) { *why = "alignment: missing opcode"; return false; }
        uint32_t op;
        switch (*c->p) {
            case 'M': op = kMatch; break;
            case 'N': op = kGap; break;
            case 'D': op = kDeletion; break;
            case 'I': op = kInsertion; break;
            default: *why = "expected align op"; return false;
        }
        if (len & ~0xFFull) { *why = "length too large"; return false; }  // flatgfa.rs:228
        out->push_back(((uint32_t)len << 8) | op);
        c->p++;
    }
    return true;
}

bool make_handle(uint32_t seg, bool forward, Handle *h, const char **why) {
    if (seg & 0x80000000u) { *why = "index too large"; return false; }  // flatgfa.rs:194
    h->bits = (seg << 1) | (forward ? 0u : 1u);
    return true;
}

// Where each path's steps go, when parse_steps_parallel has put them there.
struct StepsPre {
    bool ok = false;
    std::vector<uint64_t> begin;  // [paths + 1]
};

// One chunk of one path's step field: whole items, `want` of them, to be written from `out` on.
struct StepChunk {
    const uint8_t *p, *e;
    uint64_t want, out;
};

// Strict form of StepsParser (gfaline.rs:200-263) over whole items: digits, a sign, then a comma
// or the end of the chunk.  False for anything else; the caller then falls back to the sequential
// parser, which decides whether it is an error and which.
bool parse_step_chunk(const StepChunk &c, const NameMap &names, Handle *steps) {
    const uint8_t *s = c.p;
    uint64_t n = 0;
    while (s < c.e) {
        uint64_t seg = 0;
        const uint8_t *d = s;
        while (s < c.e && *s >= '0' && *s <= '9') seg = seg * 10 + (uint64_t)(*s++ - '0');
        if (s == d || s == c.e || (*s != '+' && *s != '-')) return false;
        uint32_t id;
        if (!names.get(seg, &id) || (id & 0x80000000u) || n == c.want) return false;
        steps[c.out + n++].bits = (id << 1) | (*s == '+' ? 0u : 1u);
        ++s;
        if (s < c.e && *s++ != ',') return false;
    }
    return n == c.want;
}

template <class S>
void parse_steps_parallel(const std::vector<Cursor> &deferred, const NameMap &names, S *st, StepsPre *pre) {
    size_t kChunk = 1 << 20, min_bytes = 4u << 20;
    if (const char *f = test_hook("FLATGFA_PARSE_CHUNK")) kChunk = std::max<size_t>(1, strtoull(f, nullptr, 10));      // tests
    if (const char *f = test_hook("FLATGFA_PARSE_MIN_BYTES")) min_bytes = strtoull(f, nullptr, 10);                    // tests
    // the step field of every path line, cut after commas into chunks of about a megabyte
    std::vector<StepChunk> chunks;
    std::vector<size_t> first_chunk;  // per path
    size_t bytes = 0;
    for (const Cursor &line : deferred) {
        if (*line.p != 'P') continue;
        if (line.size() < 2 || line.p[1] != '\t') return;
        Cursor rest{line.p + 2, line.e};
        parse_field(&rest);
        const Cursor steps = parse_field(&rest);
        first_chunk.push_back(chunks.size());
        if (!steps.empty() && steps.e[-1] == ',') return;  // (accepted by the reference: not the plain case)
        const uint8_t *a = steps.p;
        while (a < steps.e) {
            const uint8_t *b = steps.e;
            if ((size_t)(steps.e - a) > kChunk + kChunk / 2) {
                const uint8_t *c = (const uint8_t *)memchr(a + kChunk, ',', (size_t)(steps.e - a) - kChunk);
                if (c) b = c + 1;
            }
            chunks.push_back(StepChunk{a, b, b == steps.e ? 1u : 0u, 0});
            a = b;
        }
        bytes += steps.size();
    }
    first_chunk.push_back(chunks.size());
    if (bytes < min_bytes || chunks.empty()) return;  // small inputs: the sequential loop is as fast
    unsigned nt = std::max(1u, std::min({std::thread::hardware_concurrency(), 32u, (unsigned)chunks.size()}));
    if (const char *f = getenv("FLATGFA_PARSE_THREADS")) {  // 0 = the sequential parser only
        if (strtol(f, nullptr, 10) <= 0) return;
        nt = (unsigned)std::min(64l, strtol(f, nullptr, 10));
    }
    std::atomic<size_t> next{0};
    const auto run = [&](const std::function<void(size_t)> &fn) {
        next = 0;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&] {
                for (size_t i = next++; i < chunks.size(); i = next++) fn(i);
            });
        for (auto &x : th) x.join();
    };
    // items per chunk = its commas (every one ends an item; the field's last item has none)
    run([&](size_t i) {
        uint64_t n = 0;
        for (const uint8_t *q = chunks[i].p; (q = (const uint8_t *)memchr(q, ',', (size_t)(chunks[i].e - q))) != nullptr; ++q) ++n;
        chunks[i].want += n;
    });
    uint64_t total = st->steps.size();
    pre->begin.assign(first_chunk.size(), 0);
    for (size_t pth = 0; pth + 1 < first_chunk.size(); ++pth) {
        pre->begin[pth] = total;
        for (size_t i = first_chunk[pth]; i < first_chunk[pth + 1]; ++i) {
            chunks[i].out = total;
            total += chunks[i].want;
        }
    }
    pre->begin.back() = total;
    if (total > 0xFFFFFFFFull) return;  // (spans are 32-bit; the sequential loop reports what the reference would)
    const size_t before = st->steps.size();
    st->steps.resize(total);
    std::atomic<bool> plain{true};
    run([&](size_t i) {
        if (plain.load(std::memory_order_relaxed) && !parse_step_chunk(chunks[i], names, st->steps.data())) plain = false;
    });
    if (!plain) {
        st->steps.resize(before);
        return;
    }
    pre->ok = true;
}

// Parser<P: StoreFamily> (parse.rs:12-22): the same parser for the heap store and for the store inside a file.
template <class S>
bool parse_gfa_into(const uint8_t *buf, size_t n, S *st, std::string *err, bool stream_mode) {
    const auto t_begin = std::chrono::steady_clock::now();
    NameMap names;
    std::vector<Cursor> deferred;
    auto fail = [&](const char *why, size_t lineno) {
        *err = std::string("parse error (line ") + std::to_string(lineno) + "): " + why;
        return false;
    };

    // MemchrSplit (memfile.rs:51-63): a last line with no '\n' is never yielded.
    size_t pos = 0, lineno = 0;
    while (pos < n) {
        const uint8_t *nl = (const uint8_t *)memchr(buf + pos, '\n', n - pos);
        if (!nl) {
            if (!stream_mode) break;
            nl = buf + n;  // BufRead::split yields the unterminated tail (parse.rs:32)
        }
        Cursor line{buf + pos, nl};
        pos = (size_t)(nl - buf) + 1;
        ++lineno;
        if (line.empty()) return fail("empty line", lineno);  // parse.rs:83 indexes line[0]
        uint8_t kind = *line.p;
        if (kind == 'P' || kind == 'L') {
            st->line_order.push_back(kind == 'P' ? kPath : kLink);
            deferred.push_back(line);
            continue;
        }
        if (line.size() < 2 || line.p[1] != '\t') return fail("expected marker and tab", lineno);
        Cursor rest{line.p + 2, line.e};
        if (kind == 'H') {
            st->line_order.push_back(kHeader);
            if (!st->header.empty()) return fail("more than one header", lineno);  // flatgfa.rs:444
            st->header.assign(rest.p, rest.e);
        } else if (kind == 'S') {
            uint64_t name;
            if (!parse_num(&rest, &name)) return fail("expected number", lineno);
            if (!parse_byte(&rest, '\t')) return fail("expected byte", lineno);
            Cursor seq = parse_field(&rest);
            st->line_order.push_back(kSegment);
            Segment s;
            s.name = name;
            s.seq.start = (uint32_t)st->seq_data.size();
            st->seq_data.insert(st->seq_data.end(), seq.p, seq.e);
            s.seq.end = (uint32_t)st->seq_data.size();
            s.optional.start = (uint32_t)st->optional_data.size();
            st->optional_data.insert(st->optional_data.end(), rest.p, rest.e);
            s.optional.end = (uint32_t)st->optional_data.size();
            uint32_t id = (uint32_t)st->segs.size();
            st->segs.push_back(s);
            names.insert(name, id);
        } else {
            return fail("unhandled line kind", lineno);
        }
    }

    // "Unwind" the deferred links and paths in file order (parse.rs:108-123).
    size_t dn = 0;
    if (stream_mode)  // parse_stream unwinds links first, then paths (parse.rs:63-72)
        std::stable_partition(deferred.begin(), deferred.end(), [](const Cursor &c) { return *c.p == 'L'; });
    // The step lists are nearly all of a pangenome's text.  When every one of them is plain
    // -- `name(+|-)` items with known names, separated by single commas -- they are parsed by
    // several threads straight into place; anything else (an error the reference would report, an
    // oddity it accepts) leaves `pre.ok` false and the loop below does it all, in order.
    StepsPre pre;
    const auto t_lines = std::chrono::steady_clock::now();
    parse_steps_parallel(deferred, names, st, &pre);
    const auto t_steps = std::chrono::steady_clock::now();
    size_t n_paths_seen = 0;
    for (Cursor line : deferred) {
        ++dn;
        if (line.size() < 2 || line.p[1] != '\t') return fail("expected marker and tab (deferred)", dn);
        Cursor rest{line.p + 2, line.e};
        const char *why = "";
        if (*line.p == 'L') {
            uint64_t from_name, to_name;
            uint32_t fo, to;
            if (!parse_num(&rest, &from_name) || !parse_byte(&rest, '\t') || !parse_orient(&rest, &fo) ||
                !parse_byte(&rest, '\t') || !parse_num(&rest, &to_name) || !parse_byte(&rest, '\t') ||
                !parse_orient(&rest, &to) || !parse_byte(&rest, '\t'))
                return fail("malformed link", dn);
            size_t a0 = st->alignment.size();
            std::vector<uint32_t> ops;
            if (!parse_align(&rest, &ops, &why)) return fail(why, dn);
            if (!rest.empty()) return fail("expected end of line", dn);
            uint32_t fid, tid;
            Link l;
            Handle fh, th;
            if (!names.get(from_name, &fid) || !make_handle(fid, fo == 0, &fh, &why)) return fail("link: unknown segment", dn);
            if (!names.get(to_name, &tid) || !make_handle(tid, to == 0, &th, &why)) return fail("link: unknown segment", dn);
            st->alignment.insert(st->alignment.end(), ops.begin(), ops.end());
            l.from = fh.bits;
            l.to = th.bits;
            l.overlap.start = (uint32_t)a0;
            l.overlap.end = (uint32_t)st->alignment.size();
            st->links.push_back(l);
        } else {
            Cursor name = parse_field(&rest);
            Cursor steps = parse_field(&rest);
            // parse_maybe_overlap_list, gfaline.rs:102-125
            std::vector<std::vector<uint32_t>> ovs;
            if (!(rest.size() == 1 && *rest.p == '*')) {
                while (!rest.empty()) {
                    ovs.emplace_back();
                    if (!parse_align(&rest, &ovs.back(), &why)) return fail(why, dn);
                    if (!rest.empty() && !parse_byte(&rest, ',')) return fail("expected byte", dn);
                }
            }
            // StepsParser, gfaline.rs:200-263.  The byte that stops the scan is
            // consumed before `rest()` is examined (parse.rs:155).
            Path p;
            if (pre.ok) {  // the steps are already in place (parse_steps_parallel)
                p.steps.start = (uint32_t)pre.begin[n_paths_seen];
                p.steps.end = (uint32_t)pre.begin[n_paths_seen + 1];
                ++n_paths_seen;
            } else {
                p.steps.start = (uint32_t)st->steps.size();
                const uint8_t *s = steps.p;
                uint64_t seg = 0;
                bool want_seg = true;
                while (s < steps.e) {
                    uint8_t b = *s++;
                    if (want_seg) {
                        if (b == '+' || b == '-') {
                            want_seg = false;
                            uint32_t id;
                            Handle h;
                            if (!names.get(seg, &id)) return fail("path: unknown segment", dn);
                            if (!make_handle(id, b == '+', &h, &why)) return fail(why, dn);
                            st->steps.push_back(h);
                        } else if (b >= '0' && b <= '9') {
                            seg = seg * 10 + (uint64_t)(b - '0');
                        } else {
                            break;
                        }
                    } else {
                        if (b == ',') {
                            want_seg = true;
                            seg = 0;
                        } else {
                            break;
                        }
                    }
                }
                if (s != steps.e) return fail("path steps: trailing bytes", dn);
                p.steps.end = (uint32_t)st->steps.size();
            }
            p.overlaps.start = (uint32_t)st->overlaps.size();
            for (auto &ops : ovs) {
                Span a;
                a.start = (uint32_t)st->alignment.size();
                st->alignment.insert(st->alignment.end(), ops.begin(), ops.end());
                a.end = (uint32_t)st->alignment.size();
                st->overlaps.push_back(a);
            }
            p.overlaps.end = (uint32_t)st->overlaps.size();
            p.name.start = (uint32_t)st->name_data.size();
            st->name_data.insert(st->name_data.end(), name.p, name.e);
            p.name.end = (uint32_t)st->name_data.size();
            st->paths.push_back(p);
        }
    }
    if (getenv("FLATGFA_TIMING")) {
        const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "parse_gfa: lines %.1f ms, steps %.1f ms (%s), links/paths %.1f ms\n", ms(t_begin, t_lines), ms(t_lines, t_steps),
                pre.ok ? "threads" : "in order", ms(t_steps, std::chrono::steady_clock::now()));
    }
    return true;
}

}  // namespace

bool parse_gfa(const uint8_t *buf, size_t n, Store *st, std::string *err, bool stream_mode) {
    *st = Store();
    return parse_gfa_into(buf, n, st, err, stream_mode);
}

uint64_t FixedStore::pool_len(int ix) const {
    switch (ix) {
        case pHeader: return header.size();
        case pSegs: return segs.size();
        case pPaths: return paths.size();
        case pLinks: return links.size();
        case pSteps: return steps.size();
        case pSeqData: return seq_data.size();
        case pOverlaps: return overlaps.size();
        case pAlignment: return alignment.size();
        case pNameData: return name_data.size();
        case pOptionalData: return optional_data.size();
        case pLineOrder: return line_order.size();
    }
    return 0;
}

bool toc_file_size(const uint64_t cap[11], size_t *total, std::string *err) {
    unsigned __int128 t = sizeof(Toc);
    for (int i = 0; i < 11; ++i) t += (unsigned __int128)cap[i] * kPoolElemSize[i];
    if (t > (unsigned __int128)1 << 46) { *err = "preallocated flatgfa: file too large"; return false; }
    *total = (size_t)t;
    return true;
}

bool parse_gfa_prealloc(const uint8_t *buf, size_t n, bool stream_mode, const uint64_t cap[11], uint8_t *file, std::string *err) {
    // file::init (file.rs:255-272): the table of contents with every length 0, then the store over what follows it
    Toc toc;
    toc.magic = kMagic;
    for (int i = 0; i < 11; ++i) toc.pool[i] = TocSize{0, cap[i]};
    memcpy(file, &toc, sizeof toc);
    uint8_t *at[11];
    size_t off = sizeof toc;
    for (int i = 0; i < 11; ++i) {
        at[i] = file + off;
        off += cap[i] * kPoolElemSize[i];
    }
    FixedStore st;
    st.header = FixedVec<uint8_t>(at[pHeader], cap[pHeader], pHeader);
    st.segs = FixedVec<Segment>(reinterpret_cast<Segment *>(at[pSegs]), cap[pSegs], pSegs);
    st.paths = FixedVec<Path>(reinterpret_cast<Path *>(at[pPaths]), cap[pPaths], pPaths);
    st.links = FixedVec<Link>(reinterpret_cast<Link *>(at[pLinks]), cap[pLinks], pLinks);
    st.steps = FixedVec<Handle>(reinterpret_cast<Handle *>(at[pSteps]), cap[pSteps], pSteps);
    st.seq_data = FixedVec<uint8_t>(at[pSeqData], cap[pSeqData], pSeqData);
    st.overlaps = FixedVec<Span>(reinterpret_cast<Span *>(at[pOverlaps]), cap[pOverlaps], pOverlaps);
    st.alignment = FixedVec<AlignOp>(reinterpret_cast<AlignOp *>(at[pAlignment]), cap[pAlignment], pAlignment);
    st.name_data = FixedVec<uint8_t>(at[pNameData], cap[pNameData], pNameData);
    st.optional_data = FixedVec<uint8_t>(at[pOptionalData], cap[pOptionalData], pOptionalData);
    st.line_order = FixedVec<uint8_t>(at[pLineOrder], cap[pLineOrder], pLineOrder);
    try {
        if (!parse_gfa_into(buf, n, &st, err, stream_mode)) return false;
    } catch (const CapacityError &e) {  // (the push that does not fit: the reference's fixed-capacity store panics)
        *err = std::string("preallocated flatgfa: the ") + kPoolName[e.pool] + " pool needs more than the " + std::to_string(e.capacity) + " entries its capacity allows";
        return false;
    }
    for (int i = 0; i < 11; ++i) toc.pool[i].len = st.pool_len(i);  // Toc::for_fixed_store
    memcpy(file, &toc, sizeof toc);
    return true;
}

// ------------------------------------------------------ .flatgfa container ---

bool view_flatgfa(const uint8_t *data, size_t n, View *out, std::string *err) {
    if (n < sizeof(Toc)) { *err = "flatgfa file: shorter than its table of contents"; return false; }
    Toc toc;
    memcpy(&toc, data, sizeof toc);
    if (toc.magic != kMagic) { *err = "flatgfa file: bad magic number"; return false; }
    size_t off = sizeof(Toc);
    const void *ptr[11];
    for (int i = 0; i < 11; ++i) {
        const TocSize &sz = toc.pool[i];
        if (sz.len > sz.capacity) { *err = "flatgfa file: len > capacity"; return false; }
        unsigned __int128 bytes = (unsigned __int128)sz.capacity * kPoolElemSize[i];
        if (bytes > (unsigned __int128)(n - off)) {
            *err = std::string("flatgfa file: region out of bounds: ") + kPoolName[i];
            return false;
        }
        ptr[i] = data + off;
        off += (size_t)bytes;
    }
    *out = View();
    out->header = {(const uint8_t *)ptr[pHeader], (size_t)toc.pool[pHeader].len};
    out->segs = {(const Segment *)ptr[pSegs], (size_t)toc.pool[pSegs].len};
    out->paths = {(const Path *)ptr[pPaths], (size_t)toc.pool[pPaths].len};
    out->links = {(const Link *)ptr[pLinks], (size_t)toc.pool[pLinks].len};
    out->steps = {(const Handle *)ptr[pSteps], (size_t)toc.pool[pSteps].len};
    out->seq_data = {(const uint8_t *)ptr[pSeqData], (size_t)toc.pool[pSeqData].len};
    out->overlaps = {(const Span *)ptr[pOverlaps], (size_t)toc.pool[pOverlaps].len};
    out->alignment = {(const AlignOp *)ptr[pAlignment], (size_t)toc.pool[pAlignment].len};
    out->name_data = {(const uint8_t *)ptr[pNameData], (size_t)toc.pool[pNameData].len};
    out->optional_data = {(const uint8_t *)ptr[pOptionalData], (size_t)toc.pool[pOptionalData].len};
    out->line_order = {(const uint8_t *)ptr[pLineOrder], (size_t)toc.pool[pLineOrder].len};
    return validate_spans(*out, err);
}

// Every span stored inside a pool must lie inside the pool it points into, and every link handle
// must name a segment: the reference finds out when it indexes (pool.rs:341-347 panics); here a
// file that fails is rejected when it is opened, so that the accessors can index unchecked.
// O(segments + paths + links); the step ids are checked by the kernels (and by
// validate_step_ids before a host-side walk).
bool validate_spans(const View &v, std::string *err) {
    const auto inside = [](const Span &sp, size_t n) { return sp.start <= sp.end && (size_t)sp.end <= n; };
    for (size_t i = 0; i < v.segs.len; ++i) {
        const Segment sg = v.segs[i];
        if (!inside(sg.seq, v.seq_data.len)) { *err = "flatgfa file: segment " + std::to_string(i) + ": sequence span outside seq_data"; return false; }
        if (!inside(sg.optional, v.optional_data.len)) { *err = "flatgfa file: segment " + std::to_string(i) + ": optional span outside optional_data"; return false; }
    }
    for (size_t i = 0; i < v.paths.len; ++i) {
        const Path p = v.paths[i];
        if (!inside(p.name, v.name_data.len)) { *err = "flatgfa file: path " + std::to_string(i) + ": name span outside name_data"; return false; }
        if (!inside(p.steps, v.steps.len)) { *err = "flatgfa file: path " + std::to_string(i) + ": step span outside the steps pool"; return false; }
        if (!inside(p.overlaps, v.overlaps.len)) { *err = "flatgfa file: path " + std::to_string(i) + ": overlap span outside the overlaps pool"; return false; }
    }
    for (size_t i = 0; i < v.overlaps.len; ++i)
        if (!inside(v.overlaps[i], v.alignment.len)) { *err = "flatgfa file: overlap " + std::to_string(i) + ": span outside the alignment pool"; return false; }
    for (size_t i = 0; i < v.links.len; ++i) {
        const Link l = v.links[i];
        if (!inside(l.overlap, v.alignment.len)) { *err = "flatgfa file: link " + std::to_string(i) + ": overlap span outside the alignment pool"; return false; }
        if ((size_t)(l.from >> 1) >= v.segs.len || (size_t)(l.to >> 1) >= v.segs.len) { *err = "flatgfa file: link " + std::to_string(i) + ": segment id out of range"; return false; }
    }
    return true;
}

bool validate_step_ids(const View &v) {
    uint32_t mx = 0;
    for (size_t i = 0; i < v.steps.len; ++i) {
        const uint32_t b = v.steps[i].bits;  // (a copy: std::max takes references, and the pool is align-1)
        mx = std::max(mx, b);
    }
    return v.steps.len == 0 || (size_t)(mx >> 1) < v.segs.len;
}

size_t flatgfa_file_size(const View &v) {
    size_t total = sizeof(Toc);
    for (int i = 0; i < 11; ++i) total += v.pool_len(i) * kPoolElemSize[i];
    return total;
}

void dump_flatgfa(const View &v, uint8_t *buf) {
    Toc toc;
    toc.magic = kMagic;
    for (int i = 0; i < 11; ++i) toc.pool[i] = TocSize{v.pool_len(i), v.pool_len(i)};
    memcpy(buf, &toc, sizeof toc);
    size_t off = sizeof toc;
    for (int i = 0; i < 11; ++i) {
        size_t bytes = v.pool_len(i) * kPoolElemSize[i];
        if (bytes) memcpy(buf + off, v.pool_data(i), bytes);
        off += bytes;
    }
}

// The preallocated ("in-place") container of `fgfa -m -p N -o OUT [-I GFA]` (cli/main.rs:216-248):
// every pool's region is `capacity` items long, of which `len` are in use; capacities come from
// measurements of the GFA text (parse.rs:176-216 -> Toc::estimate, file.rs:136-158) or, with no
// text to measure (stdin), from Toc::guess(factor) (file.rs:117-132).
bool estimate_toc(const uint8_t *buf, size_t n, uint64_t cap[11], std::string *err) {
    uint64_t segs = 0, links = 0, paths = 0, header_bytes = 0, seg_bytes = 0, path_bytes = 0;
    size_t at = 0;
    while (at < n) {
        const uint8_t marker = buf[at];
        const void *nl = memchr(buf + at, '\n', n - at);
        const size_t rest = n - at;
        const size_t next = nl ? (size_t)((const uint8_t *)nl - (buf + at)) : rest + 1;  // (memchr(..).unwrap_or(rest.len() + 1))
        switch (marker) {
            case 'H': header_bytes += next; break;
            case 'S': segs += 1; seg_bytes += next; break;
            case 'L': links += 1; break;
            case 'P': paths += 1; path_bytes += next; break;
            default: *err = "unknown line type"; return false;  // (the reference panics, parse.rs:205)
        }
        if (next >= rest) break;
        at += next + 1;
    }
    const uint64_t c[11] = {header_bytes, segs, paths, links, path_bytes / 3, seg_bytes, (links + paths) * 2, links * 2 + paths * 4,
                            paths * 512, links * 16, segs + links + paths + 8};
    memcpy(cap, c, sizeof c);
    return true;
}

void guess_toc(uint64_t f, uint64_t cap[11]) {
    const uint64_t c[11] = {128, 32 * f * f, f, 32 * f * f, 1024 * f * f, 512 * f * f, 256 * f, 64 * f * f, 64 * f, 512 * f * f, 64 * f * f};
    memcpy(cap, c, sizeof c);
}

bool prealloc_file_size(const View &v, const uint64_t cap[11], size_t *total, std::string *err) {
    static const char *names[11] = {"header", "segs", "paths", "links", "steps", "seq_data", "overlaps", "alignment", "name_data", "optional_data", "line_order"};
    unsigned __int128 t = sizeof(Toc);
    for (int i = 0; i < 11; ++i) {
        if (v.pool_len(i) > cap[i]) {  // (where the reference's fixed-capacity store panics on the push that does not fit)
            *err = std::string("preallocated flatgfa: the ") + names[i] + " pool needs " + std::to_string(v.pool_len(i)) + " entries, the estimate allows " + std::to_string(cap[i]);
            return false;
        }
        t += (unsigned __int128)cap[i] * kPoolElemSize[i];
    }
    if (t > (unsigned __int128)1 << 46) { *err = "preallocated flatgfa: file too large"; return false; }
    *total = (size_t)t;
    return true;
}

void dump_flatgfa_prealloc(const View &v, const uint64_t cap[11], uint8_t *buf) {  // buf: prealloc_file_size bytes, zeroed
    Toc toc;
    toc.magic = kMagic;
    for (int i = 0; i < 11; ++i) toc.pool[i] = TocSize{v.pool_len(i), cap[i]};
    memcpy(buf, &toc, sizeof toc);
    size_t off = sizeof toc;
    for (int i = 0; i < 11; ++i) {
        const size_t bytes = v.pool_len(i) * kPoolElemSize[i];
        if (bytes) memcpy(buf + off, v.pool_data(i), bytes);
        off += cap[i] * kPoolElemSize[i];
    }
}

// ------------------------------------------------------------ GFA printer ---

static void put_u64(std::string *o, uint64_t v) {
    char b[24];
    int n = snprintf(b, sizeof b, "%llu", (unsigned long long)v);
    o->append(b, (size_t)n);
}

namespace {

void put_alignment(const View &v, Span a, std::string *o) {
    static const char letters[4] = {'M', 'N', 'D', 'I'};  // print.rs:14-23
    if (a.start == a.end) o->append("0M");
    for (uint32_t i = a.start; i < a.end; ++i) {
        uint32_t op = v.alignment[i].bits;
        put_u64(o, op >> 8);
        o->push_back(letters[op & 3]);
    }
}

void put_handle(const View &v, uint32_t bits, std::string *o) {
    put_u64(o, v.segs[bits >> 1].name);
    o->push_back((bits & 1) ? '-' : '+');
}

bool put_seg(const View &v, size_t i, std::string *o) {
    const Segment &s = v.segs[i];
    o->append("S\t");
    put_u64(o, s.name);
    o->push_back('\t');
    o->append((const char *)v.seq_data.data + s.seq.start, s.seq.len());
    if (s.optional.start != s.optional.end) {
        o->push_back('\t');
        o->append((const char *)v.optional_data.data + s.optional.start, s.optional.len());
    }
    o->push_back('\n');
    return true;
}

bool put_path(const View &v, size_t i, std::string *o, std::string *err) {
    const Path &p = v.paths[i];
    o->append("P\t");
    o->append((const char *)v.name_data.data + p.name.start, p.name.len());
    o->push_back('\t');
    if (p.steps.start == p.steps.end) { *err = "print: path with no steps"; return false; }  // print.rs:48 steps[0]
    for (uint32_t k = p.steps.start; k < p.steps.end; ++k) {
        if (k != p.steps.start) o->push_back(',');
        put_handle(v, v.steps[k].bits, o);
    }
    o->push_back('\t');
    if (p.overlaps.start == p.overlaps.end) {
        o->push_back('*');
    } else {
        for (uint32_t k = p.overlaps.start; k < p.overlaps.end; ++k) {
            if (k != p.overlaps.start) o->push_back(',');
            put_alignment(v, v.overlaps[k], o);
        }
    }
    o->push_back('\n');
    return true;
}

void put_link(const View &v, size_t i, std::string *o) {
    const Link &l = v.links[i];
    o->append("L\t");
    put_u64(o, v.segs[l.from >> 1].name);
    o->append((l.from & 1) ? "\t-\t" : "\t+\t");
    put_u64(o, v.segs[l.to >> 1].name);
    o->append((l.to & 1) ? "\t-\t" : "\t+\t");
    put_alignment(v, l.overlap, o);
    o->push_back('\n');
}

void put_header(const View &v, std::string *o) {
    o->append("H\t");
    o->append((const char *)v.header.data, v.header.len);
    o->push_back('\n');
}

}  // namespace

bool print_gfa(const View &v, std::string *o, std::string *err) {
    if (v.line_order.len == 0) {  // write_normalized, print.rs:134-150
        if (v.header.len) put_header(v, o);
        for (size_t i = 0; i < v.segs.len; ++i) put_seg(v, i, o);
        for (size_t i = 0; i < v.paths.len; ++i)
            if (!put_path(v, i, o, err)) return false;
        for (size_t i = 0; i < v.links.len; ++i) put_link(v, i, o);
        return true;
    }
    size_t si = 0, pi = 0, li = 0;  // write_preserved, print.rs:99-131
    for (size_t k = 0; k < v.line_order.len; ++k) {
        switch (v.line_order[k]) {
            case kHeader:
                if (!v.header.len) { *err = "print: empty header"; return false; }
                put_header(v, o);
                break;
            case kSegment:
                if (si >= v.segs.len) { *err = "print: too few segments"; return false; }
                put_seg(v, si++, o);
                break;
            case kPath:
                if (pi >= v.paths.len) { *err = "print: too few paths"; return false; }
                if (!put_path(v, pi++, o, err)) return false;
                break;
            case kLink:
                if (li >= v.links.len) { *err = "print: too few links"; return false; }
                put_link(v, li++, o);
                break;
            default:
                *err = "print: bad line kind";
                return false;
        }
    }
    return true;
}

// ---------------------------------------------------------------- emitters ---

std::string format_float(double x, int digits) {
    char buf[512];
    int n;
    // Rust's Display spells these "NaN" / "inf" / "-inf".
    if (std::isnan(x)) n = snprintf(buf, sizeof buf, "NaN");
    else if (std::isinf(x)) n = snprintf(buf, sizeof buf, x > 0 ? "inf" : "-inf");
    else n = snprintf(buf, sizeof buf, "%.*f", digits, x);  // glibc: correctly rounded, like Rust
    while (n > 0 && buf[n - 1] == '0') --n;
    while (n > 0 && buf[n - 1] == '.') --n;
    return std::string(buf, (size_t)n);
}

// decimal digits of x at p, returns the end.  Two digits at a time from a table, written backwards from where the number
// ends (a million-line table is three million numbers: a division per digit and a reversal were two thirds of its 2 ms).
static const char kDigitPairs[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
static inline unsigned digits_u64(uint64_t x) {
    unsigned n = 1;
    while (x >= 10000u) {
        x /= 10000u;
        n += 4;
    }
    return n + (x >= 10u) + (x >= 100u) + (x >= 1000u);
}
static inline char *write_u64(char *p, uint64_t x) {
    char *const end = p + digits_u64(x);
    char *q = end;
    while (x >= 100u) {
        const unsigned r = (unsigned)(x % 100u);
        x /= 100u;
        q -= 2;
        memcpy(q, kDigitPairs + 2 * r, 2);
    }
    if (x >= 10u) memcpy(q - 2, kDigitPairs + 2 * x, 2);
    else q[-1] = (char)('0' + x);
    return end;
}
// (32-bit numbers: the device's counts and `seg.name as u32` -- 32-bit divisions)
static inline char *write_u64(char *p, uint32_t x32) {
    uint32_t x = x32;
    const unsigned n = 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) +
                       (x >= 100000000u) + (x >= 1000000000u);
    char *const end = p + n;
    char *q = end;
    while (x >= 100u) {
        const uint32_t r = x % 100u;
        x /= 100u;
        q -= 2;
        memcpy(q, kDigitPairs + 2 * r, 2);
    }
    if (x >= 10u) memcpy(q - 2, kDigitPairs + 2 * x, 2);
    else q[-1] = (char)('0' + x);
    return end;
}

// One line per segment, in pool order: `{seg.name as u32}\t{depth}\t{uniq}\n` (depth.rs:70-79).
// A million lines are 13 MB of decimal digits: big tables are formatted by several threads, each
// its own stretch of segments into its own buffer, and stitched together in order.
template <typename T>
static char *emit_seg_lines(const View &v, const T *depth, const T *uniq, size_t lo, size_t hi, char *p) {
    for (size_t i = lo; i < hi; ++i) {
        p = write_u64(p, (uint32_t)v.segs[i].name);  // `seg.name as u32`, depth.rs:71
        *p++ = '\t';
        p = write_u64(p, depth[i]);
        *p++ = '\t';
        p = write_u64(p, uniq[i]);
        *p++ = '\n';
    }
    return p;
}

void emit_seg_depth(const View &v, const uint64_t *depth, const uint64_t *uniq, std::string *out) {
    out->append("#node.id\tdepth\tdepth.uniq\n");
    constexpr size_t kLineMax = 53;  // at most 10 + 20 + 20 digits, two tabs and a newline
    const size_t S = v.segs.len;
    unsigned nthr = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (S < (1u << 16)) nthr = 1;
    if (nthr == 1) {
        const size_t head = out->size();
        out->resize(head + S * kLineMax);
        char *p = emit_seg_lines(v, depth, uniq, 0, S, &(*out)[head]);
        out->resize((size_t)(p - out->data()));
        return;
    }
    struct Part {
        std::unique_ptr<char[]> buf;
        size_t len = 0;
    };
    std::vector<Part> parts(nthr);
    std::vector<std::thread> workers;
    for (unsigned t = 0; t < nthr; ++t)
        workers.emplace_back([&, t]() {
            const size_t lo = S * t / nthr, hi = S * (t + 1) / nthr;
            parts[t].buf.reset(new char[(hi - lo) * kLineMax + 1]);
            parts[t].len = (size_t)(emit_seg_lines(v, depth, uniq, lo, hi, parts[t].buf.get()) - parts[t].buf.get());
        });
    for (auto &w : workers) w.join();
    size_t total = out->size();
    std::vector<size_t> at(nthr);
    for (unsigned t = 0; t < nthr; ++t) {
        at[t] = total;
        total += parts[t].len;
    }
    out->resize(total);
    workers.clear();
    for (unsigned t = 0; t < nthr; ++t)
        workers.emplace_back([&, t]() { memcpy(&(*out)[at[t]], parts[t].buf.get(), parts[t].len); });
    for (auto &w : workers) w.join();
}

// (decimal digits of a 32-bit number)
static inline unsigned digits_u32(uint32_t x) {
    return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) + (x >= 100000000u) +
           (x >= 1000000000u);
}

// The same table from the device's own 32-bit counts, into ONE malloc'd buffer the caller owns (flatgfa_depth_table: no
// widening pass, no intermediate string).  Every thread first counts the bytes of its stretch of lines (digits only: a third of
// a nanosecond a number), the stretches' offsets follow, and every thread then writes its lines where they belong: the text is
// written once, into pages touched once (per-thread buffers stitched together afterwards were 13 MB more of first touches and
// a second pass over the text: 2.8 -> 1.9 ms for a million lines).  NUL-terminated; *len excludes the NUL.  nullptr: out of memory.
char *emit_seg_depth_u32_malloc(const View &v, const uint32_t *depth, const uint32_t *uniq, size_t *len) {
    static const char kHead[] = "#node.id\tdepth\tdepth.uniq\n";
    constexpr size_t kHeadLen = sizeof kHead - 1;
    const size_t S = v.segs.len;
    unsigned nthr = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (S < (1u << 16)) nthr = 1;
    std::vector<size_t> bytes(nthr, 0), at(nthr + 1, 0);
    const auto count = [&](unsigned t) {
        const size_t lo = S * t / nthr, hi = S * (t + 1) / nthr;
        size_t n = 3 * (hi - lo);  // two tabs and a newline per line
        for (size_t i = lo; i < hi; ++i) n += digits_u32((uint32_t)v.segs[i].name) + digits_u32(depth[i]) + digits_u32(uniq[i]);
        bytes[t] = n;
    };
    char *out = nullptr;
    bool failed = false;
    if (nthr == 1) {
        count(0);
        at[1] = kHeadLen + bytes[0];
        out = (char *)malloc(at[1] + 1);
        if (!out) return nullptr;
        char *end = emit_seg_lines(v, depth, uniq, 0, S, out + kHeadLen);
        (void)end;
    } else {
        // one round of threads: count, meet, (thread 0 lays the stretches out and allocates), meet, write
        std::mutex mu;
        std::condition_variable cv;
        unsigned arrived = 0, phase = 0;
        const auto meet = [&](bool leader, const std::function<void()> &then) {
            std::unique_lock<std::mutex> lk(mu);
            const unsigned my = phase;
            if (++arrived == nthr) {
                arrived = 0;
                then();
                ++phase;
                cv.notify_all();
            } else {
                cv.wait(lk, [&] { return phase != my; });
            }
            (void)leader;
        };
        const auto work = [&](unsigned t) {
            count(t);
            meet(t == 0, [&] {  // (whoever arrives last does it: everything it reads has been written under the lock's order)
                at[0] = kHeadLen;
                for (unsigned k = 0; k < nthr; ++k) at[k + 1] = at[k] + bytes[k];
                out = (char *)malloc(at[nthr] + 1);
                failed = out == nullptr;
            });
            if (failed) return;
            const size_t lo = S * t / nthr, hi = S * (t + 1) / nthr;
            emit_seg_lines(v, depth, uniq, lo, hi, out + at[t]);
        };
        std::vector<std::thread> workers;
        for (unsigned t = 1; t < nthr; ++t) workers.emplace_back(work, t);
        work(0);
        for (auto &w : workers) w.join();
        if (failed) return nullptr;
    }
    memcpy(out, kHead, kHeadLen);
    out[at[nthr]] = 0;
    *len = at[nthr];
    return out;
}

void emit_path_depth(const View &v, const uint32_t *path_ids, size_t n, const uint64_t *lengths,
                     const double *means, std::string *out) {
    out->append("#path\tstart\tend\tmean.depth\n");
    for (size_t k = 0; k < n; ++k) {
        const Path &p = v.paths[path_ids[k]];
        out->append((const char *)v.name_data.data + p.name.start, p.name.len());
        out->append("\t0\t");
        put_u64(out, lengths[k]);
        out->push_back('\t');
        out->append(format_float(means[k], 2));
        out->push_back('\n');
    }
}

// ------------------------------------------------------------- mapped file ---

bool MappedFile::open(const char *path, std::string *err) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { *err = std::string("cannot open ") + path; return false; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { ::close(fd); *err = std::string("cannot stat ") + path; return false; }
    size = (size_t)sb.st_size;
    if (size == 0) {
        data = (const uint8_t *)"";
        ::close(fd);
        return true;
    }
    void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { size = 0; *err = std::string("cannot mmap ") + path; return false; }
    data = (const uint8_t *)m;
    return true;
}

MappedFile::~MappedFile() {
    if (data && size) munmap((void *)data, size);
}

}  // namespace fgfa

// ------------------------------------------------- window / interval depth ---
// Host side of flatgfa/src/ops/window_depth.rs: node depth comes from the GPU; the f64
// accumulation is order-dependent, so it is done here exactly in the reference's order.

namespace fgfa {

bool parse_bed(const uint8_t *buf, size_t n, Bed *out, std::string *err) {
    *out = Bed();
    size_t pos = 0;
    while (pos < n) {  // MemchrSplit: an unterminated last line is dropped (memfile.rs:51-63)
        const uint8_t *nl = (const uint8_t *)memchr(buf + pos, '\n', n - pos);
        if (!nl) break;
        const uint8_t *p = buf + pos, *e = nl;
        pos = (size_t)(nl - buf) + 1;
        if (p < e && *p == '#') continue;  // flatbed.rs:141-143
        const uint8_t *t = (const uint8_t *)memchr(p, '\t', (size_t)(e - p));
        const uint8_t *name_end = t ? t : e;
        const uint8_t *q = t ? t + 1 : e;
        uint64_t start = 0, end = 0;
        const uint8_t *s = q;
        while (s < e && *s >= '0' && *s <= '9') start = start * 10 + (uint64_t)(*s++ - '0');
        if (s == q) { *err = "BED: expected number"; return false; }
        if (s >= e) { *err = "BED: line ends after the start column"; return false; }  // rest[1..] panics
        const uint8_t *r = s + 1;
        s = r;
        while (s < e && *s >= '0' && *s <= '9') end = end * 10 + (uint64_t)(*s++ - '0');
        if (s == r) { *err = "BED: expected number"; return false; }
        BedEntry be;
        be.name_start = (uint32_t)out->name_data.size();
        out->name_data.insert(out->name_data.end(), p, name_end);
        be.name_end = (uint32_t)out->name_data.size();
        be.start = start;
        be.end = end;
        out->entries.push_back(be);
    }
    return true;
}

void make_windows(const uint8_t *name, size_t name_len, uint64_t start, uint64_t end, uint64_t size, Bed *out) {
    *out = Bed();
    out->name_data.assign(name, name + name_len);
    for (uint64_t pos = start; pos < end;) {  // window_depth.rs:41-51
        const uint64_t e = std::min(pos + size, end);
        out->entries.push_back(BedEntry{0u, (uint32_t)name_len, pos, e});
        pos = e;
    }
}

uint64_t path_length(const View &v, uint32_t path) {  // window_depth.rs:69-77
    uint64_t total = 0;
    const Span sp = v.paths[path].steps;
    for (uint32_t i = sp.start; i < sp.end; ++i) total += v.segs[v.steps[i].segment()].seq.len();
    return total;
}

void interval_depth(const View &v, const uint64_t *seg_depth, uint32_t path, const BedEntry *win, size_t n_win, double *out) {
    for (size_t i = 0; i < n_win; ++i) out[i] = 0.0;
    size_t cur = 0;
    uint64_t pos = 0;
    const Span sp = v.paths[path].steps;
    for (uint32_t i = sp.start; i < sp.end; ++i) {  // weighted_depths, window_depth.rs:84-103
        const uint32_t seg = v.steps[i].segment();
        const uint64_t len = v.segs[seg].seq.len();
        const uint64_t r0 = pos, r1 = pos + len;
        pos = r1;
        const double sdepth = (double)(seg_depth[seg] * len);
        while (cur < n_win) {  // assign_depths, window_depth.rs:116-147
            const uint64_t w0 = win[cur].start, w1 = win[cur].end;
            const uint64_t o0 = std::max(w0, r0), o1 = std::min(w1, r1);
            if (o1 > o0) {
                const double amt = (double)(o1 - o0) / (double)(r1 - r0);
                out[cur] += (sdepth * amt) / (double)(w1 - w0);
            }
            if (w1 > r1) break;
            cur += 1;
        }
    }
}

void emit_interval_depth(const Bed &bed, const double *depths, std::string *out) {  // window_depth.rs:158-170
    for (size_t i = 0; i < bed.entries.size(); ++i) {
        const BedEntry &e = bed.entries[i];
        out->append((const char *)bed.name_data.data() + e.name_start, e.name_end - e.name_start);
        out->push_back('\t');
        put_u64(out, e.start);
        out->push_back('\t');
        put_u64(out, e.end);
        out->push_back('\t');
        out->append(format_float(depths[i], 4));
        out->push_back('\n');
    }
}

void emit_overlap(const View &v, const uint32_t *query_ids, size_t n_q, const uint64_t *path_len, const uint8_t *touch,
                  std::string *out) {  // slow_odgi/overlap.py:17-32
    bool header = false;
    for (size_t k = 0; k < n_q; ++k) {
        const Path &ip = v.paths[query_ids[k]];
        for (size_t j = 0; j < v.paths.len; ++j) {
            if (!touch[k * v.paths.len + j]) continue;
            if (!header) {
                out->append("#path\tstart\tend\tpath.touched\n");
                header = true;
            }
            out->append((const char *)v.name_data.data + ip.name.start, ip.name.len());
            out->append("\t0\t");
            put_u64(out, path_len[k]);
            out->push_back('\t');
            const Path &q = v.paths[j];
            out->append((const char *)v.name_data.data + q.name.start, q.name.len());
            out->push_back('\n');
        }
    }
}

}  // namespace fgfa
