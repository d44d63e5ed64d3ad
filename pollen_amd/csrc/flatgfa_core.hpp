// FlatGFA host-side data model for the MI355X depth engine.
//
// Re-creates, from scratch, the reference's flat-array pangenome graph
// (cucapra/pollen flatgfa/src/flatgfa.rs, pool.rs, file.rs, parse.rs) as far
// as the depth path needs it: byte-compatible PODs, the eleven pools, the GFA
// text parser, the zero-copy `.flatgfa` container, and the odgi-style emitters.
// Device code lives in depth_device.hip; this header is plain C++17.
#pragma once
#include <cstdlib>
// (a test hook, not a user-facing switch: see device_common.hpp)
#ifndef FGFA_TEST_HOOK_DEFINED
#define FGFA_TEST_HOOK_DEFINED
inline const char *test_hook(const char *name) { return std::getenv(name); }
#endif
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace fgfa {

// ---- PODs: byte-compatible with the reference (all repr(packed), align 1) ----
#pragma pack(push, 1)
// pool.rs:80-86
struct Span {
    uint32_t start, end;
    uint32_t len() const { return end - start; }
};
// flatgfa.rs:71-82
struct Segment {
    uint64_t name;
    Span seq;
    Span optional;
};
// flatgfa.rs:99-112
struct Path {
    Span name;
    Span steps;
    Span overlaps;
};
// flatgfa.rs:121-133
struct Link {
    uint32_t from, to;
    Span overlap;
};
// flatgfa.rs:186-209 -- (segment << 1) | orient; Forward = 0, Backward = 1
struct Handle {
    uint32_t bits;
    uint32_t segment() const { return bits >> 1; }
    bool is_forward() const { return (bits & 1u) == 0; }
};
// flatgfa.rs:213-249 -- one op of a CIGAR-like alignment: opcode in the low byte, length above it.  A packed
// wrapper, so that a reference into a file image (whose pools start at any byte) is an align-1 reference.
struct AlignOp {
    uint32_t bits;
};
// file.rs:29-38, 12-27
struct TocSize {
    uint64_t len, capacity;
};
struct Toc {
    uint64_t magic;
    TocSize pool[11];
};
#pragma pack(pop)
static_assert(sizeof(Span) == 8 && sizeof(Segment) == 24 && sizeof(Path) == 24, "layout");
static_assert(sizeof(Link) == 16 && sizeof(Handle) == 4 && sizeof(Toc) == 184 && sizeof(AlignOp) == 4 && alignof(AlignOp) == 1, "layout");

constexpr uint64_t kMagic = 0xB1011054ull;  // file.rs:9
// flatgfa.rs:262-269
enum LineKind : uint8_t { kHeader = 0, kSegment = 1, kPath = 2, kLink = 3 };
// flatgfa.rs:213-221 with the letter mapping of gfaline.rs:178-184 / print.rs:14-23
enum AlignOpcode : uint8_t { kMatch = 0, kGap = 1, kInsertion = 2, kDeletion = 3 };

// The pool order inside a .flatgfa file (file.rs:14-27).
enum PoolIx { pHeader, pSegs, pPaths, pLinks, pSteps, pSeqData, pOverlaps, pAlignment, pNameData, pOptionalData, pLineOrder };
extern const size_t kPoolElemSize[11];
extern const char *const kPoolName[11];

template <class T>
struct Pool {
    const T *data = nullptr;
    size_t len = 0;
    const T &operator[](size_t i) const { return data[i]; }
    const T *begin() const { return data; }
    const T *end() const { return data + len; }
};

// A borrowed view of a whole graph: flatgfa.rs:19-67.
struct View {
    Pool<uint8_t> header;
    Pool<Segment> segs;
    Pool<Path> paths;
    Pool<Link> links;
    Pool<Handle> steps;
    Pool<uint8_t> seq_data;
    Pool<Span> overlaps;
    Pool<AlignOp> alignment;
    Pool<uint8_t> name_data;
    Pool<uint8_t> optional_data;
    Pool<uint8_t> line_order;

    size_t pool_len(int ix) const;
    const void *pool_data(int ix) const;
    // flatgfa.rs:387-394
    int64_t find_path(const uint8_t *name, size_t n) const;
};

// An owning heap store (the reference's HeapGFAStore, flatgfa.rs:428-552).
struct Store {
    std::vector<uint8_t> header;
    std::vector<Segment> segs;
    std::vector<Path> paths;
    std::vector<Link> links;
    std::vector<Handle> steps;
    std::vector<uint8_t> seq_data;
    std::vector<Span> overlaps;
    std::vector<uint32_t> alignment;
    std::vector<uint8_t> name_data;
    std::vector<uint8_t> optional_data;
    std::vector<uint8_t> line_order;
    View view() const;
};

// A fixed-capacity vector over a region of a mapped file: the reference's SliceVec (file.rs:215-225,
// pool.rs FixedStore).  Only what the parser uses of std::vector.  Pushing beyond the capacity throws
// CapacityError where the reference panics.  Elements are align-1 PODs (a pool starts at any byte).
struct CapacityError {
    int pool;
    uint64_t capacity;
};
template <class T>
class FixedVec {
  public:
    FixedVec() = default;
    FixedVec(T *p, size_t cap, int pool) : p_(p), cap_(cap), pool_(pool) { static_assert(alignof(T) == 1, "pools start at any byte"); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    T *end() { return p_ + n_; }
    void push_back(const T &v) {
        need(1);
        p_[n_++] = v;
    }
    template <class It>
    void insert(T *at, It b, It e) {  // (appends only)
        (void)at;
        need((size_t)(e - b));
        for (; b != e; ++b) p_[n_++] = T{*b};
    }
    template <class It>
    void assign(It b, It e) {
        resize(0);
        insert(end(), b, e);
    }
    void resize(size_t m) {  // (what is given up reads as zeros again, as the fresh file does)
        if (m > n_) need(m - n_);
        if (m > n_) memset((void *)(p_ + n_), 0, (m - n_) * sizeof(T));
        else memset((void *)(p_ + m), 0, (n_ - m) * sizeof(T));
        n_ = m;
    }

  private:
    void need(size_t k) const {
        if (k > cap_ - n_) throw CapacityError{pool_, (uint64_t)cap_};
    }
    T *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    int pool_ = 0;
};

// The reference's FixedGFAStore over a file image laid out by a table of contents (file.rs:227-253).
struct FixedStore {
    FixedVec<uint8_t> header;
    FixedVec<Segment> segs;
    FixedVec<Path> paths;
    FixedVec<Link> links;
    FixedVec<Handle> steps;
    FixedVec<uint8_t> seq_data;
    FixedVec<Span> overlaps;
    FixedVec<AlignOp> alignment;
    FixedVec<uint8_t> name_data;
    FixedVec<uint8_t> optional_data;
    FixedVec<uint8_t> line_order;
    uint64_t pool_len(int ix) const;
};

// namemap.rs:7-33
class NameMap {
  public:
    void insert(uint64_t name, uint32_t id);
    // false where the reference would panic on a missing key
    bool get(uint64_t name, uint32_t *id) const;

  private:
    uint64_t sequential_max_ = 0;
    std::unordered_map<uint64_t, uint32_t> others_;
};

// Parser::parse_mem (parse.rs:77-159).  Returns false and sets `err` exactly
// where the reference panics.  With `stream_mode` it follows Parser::parse_stream
// (parse.rs:24-74, used for stdin) instead: a final line without '\n' is kept, and
// all deferred links are added before all deferred paths.
bool parse_gfa(const uint8_t *buf, size_t n, Store *out, std::string *err, bool stream_mode = false);

// file::view (file.rs:163-213).  `data` must outlive the view.
bool view_flatgfa(const uint8_t *data, size_t n, View *out, std::string *err);
// file::size / file::dump with Toc::full (file.rs:82-98, 290-313)
size_t flatgfa_file_size(const View &v);
void dump_flatgfa(const View &v, uint8_t *buf);
// The preallocated container (`fgfa -m -p`): capacities as parse.rs:176-216 + file.rs:136-158
// estimate them from a GFA text, or as file.rs:117-132 guesses them; the file with those capacities.
bool estimate_toc(const uint8_t *buf, size_t n, uint64_t cap[11], std::string *err);
void guess_toc(uint64_t factor, uint64_t cap[11]);
bool prealloc_file_size(const View &v, const uint64_t cap[11], size_t *total, std::string *err);
// prealloc_translate proper (cli/main.rs:216-248, file.rs:255-272): `file` is the mapped, zero-filled output of
// toc_file_size(cap) bytes.  file::init writes the empty table of contents, the parser pushes straight into the
// file's regions (Parser::for_slice), and the table's lengths are set when it is done.  False, with the table
// still empty, where a pool does not fit its capacity (the reference's panic) or the text does not parse.
bool toc_file_size(const uint64_t cap[11], size_t *total, std::string *err);
bool parse_gfa_prealloc(const uint8_t *buf, size_t n, bool stream_mode, const uint64_t cap[11], uint8_t *file, std::string *err);
void dump_flatgfa_prealloc(const View &v, const uint64_t cap[11], uint8_t *buf);

// print.rs:99-153 (preserved order when line_order is non-empty, else normalized)
bool print_gfa(const View &v, std::string *out, std::string *err);
// see flatgfa_core.cpp: what view_flatgfa checks, and what a host-side walk of the steps needs
bool validate_spans(const View &v, std::string *err);
bool validate_step_ids(const View &v);

// ops/depth.rs:192-197
std::string format_float(double x, int digits);
// ops/depth.rs:67-82
void emit_seg_depth(const View &v, const uint64_t *depth, const uint64_t *uniq, std::string *out);
// ... from 32-bit counts into one malloc'd, NUL-terminated buffer (nullptr: out of memory)
char *emit_seg_depth_u32_malloc(const View &v, const uint32_t *depth, const uint32_t *uniq, size_t *len);
// ops/depth.rs:143-160
void emit_path_depth(const View &v, const uint32_t *path_ids, size_t n, const uint64_t *lengths,
                     const double *means, std::string *out);

// A parsed BED file (flatgfa/src/flatbed.rs:9-25,125-158).
struct BedEntry {
    uint32_t name_start, name_end;
    uint64_t start, end;
};
struct Bed {
    std::vector<uint8_t> name_data;
    std::vector<BedEntry> entries;
};
bool parse_bed(const uint8_t *buf, size_t n, Bed *out, std::string *err);
// Windows{name, start, end, size}.as_bed(), window_depth.rs:22-57
void make_windows(const uint8_t *name, size_t name_len, uint64_t start, uint64_t end, uint64_t size, Bed *out);
// window_depth.rs:69-77
uint64_t path_length(const View &v, uint32_t path);
// weighted_depths + assign_depths, window_depth.rs:84-147 (f64, reference order of operations)
void interval_depth(const View &v, const uint64_t *seg_depth, uint32_t path, const BedEntry *win, size_t n_win, double *out);
// IntervalDepth::emit, window_depth.rs:158-170
void emit_interval_depth(const Bed &bed, const double *depths, std::string *out);
// slow_odgi/slow_odgi/overlap.py:17-32
void emit_overlap(const View &v, const uint32_t *query_ids, size_t n_q, const uint64_t *path_len, const uint8_t *touch,
                  std::string *out);

// The synthetic-graph generator of SURVEY.md 8(d) (spec: oracle/synth.py).
// model 0 = pangenome, 1 = uniform, 2 = chromosome, 3 = haplotype (oracle/synth.py has the spec).
void synth_store(uint64_t seed, uint32_t S, uint32_t P, uint32_t L, int model, bool with_seq, Store *out);

// A read-only memory-mapped file (memfile.rs:7-10).
struct MappedFile {
    const uint8_t *data = nullptr;
    size_t size = 0;
    bool open(const char *path, std::string *err);
    ~MappedFile();
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

}  // namespace fgfa
