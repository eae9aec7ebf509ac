// tblload.cpp — native loader for dbgen-style delimited text tables (host code, no GPU).
//
// Replaces the reference's pure-Python loader `read_csv_to_np_array`
// (reference src/sdqlpy/sdql_lib.py:69-115), which is a csv.reader loop appending to Python lists —
// minutes at SF=10.  Same result contract: one array per schema column, int -> int64,
// float -> float64, date "yyyy-mm-dd" -> yyyymmdd int64, string(n) -> n UCS4 code units
// (numpy '<U n': truncated to n, zero padded); field i of a line belongs to column i and the
// trailing empty field left by a line-terminating delimiter fills the schema's *_NA column.
//
// Two passes over an mmap of the file: line starts (parallel chunks), then a parallel parse straight
// into the caller's column buffers.  Anything this parser does not reproduce exactly as Python
// would (quoted fields, digit separators, "nan"/"inf", invalid UTF-8, ragged lines) is reported
// with a status code and the caller falls back to its general csv path for that file.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

enum { T_INT = 0, T_FLOAT = 1, T_DATE = 2, T_STR = 3 };
enum { OK = 0, ERR_IO = 1, ERR_UNSUPPORTED = 2, ERR_RAGGED = 3, ERR_ARG = 4 };

struct Table {
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;
    char delim = '|';
    std::vector<size_t> line_start;      // offset of every line; line i ends at line_start[i+1] (exclusive of '\n')
    std::string error;
};

int set_error(Table* t, int code, const std::string& msg) { if (t) t->error = msg; return code; }

bool parse_i64(const char* b, const char* e, int64_t* out) {
    if (b == e) return false;
    auto r = std::from_chars(b + (*b == '+' ? 1 : 0), e, *out);
    return r.ec == std::errc() && r.ptr == e && !(*b == '+' && b + 1 < e && b[1] == '-');
}

// "yyyy-mm-dd" -> yyyymmdd, as int("".join(cell.split("-"))): any '-' separated digit groups
bool parse_date(const char* b, const char* e, int64_t* out) {
    char buf[32];
    size_t n = 0;
    for (const char* p = b; p < e; ++p) {
        if (*p == '-') continue;
        if (*p < '0' || *p > '9' || n >= sizeof(buf) - 1) return false;
        buf[n++] = *p;
    }
    if (n == 0) return false;
    auto r = std::from_chars(buf, buf + n, *out);
    return r.ec == std::errc() && r.ptr == buf + n;
}

// plain decimal / exponent notation only; everything else Python's float() accepts is left to the caller
bool parse_f64(const char* b, const char* e, double* out) {
    if (b == e) return false;
    for (const char* p = b; p < e; ++p) {
        const char c = *p;
        if (!((c >= '0' && c <= '9') || c == '.' || c == '-' || c == '+' || c == 'e' || c == 'E')) return false;
    }
    const char* s = b + (*b == '+' ? 1 : 0);
    auto r = std::from_chars(s, e, *out);                                // correctly rounded, like Python's float()
    return r.ec == std::errc() && r.ptr == e;
}

// UTF-8 -> UCS4, at most `width` code units, zero padded.  false = malformed input.
bool parse_str(const char* b, const char* e, uint32_t* out, int width) {
    int n = 0;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(b);
    const unsigned char* end = reinterpret_cast<const unsigned char*>(e);
    while (p < end) {
        uint32_t cp; int extra;
        const unsigned char c = *p++;
        if (c < 0x80) { cp = c; extra = 0; }
        else if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; extra = 1; }
        else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; extra = 2; }
        else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; extra = 3; }
        else return false;
        if (end - p < extra) return false;
        for (int k = 0; k < extra; ++k) { if ((*p & 0xC0) != 0x80) return false; cp = (cp << 6) | (*p++ & 0x3F); }
        if ((extra == 1 && cp < 0x80) || (extra == 2 && cp < 0x800) || (extra == 3 && cp < 0x10000) || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        if (cp == 0) return false;                                        // numpy would end the string here: leave it to the general path
        if (n < width) out[n++] = cp;
    }
    for (; n < width; ++n) out[n] = 0;
    return true;
}

}  // namespace

extern "C" {

// Open + index the file.  *nrows = number of lines (a final line without '\n' counts; an empty
// file has 0 rows).  Quote characters anywhere make the file "unsupported" (csv.reader would
// interpret them).
int sdql_tbl_open(const char* path, char delim, int nthreads, void** handle, int64_t* nrows) {
    if (!path || !handle || !nrows) return ERR_ARG;
    Table* t = new Table();
    *handle = t;
    t->delim = delim;
    t->fd = ::open(path, O_RDONLY);
    if (t->fd < 0) return set_error(t, ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(t->fd, &st) != 0) return set_error(t, ERR_IO, "fstat failed");
    t->size = (size_t)st.st_size;
    if (t->size) {
        void* m = mmap(nullptr, t->size, PROT_READ, MAP_PRIVATE, t->fd, 0);
        if (m == MAP_FAILED) return set_error(t, ERR_IO, "mmap failed");
        t->data = static_cast<const char*>(m);
        madvise(m, t->size, MADV_SEQUENTIAL);
    }
    const int nt = std::max(1, std::min(nthreads, (int)(t->size / (1 << 20)) + 1));
    std::vector<std::vector<size_t>> starts((size_t)nt);
    std::atomic<int> bad{0};
    auto work = [&](int k) {
        const size_t lo = t->size * (size_t)k / (size_t)nt, hi = t->size * (size_t)(k + 1) / (size_t)nt;
        std::vector<size_t>& v = starts[(size_t)k];
        if (k == 0 && t->size) v.push_back(0);
        for (size_t i = lo; i < hi; ++i) {
            const char c = t->data[i];
            if (c == '\n') { if (i + 1 < t->size) v.push_back(i + 1); }
            else if (c == '"' || c == '\r') bad.store(1);
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nt; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& x : th) x.join();
    if (bad.load()) return set_error(t, ERR_UNSUPPORTED, "quote or carriage-return characters: needs the general csv path");
    size_t total = 0;
    for (auto& v : starts) total += v.size();
    t->line_start.reserve(total + 1);
    for (auto& v : starts) t->line_start.insert(t->line_start.end(), v.begin(), v.end());
    t->line_start.push_back(t->size + ((t->size && t->data[t->size - 1] == '\n') ? 0 : 1));   // sentinel: "one past the newline" of the last line
    *nrows = (int64_t)total;
    return OK;
}

// Parse every line into the column buffers: out[c] has nrows elements of int64 / double / width[c]
// uint32 code units; out[c] == NULL skips the column.  Lines must have exactly ncols fields.
int sdql_tbl_parse(void* handle, int ncols, const int* types, const int* widths, void* const* out, int nthreads) {
    Table* t = static_cast<Table*>(handle);
    if (!t || ncols < 1 || !types || !widths || !out) return ERR_ARG;
    const size_t nrows = t->line_start.size() - 1;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, nthreads), nrows / 4096 + 1));
    std::atomic<int> status{OK};
    std::vector<std::string> errors((size_t)nt);
    auto work = [&](int k) {
        const size_t r0 = nrows * (size_t)k / (size_t)nt, r1 = nrows * (size_t)(k + 1) / (size_t)nt;
        for (size_t r = r0; r < r1 && status.load(std::memory_order_relaxed) == OK; ++r) {
            const char* p = t->data + t->line_start[r];
            const char* end = t->data + t->line_start[r + 1] - 1;            // the '\n' (or one past the data for an unterminated last line)
            int c = 0;
            for (;;) {
                const char* q = static_cast<const char*>(memchr(p, t->delim, (size_t)(end - p)));
                const char* fe = q ? q : end;
                if (c >= ncols) { status.store(ERR_RAGGED); errors[(size_t)k] = "line " + std::to_string(r + 1) + ": more fields than schema columns"; return; }
                bool ok = true;
                if (out[c]) {
                    switch (types[c]) {
                        case T_INT: ok = parse_i64(p, fe, static_cast<int64_t*>(out[c]) + r); break;
                        case T_DATE: ok = parse_date(p, fe, static_cast<int64_t*>(out[c]) + r); break;
                        case T_FLOAT: ok = parse_f64(p, fe, static_cast<double*>(out[c]) + r); break;
                        case T_STR: ok = parse_str(p, fe, static_cast<uint32_t*>(out[c]) + r * (size_t)widths[c], widths[c]); break;
                        default: ok = false;
                    }
                }
                if (!ok) { status.store(ERR_UNSUPPORTED); errors[(size_t)k] = "line " + std::to_string(r + 1) + ", field " + std::to_string(c + 1) + ": not a plain value"; return; }
                ++c;
                if (!q) break;
                p = q + 1;
            }
            if (c != ncols) { status.store(ERR_RAGGED); errors[(size_t)k] = "line " + std::to_string(r + 1) + ": " + std::to_string(c) + " fields, schema has " + std::to_string(ncols); return; }
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nt; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& x : th) x.join();
    if (status.load() != OK) for (auto& e : errors) if (!e.empty()) { t->error = e; break; }
    return status.load();
}

// Dictionary encoding of a fixed-width UCS4 column (numpy '<U width'): codes[i] = index of row i's
// text in the sorted list of distinct texts, for columns with at most max_distinct of them (text
// group keys and low-cardinality string predicates are then integer work on the device).
// Returns the number of distinct values, or -1 when there are more than max_distinct.
// dict_out: max_distinct * width code units, filled in sorted (code) order.
int64_t sdql_dict_encode(const uint32_t* data, int64_t nrows, int width, int max_distinct, int nthreads,
                         int64_t* codes, uint32_t* dict_out) {
    if (!data || nrows < 0 || width < 1 || max_distinct < 1 || !codes || !dict_out) return -2;
    const size_t cap = 1u << 14;                                         // open addressing, >= 4 x max_distinct
    if ((size_t)max_distinct * 4 > cap) return -2;
    auto hash_of = [width](const uint32_t* s) { uint64_t h = 1469598103934665603ull; for (int k = 0; k < width; ++k) { h ^= s[k]; h *= 1099511628211ull; } return h; };
    // pass 1 (parallel): every thread collects the distinct texts of ITS slice of the rows in a table of its own (a serial scan of
    // 60 M rows — hash, probe, compare per row — was 100 ms of a 126 ms call: most of what Q1's first run at SF=10 cost); the slices'
    // lists are then merged in slice order, so "first row with that text" is the first row of the whole column, as before
    const int nt1 = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(1, nthreads), nrows / 65536 + 1));
    std::vector<std::vector<int64_t>> local_first((size_t)nt1);
    std::atomic<int> too_many{0};
    auto collect = [&](int k) {
        const int64_t r0 = nrows * k / nt1, r1 = nrows * (k + 1) / nt1;
        std::vector<int32_t> slots(cap, -1);
        std::vector<int64_t>& mine = local_first[(size_t)k];
        for (int64_t r = r0; r < r1; ++r) {
            if ((r & 0xFFFF) == 0 && too_many.load(std::memory_order_relaxed)) return;
            const uint32_t* s = data + (size_t)r * width;
            size_t h = hash_of(s) & (cap - 1);
            for (;;) {
                const int32_t e = slots[h];
                if (e < 0) {
                    if ((int)mine.size() == max_distinct) { too_many.store(1); return; }
                    slots[h] = (int32_t)mine.size(); mine.push_back(r);
                    break;
                }
                if (std::memcmp(data + (size_t)mine[(size_t)e] * width, s, (size_t)width * 4) == 0) break;
                h = (h + 1) & (cap - 1);
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int k = 1; k < nt1; ++k) th.emplace_back(collect, k);
        collect(0);
        for (auto& x : th) x.join();
    }
    if (too_many.load()) return -1;
    std::vector<int32_t> slot_row(cap, -1);                              // slot -> index into first_rows
    std::vector<int64_t> first_rows;
    for (int k = 0; k < nt1; ++k)
        for (int64_t r : local_first[(size_t)k]) {
            const uint32_t* s = data + (size_t)r * width;
            size_t h = hash_of(s) & (cap - 1);
            for (;;) {
                const int32_t e = slot_row[h];
                if (e < 0) {
                    if ((int)first_rows.size() == max_distinct) return -1;
                    slot_row[h] = (int32_t)first_rows.size(); first_rows.push_back(r);
                    break;
                }
                if (std::memcmp(data + (size_t)first_rows[(size_t)e] * width, s, (size_t)width * 4) == 0) break;
                h = (h + 1) & (cap - 1);
            }
        }
    const int nd = (int)first_rows.size();
    // sorted order = numpy's: lexicographic on code units (UCS4 values), zero padded
    std::vector<int> order((size_t)nd), rank((size_t)nd);
    for (int i = 0; i < nd; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        const uint32_t* x = data + (size_t)first_rows[(size_t)a] * width; const uint32_t* y = data + (size_t)first_rows[(size_t)b] * width;
        for (int k = 0; k < width; ++k) if (x[k] != y[k]) return x[k] < y[k];
        return false;
    });
    for (int i = 0; i < nd; ++i) { rank[(size_t)order[(size_t)i]] = i; std::memcpy(dict_out + (size_t)i * width, data + (size_t)first_rows[(size_t)order[(size_t)i]] * width, (size_t)width * 4); }
    // pass 2 (parallel): codes
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(1, nthreads), nrows / 65536 + 1));
    auto work = [&](int k) {
        const int64_t r0 = nrows * k / nt, r1 = nrows * (k + 1) / nt;
        for (int64_t r = r0; r < r1; ++r) {
            const uint32_t* s = data + (size_t)r * width;
            size_t h = hash_of(s) & (cap - 1);
            for (;;) {
                const int32_t e = slot_row[h];
                if (std::memcmp(data + (size_t)first_rows[(size_t)e] * width, s, (size_t)width * 4) == 0) { codes[r] = rank[(size_t)e]; break; }
                h = (h + 1) & (cap - 1);
            }
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nt; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& x : th) x.join();
    return nd;
}

const char* sdql_tbl_error(void* handle) { return handle ? static_cast<Table*>(handle)->error.c_str() : "no handle"; }

void sdql_tbl_close(void* handle) {
    Table* t = static_cast<Table*>(handle);
    if (!t) return;
    if (t->data) munmap(const_cast<char*>(t->data), t->size);
    if (t->fd >= 0) ::close(t->fd);
    delete t;
}

}  // extern "C"
