// dimn_csv.h -- the CSV edges of the drop-in: the reference's deepImpute.py:13
// `pd.read_csv(inputFile, index_col=0)` and :35 `imputed.to_csv(output)` as multi-threaded host code.
//
// At 50k cells x 20k genes the two pandas calls take minutes (1e9 fields) around five seconds of GPU work, so the CLI
// gets its own reader / writer -- for exactly the shape the tool is specified for: a rectangular matrix of RAW COUNTS
// (inspect_data rejects anything else, multinet.py:43-63) with unquoted labels.
//   reader: every data field must be an integer literal ([+-]?digits).  Integers are exact in any parser, so the frame
//           equals pandas' (int64).  Anything else -- a decimal point, an exponent, an empty field, a quote, a ragged
//           row -- makes the call return DIMN_ERR_UNSUP and the Python side falls back to pd.read_csv itself (pandas'
//           own float parser is not correctly rounded; reproducing its last bit is not worth a second parser).
//   writer: float64 in the shortest round-trip form with Python's repr() switch-over (scientific iff the decimal
//           exponent is < -4 or >= 16, ".0" on integral values) -- byte-identical to DataFrame.to_csv for finite
//           values; NaN -> "" and +-inf -> "inf"/"-inf" as pandas writes them.
// Host-only; no HIP call in this file.
#pragma once
#include <charconv>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

struct CsvMap {
    const char* p = nullptr; size_t n = 0; int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        (void)madvise(m, n, MADV_SEQUENTIAL);
        p = (const char*)m;
        return true;
    }
    ~CsvMap() { if (p && n) munmap((void*)p, n); if (fd >= 0) ::close(fd); }
};

static unsigned csv_threads() {
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hw ? hw / 2 : 8u, 48u));
}

// line starts of the file (offsets of the first byte of every non-empty line), found on several threads
static void csv_line_starts(const char* p, size_t n, std::vector<size_t>& starts) {
    const unsigned nt = n > (8u << 20) ? csv_threads() : 1;
    std::vector<std::vector<size_t>> part(nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            const size_t a = n * t / nt, b = n * (t + 1) / nt;
            for (size_t i = a; i < b; ++i)
                if (p[i] == '\n' && i + 1 < n) part[t].push_back(i + 1);
        });
    for (auto& x : th) x.join();
    starts.clear();
    if (n) starts.push_back(0);
    for (auto& v : part) starts.insert(starts.end(), v.begin(), v.end());
    // drop empty lines (a trailing "\n\n" or "\r\n" tails)
    std::vector<size_t> keep;
    for (size_t s : starts) {
        size_t e = s;
        while (e < n && p[e] != '\n') ++e;
        size_t len = e - s;
        if (len && p[s + len - 1] == '\r') --len;
        if (len) keep.push_back(s);
    }
    starts.swap(keep);
}
static inline size_t csv_line_end(const char* p, size_t n, size_t s) {      // exclusive, without "\r"
    size_t e = s;
    while (e < n && p[e] != '\n') ++e;
    if (e > s && p[e - 1] == '\r') --e;
    return e;
}

// pass 1: shape and label bytes.  Returns 0, or -4 (DIMN_ERR_UNSUP) when the file is not a plain count matrix.
static int csv_scan(const char* path, int64_t* n_rows, int64_t* n_cols, int64_t* label_bytes, std::string& err) {
    CsvMap m;
    if (!m.open(path)) { err = "cannot open or map the file"; return -1; }
    if (memchr(m.p, '"', m.n)) { err = "quoted fields"; return -4; }
    std::vector<size_t> ls;
    csv_line_starts(m.p, m.n, ls);
    if (ls.size() < 2) { err = "no data rows"; return -4; }
    const size_t he = csv_line_end(m.p, m.n, ls[0]);
    int64_t cols = 0;
    for (size_t i = ls[0]; i < he; ++i) cols += m.p[i] == ',';
    if (cols < 1) { err = "no data columns"; return -4; }
    int64_t bytes = (int64_t)(he - ls[0]) + 1;
    for (size_t r = 1; r < ls.size(); ++r) {
        const char* c = (const char*)memchr(m.p + ls[r], ',', csv_line_end(m.p, m.n, ls[r]) - ls[r]);
        if (!c) { err = "a row without fields"; return -4; }
        bytes += (int64_t)(c - (m.p + ls[r])) + 1;
    }
    *n_rows = (int64_t)ls.size() - 1; *n_cols = cols; *label_bytes = bytes + 16;
    return 0;
}

// pass 2: values[n_rows][n_cols] (int64), labels = index name, column labels, row labels, each NUL-terminated
static int csv_read(const char* path, int64_t n_rows, int64_t n_cols, int64_t* values, char* labels, int64_t label_cap, std::string& err) {
    CsvMap m;
    if (!m.open(path)) { err = "cannot open or map the file"; return -1; }
    std::vector<size_t> ls;
    csv_line_starts(m.p, m.n, ls);
    if ((int64_t)ls.size() != n_rows + 1) { err = "the file changed between the two passes"; return -1; }
    char* lp = labels; char* const lend = labels + label_cap;
    auto put = [&](const char* a, const char* b) -> bool {
        if (lp + (b - a) + 1 > lend) return false;
        memcpy(lp, a, (size_t)(b - a)); lp += b - a; *lp++ = 0;
        return true;
    };
    {   // header: index name, then the column labels
        const char* a = m.p + ls[0]; const char* e = m.p + csv_line_end(m.p, m.n, ls[0]);
        int64_t cnt = 0;
        while (true) {
            const char* c = (const char*)memchr(a, ',', (size_t)(e - a));
            const char* b = c ? c : e;
            if (!put(a, b)) { err = "label buffer too small"; return -1; }
            ++cnt;
            if (!c) break;
            a = c + 1;
        }
        if (cnt != n_cols + 1) { err = "header width changed"; return -1; }
    }
    std::vector<const char*> row_lab_end((size_t)n_rows);
    const unsigned nt = n_rows * n_cols > (1 << 20) ? csv_threads() : 1;
    std::vector<int> bad(nt, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            const int64_t r0 = n_rows * t / nt, r1 = n_rows * (t + 1) / nt;
            for (int64_t r = r0; r < r1 && !bad[t]; ++r) {
                const char* a = m.p + ls[(size_t)r + 1]; const char* e = m.p + csv_line_end(m.p, m.n, ls[(size_t)r + 1]);
                const char* c = (const char*)memchr(a, ',', (size_t)(e - a));
                if (!c) { bad[t] = 1; break; }
                row_lab_end[(size_t)r] = c;
                const char* q = c + 1;
                int64_t* out = values + r * n_cols;
                for (int64_t j = 0; j < n_cols; ++j) {
                    bool neg = false;
                    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
                    if (q >= e || *q < '0' || *q > '9') { bad[t] = 1; break; }
                    uint64_t v = 0; int digits = 0;
                    while (q < e && *q >= '0' && *q <= '9') { v = v * 10 + (uint64_t)(*q - '0'); ++q; ++digits; }
                    if (digits > 18) { bad[t] = 1; break; }
                    out[j] = neg ? -(int64_t)v : (int64_t)v;
                    if (j + 1 < n_cols) { if (q >= e || *q != ',') { bad[t] = 1; break; } ++q; }
                    else if (q != e) { bad[t] = 1; break; }
                }
            }
        });
    for (auto& x : th) x.join();
    for (int b : bad) if (b) { err = "a field that is not an integer literal (or a ragged row)"; return -4; }
    for (int64_t r = 0; r < n_rows; ++r)
        if (!put(m.p + ls[(size_t)r + 1], row_lab_end[(size_t)r])) { err = "label buffer too small"; return -1; }
    return 0;
}

// repr(float) of Python / numpy: shortest round-trip digits; scientific iff exp10 < -4 or >= 16; ".0" on integral values
static inline char* csv_fmt_double(char* o, double x) {
    if (x != x) return o;                                        // NaN -> "" (na_rep)
    if (x == __builtin_inf()) { memcpy(o, "inf", 3); return o + 3; }
    if (x == -__builtin_inf()) { memcpy(o, "-inf", 4); return o + 4; }
    if (x == 0.0) { if (__builtin_signbit(x)) *o++ = '-'; memcpy(o, "0.0", 3); return o + 3; }
    char s[40];
    const auto r = std::to_chars(s, s + sizeof s - 1, x, std::chars_format::scientific);     // d[.ddd]e[+-]XX, shortest
    *r.ptr = 0;
    const char* p = s;
    if (*p == '-') *o++ = *p++;
    const char* e = p;
    while (*e != 'e') ++e;
    const int ex = atoi(e + 1);
    if (ex < -4 || ex >= 16) { memcpy(o, p, (size_t)(r.ptr - p)); return o + (r.ptr - p); }
    char dig[24]; int nd = 0;
    for (const char* q = p; q < e; ++q) if (*q != '.') dig[nd++] = *q;
    if (ex >= 0) {
        for (int i = 0; i <= ex; ++i) *o++ = i < nd ? dig[i] : '0';
        *o++ = '.';
        if (nd > ex + 1) { memcpy(o, dig + ex + 1, (size_t)(nd - ex - 1)); o += nd - ex - 1; }
        else *o++ = '0';
    } else {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -ex - 1; ++i) *o++ = '0';
        memcpy(o, dig, (size_t)nd); o += nd;
    }
    return o;
}

// DataFrame.to_csv(path): header "<index name>,<col>,<col>...", rows "<label>,<v>,<v>..."; labels NUL-separated
static int csv_write(const char* path, const double* values, int64_t n_rows, int64_t n_cols, const char* index_name, const char* col_labels,
                     const char* row_labels, std::string& err) {
    FILE* f = fopen(path, "wb");
    if (!f) { err = "cannot create the file"; return -1; }
    std::string head(index_name ? index_name : "");
    const char* c = col_labels;
    for (int64_t j = 0; j < n_cols; ++j) { head += ','; head += c; c += strlen(c) + 1; }
    head += '\n';
    bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();
    std::vector<const char*> rl((size_t)n_rows);
    c = row_labels;
    for (int64_t r = 0; r < n_rows; ++r) { rl[(size_t)r] = c; c += strlen(c) + 1; }
    const unsigned nt = n_rows * n_cols > (1 << 18) ? csv_threads() : 1;
    const int64_t rows_per_task = std::max<int64_t>(1, (int64_t)(4u << 20) / std::max<int64_t>(1, n_cols * 8));   // ~4 MB of text per task
    std::vector<std::string> buf(nt);
    for (int64_t r0 = 0; r0 < n_rows && ok; r0 += rows_per_task * nt) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                const int64_t a = r0 + (int64_t)t * rows_per_task, b = std::min(n_rows, a + rows_per_task);
                std::string& s = buf[t];
                s.clear();
                if (a >= b) return;
                s.resize((size_t)(b - a) * (size_t)(n_cols * 26 + 64) + 1024);
                char* o = &s[0];
                for (int64_t r = a; r < b; ++r) {
                    const size_t ll = strlen(rl[(size_t)r]);
                    if ((size_t)(&s[0] + s.size() - o) < ll + (size_t)n_cols * 26 + 8) {           // very long labels
                        const size_t used = (size_t)(o - &s[0]);
                        s.resize(s.size() + ll + (size_t)n_cols * 26 + 1024);
                        o = &s[0] + used;
                    }
                    memcpy(o, rl[(size_t)r], ll); o += ll;
                    const double* v = values + r * n_cols;
                    for (int64_t j = 0; j < n_cols; ++j) { *o++ = ','; o = csv_fmt_double(o, v[j]); }
                    *o++ = '\n';
                }
                s.resize((size_t)(o - &s[0]));
            });
        for (auto& x : th) x.join();
        for (unsigned t = 0; t < nt && ok; ++t) ok = fwrite(buf[t].data(), 1, buf[t].size(), f) == buf[t].size();
    }
    if (fclose(f) != 0) ok = false;
    if (!ok) { err = "write failed"; return -1; }
    return 0;
}
