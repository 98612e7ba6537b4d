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
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <functional>
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

// fn(0 .. n-1) on the library's host pool (dimn.hip: host_pool().run; fn(0) on the caller)
static void csv_parallel(int n, const std::function<void(int)>& fn);

// The non-empty lines of the file: [s[i], e[i]) without the line terminator ("\n" or "\r\n").  Found on several threads with memchr (the
// byte loops this replaces -- one of them single-threaded over the whole file, three times per read -- were 1.5 of the 1.9 s that
// read_csv took at 50k x 20k, round 5); `quoted`: the file holds a double quote somewhere (then it is not the plain matrix this reader takes).
struct CsvLines { std::vector<size_t> s, e; bool quoted = false; };
static void csv_lines(const char* p, size_t n, CsvLines& L, bool look_for_quotes) {
    const int nt = n > (8u << 20) ? (int)csv_threads() : 1;
    std::vector<std::vector<size_t>> part((size_t)nt);
    std::vector<int> quote((size_t)nt, 0);
    csv_parallel(nt, [&](int t) {
        const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt;
        if (look_for_quotes && memchr(p + a, '"', b - a)) quote[(size_t)t] = 1;
        const char* q = p + a;
        while (q < p + b) {
            const char* nl = (const char*)memchr(q, '\n', (size_t)(p + b - q));
            if (!nl) break;
            part[(size_t)t].push_back((size_t)(nl - p));
            q = nl + 1;
        }
    });
    L.s.clear(); L.e.clear(); L.quoted = false;
    for (int q : quote) L.quoted |= q != 0;
    size_t start = 0;
    auto line = [&](size_t a, size_t b) {                        // [a, b): one line without its "\n"
        if (b > a && p[b - 1] == '\r') --b;
        if (b > a) { L.s.push_back(a); L.e.push_back(b); }
    };
    for (auto& v : part)
        for (size_t nl : v) { line(start, nl); start = nl + 1; }
    if (start < n) line(start, n);                               // a last line without a terminator
}

// pass 1: shape and label bytes.  Returns 0, or -4 (DIMN_ERR_UNSUP) when the file is not a plain count matrix.
static int csv_scan(const char* path, int64_t* n_rows, int64_t* n_cols, int64_t* label_bytes, std::string& err) {
    CsvMap m;
    if (!m.open(path)) { err = "cannot open or map the file"; return -1; }
    CsvLines L;
    csv_lines(m.p, m.n, L, true);
    if (L.quoted) { err = "quoted fields"; return -4; }
    if (L.s.size() < 2) { err = "no data rows"; return -4; }
    int64_t cols = 0;
    for (size_t i = L.s[0]; i < L.e[0]; ++i) cols += m.p[i] == ',';
    if (cols < 1) { err = "no data columns"; return -4; }
    int64_t bytes = (int64_t)(L.e[0] - L.s[0]) + 1;
    for (size_t r = 1; r < L.s.size(); ++r) {
        const char* c = (const char*)memchr(m.p + L.s[r], ',', L.e[r] - L.s[r]);
        if (!c) { err = "a row without fields"; return -4; }
        bytes += (int64_t)(c - (m.p + L.s[r])) + 1;
    }
    *n_rows = (int64_t)L.s.size() - 1; *n_cols = cols; *label_bytes = bytes + 16;
    return 0;
}

// pass 2: values[n_rows][n_cols] (int64), labels = index name, column labels, row labels, each NUL-terminated
static int csv_read(const char* path, int64_t n_rows, int64_t n_cols, int64_t* values, char* labels, int64_t label_cap, std::string& err) {
    CsvMap m;
    if (!m.open(path)) { err = "cannot open or map the file"; return -1; }
    CsvLines L;
    csv_lines(m.p, m.n, L, false);
    if ((int64_t)L.s.size() != n_rows + 1) { err = "the file changed between the two passes"; return -1; }
    char* lp = labels; char* const lend = labels + label_cap;
    auto put = [&](const char* a, const char* b) -> bool {
        if (lp + (b - a) + 1 > lend) return false;
        memcpy(lp, a, (size_t)(b - a)); lp += b - a; *lp++ = 0;
        return true;
    };
    {   // header: index name, then the column labels
        const char* a = m.p + L.s[0]; const char* e = m.p + L.e[0];
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
    const int nt = n_rows * n_cols > (1 << 20) ? (int)std::min<int64_t>(csv_threads(), n_rows) : 1;
    std::vector<int> bad((size_t)nt, 0);
    csv_parallel(nt, [&](int t) {
        const int64_t r0 = n_rows * t / nt, r1 = n_rows * (t + 1) / nt;
        for (int64_t r = r0; r < r1 && !bad[(size_t)t]; ++r) {
            const char* a = m.p + L.s[(size_t)r + 1]; const char* e = m.p + L.e[(size_t)r + 1];
            const char* c = (const char*)memchr(a, ',', (size_t)(e - a));
            if (!c) { bad[(size_t)t] = 1; break; }
            row_lab_end[(size_t)r] = c;
            const char* q = c + 1;
            int64_t* out = values + r * n_cols;
            for (int64_t j = 0; j < n_cols; ++j) {
                bool neg = false;
                if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
                if (q >= e || *q < '0' || *q > '9') { bad[(size_t)t] = 1; break; }
                uint64_t v = 0; int digits = 0;
                while (q < e && *q >= '0' && *q <= '9') { v = v * 10 + (uint64_t)(*q - '0'); ++q; ++digits; }
                if (digits > 18) { bad[(size_t)t] = 1; break; }
                out[j] = neg ? -(int64_t)v : (int64_t)v;
                if (j + 1 < n_cols) { if (q >= e || *q != ',') { bad[(size_t)t] = 1; break; } ++q; }
                else if (q != e) { bad[(size_t)t] = 1; break; }
            }
        }
    });
    for (int b : bad) if (b) { err = "a field that is not an integer literal (or a ragged row)"; return -4; }
    for (int64_t r = 0; r < n_rows; ++r)
        if (!put(m.p + L.s[(size_t)r + 1], row_lab_end[(size_t)r])) { err = "label buffer too small"; return -1; }
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

// DataFrame.to_csv(path): header "<index name>,<col>,<col>...", rows "<label>,<v>,<v>..."; labels NUL-separated.
// Row blocks are formatted on the host pool into one of two buffer sets while a writer thread streams the other set to the file in
// order: the text of a 50k x 20k frame is 9.1 GB, and copying it into the page cache -- one thread, the inode's write lock admits no
// second one: pwrite from 48 threads at computed offsets was measured SLOWER, 3.65 vs 2.0 s -- now runs beside the formatting.
static int csv_write(const char* path, const double* values, int64_t n_rows, int64_t n_cols, const char* index_name, const char* col_labels,
                     const char* row_labels, std::string& err) {
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { err = "cannot create the file"; return -1; }
    std::string head(index_name ? index_name : "");
    const char* c = col_labels;
    for (int64_t j = 0; j < n_cols; ++j) { head += ','; head += c; c += strlen(c) + 1; }
    head += '\n';
    int werr = 0;
    auto write_all = [&](const char* p, size_t len) -> bool {
        while (len) {
            const ssize_t w = ::write(fd, p, len);
            if (w < 0 && errno == EINTR) continue;            // a signal handler ran (CPython installs its own without SA_RESTART): not a failure
            if (w <= 0) { werr = w < 0 ? errno : EIO; return false; }      // (errno is per thread: kept for the caller's message)
            p += w; len -= (size_t)w;
        }
        return true;
    };
    bool ok = write_all(head.data(), head.size());
    std::vector<const char*> rl((size_t)n_rows);
    c = row_labels;
    for (int64_t r = 0; r < n_rows; ++r) { rl[(size_t)r] = c; c += strlen(c) + 1; }
    const int nt = n_rows * n_cols > (1 << 18) ? (int)csv_threads() : 1;
    const int64_t rows_per_task = std::max<int64_t>(1, (int64_t)(4u << 20) / std::max<int64_t>(1, n_cols * 8));   // ~4 MB of text per task
    std::vector<std::string> buf[2] = {std::vector<std::string>((size_t)nt), std::vector<std::string>((size_t)nt)};
    std::thread writer;
    bool wrote = true;                                            // result of the writer's last batch (read after join)
    int set = 0;
    for (int64_t r0 = 0; r0 < n_rows && ok; r0 += rows_per_task * nt, set ^= 1) {
        std::vector<std::string>& cur = buf[set];
        csv_parallel(nt, [&](int t) {
            const int64_t a = r0 + (int64_t)t * rows_per_task, b = std::min(n_rows, a + rows_per_task);
            std::string& s = cur[(size_t)t];
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
        if (writer.joinable()) { writer.join(); ok = ok && wrote; }         // the other set is on disk: the next batch may be formatted into it
        writer = std::thread([&, set] {
            sigset_t all;                                         // signals belong to the interpreter's main thread, not to this helper
            sigfillset(&all);
            pthread_sigmask(SIG_BLOCK, &all, nullptr);
            bool good = true;
            for (const std::string& s : buf[set]) good = good && (s.empty() || write_all(s.data(), s.size()));
            wrote = good;
        });
    }
    if (writer.joinable()) { writer.join(); ok = ok && wrote; }
    if (::close(fd) != 0) { ok = false; if (!werr) werr = errno; }
    if (!ok) {
        err = std::string("write failed (") + strerror(werr) + "); the partial file was removed";
        ::unlink(path);                                           // never leave a truncated CSV behind under the final name
        return -1;
    }
    return 0;
}
