// sparse_matrix.hpp -- host data model of the MI355X merge-based CsrMV drivers.
//
// Reproduces the BEHAVIOUR of the reference's sparse_matrix.h (the CSR input
// layout north_star says to keep, plus its Matrix Market reader, synthetic
// generators, statistics and histogram) with a different design:
//   * COO is structure-of-arrays (row[], col[], val[]) instead of an array of
//     tuples, and is built straight into std::vector storage;
//   * COO -> CSR is a stable counting sort by row followed by an independent,
//     OpenMP-parallel stable sort by column inside each row -- the same order
//     as the reference's single-threaded std::stable_sort by (row, col)
//     (sparse_matrix.h:636-643,676; duplicates kept), but O(nnz) + parallel,
//     which is what SURVEY.md 8(f) N2 asks for at corpus scale;
//   * no MKL / libnuma allocation paths (sparse_matrix.h:679-699).
// Each function cites the reference lines whose behaviour it matches; the
// parity tests (tests/test_host_model.py) compare against golden vectors
// produced by the reference's own header.
#pragma once

#include <omp.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

namespace mspmv_host {

/// Wall-clock seconds of the last ingest, by phase (drivers print them with --timing).
struct IngestTimes { double read_s = 0, parse_s = 0, convert_s = 0, cache_save_s = 0, cache_load_s = 0; };
inline IngestTimes &ingest_times() { static IngestTimes t; return t; }

/// Row-length statistics (reference GraphStats, sparse_matrix.h:59-107).
struct GraphStats {
    int num_rows = 0, num_cols = 0, num_nonzeros = 0;
    double row_length_mean = 0, row_length_std_dev = 0, row_length_variation = 0, row_length_skewness = 0;

    /// Labelled (human) or CSV form; byte-for-byte the reference's formats
    /// (sparse_matrix.h:72-106).
    void Display(bool show_labels, FILE *out = stdout) const
    {
        if (show_labels)
            fprintf(out,
                    "\n"
                    "\t num_rows: %d\n"
                    "\t num_cols: %d\n"
                    "\t num_nonzeros: %d\n"
                    "\t row_length_mean: %.5f\n"
                    "\t row_length_std_dev: %.5f\n"
                    "\t row_length_variation: %.5f\n"
                    "\t row_length_skewness: %.5f\n",
                    num_rows, num_cols, num_nonzeros, row_length_mean, row_length_std_dev, row_length_variation,
                    row_length_skewness);
        else
            fprintf(out, "%d, %d, %d, %.5f, %.5f, %.5f, %.5f, ", num_rows, num_cols, num_nonzeros, row_length_mean,
                    row_length_std_dev, row_length_variation, row_length_skewness);
    }
};

struct MarketError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

/// Coordinate-format matrix in emission order (not necessarily sorted).
/// std::vector whose resize() leaves trivially constructible elements uninitialised: the multi-gigabyte arrays of the
/// ingest path are then first touched -- and their pages faulted in -- by the parallel loops that fill them, not zeroed by
/// one thread first (4.7 GB of per-line records and 3.7 GB of COO arrays on the 117 M-line file: seconds of serial memset).
template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U *ptr) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(ptr)) U; }
    template <typename U, typename... Args> void construct(U *ptr, Args &&...args) { ::new (static_cast<void *>(ptr)) U(std::forward<Args>(args)...); }
};
template <typename T> using NoInitVector = std::vector<T, DefaultInitAllocator<T>>;

template <typename ValueT>
struct CooMatrix {
    int num_rows = 0, num_cols = 0;
    NoInitVector<int> row, col;
    NoInitVector<ValueT> val;

    int num_nonzeros() const { return (int) row.size(); }
    void Clear() { NoInitVector<int>().swap(row); NoInitVector<int>().swap(col); NoInitVector<ValueT>().swap(val); }
    void Reserve(size_t n) { row.reserve(n); col.reserve(n); val.reserve(n); }
    void Push(int r, int c, ValueT v) { row.push_back(r); col.push_back(c); val.push_back(v); }

    /// Dense rows x cols, row-major, all = default_value (InitDense, sparse_matrix.h:386-413).
    void InitDense(int rows, int cols, ValueT default_value = 1.0)
    {
        num_rows = rows; num_cols = cols;
        const size_t n = (size_t) rows * cols;
        row.resize(n); col.resize(n); val.assign(n, default_value);
#pragma omp parallel for schedule(static)
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) { row[(size_t) r * cols + c] = r; col[(size_t) r * cols + c] = c; }
    }

    /// Hub-and-rim wheel with `spokes` spokes (InitWheel, sparse_matrix.h:419-452).
    void InitWheel(int spokes, ValueT default_value = 1.0)
    {
        num_rows = num_cols = spokes + 1;
        Reserve((size_t) spokes * 2);
        for (int i = 0; i < spokes; ++i) Push(0, i + 1, default_value);
        for (int i = 0; i < spokes; ++i) Push(i + 1, (i + 1) % spokes + 1, default_value);
    }

    /// width^2 lattice, neighbours emitted W, E, N, S (+ self) (InitGrid2d, sparse_matrix.h:461-526).
    void InitGrid2d(int width, bool self_loop, ValueT default_value = 1.0)
    {
        num_rows = num_cols = width * width;
        Reserve((size_t) num_rows * (self_loop ? 5 : 4));
        for (int j = 0; j < width; ++j)
            for (int k = 0; k < width; ++k) {
                const int me = j * width + k;
                if (k - 1 >= 0) Push(me, me - 1, default_value);
                if (k + 1 < width) Push(me, me + 1, default_value);
                if (j - 1 >= 0) Push(me, me - width, default_value);
                if (j + 1 < width) Push(me, me + width, default_value);
                if (self_loop) Push(me, me, default_value);
            }
    }

    /// width^3 lattice, neighbours k-1, k+1, j-1, j+1, i-1, i+1 (+ self) (InitGrid3d, sparse_matrix.h:533-617).
    void InitGrid3d(int width, bool self_loop, ValueT default_value = 1.0)
    {
        const int w = width, w2 = width * width;
        num_rows = num_cols = w2 * w;
        Reserve((size_t) num_rows * (self_loop ? 7 : 6));
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < w; ++j)
                for (int k = 0; k < w; ++k) {
                    const int me = i * w2 + j * w + k;
                    if (k - 1 >= 0) Push(me, me - 1, default_value);
                    if (k + 1 < w) Push(me, me + 1, default_value);
                    if (j - 1 >= 0) Push(me, me - w, default_value);
                    if (j + 1 < w) Push(me, me + w, default_value);
                    if (i - 1 >= 0) Push(me, me - w2, default_value);
                    if (i + 1 < w) Push(me, me + w2, default_value);
                    if (self_loop) Push(me, me, default_value);
                }
    }

    /// Matrix Market reader with the reference's quirks (InitMarket, sparse_matrix.h:217-380):
    /// banner flags are substring tests for "symmetric", "skew", "array" only (:265-267);
    /// a line of >= 1024 characters, or a last line without a newline, ends parsing
    /// (getline(line,1024) + good() test, :244-250); indices by strtol base 0 (:330-345);
    /// a missing value becomes default_value (:351-355); entries are 1-based (:357);
    /// symmetric input mirrors off-diagonal entries, negated when skew (:362-368);
    /// array format is column-major and NOT index-shifted (:316-324); the final
    /// nonzero count is the number of entries produced (:373).
    /// Throws MarketError where the reference prints to stderr and exit(1)s.
    ///
    /// The file is read into memory once and its entry lines are parsed by all OpenMP
    /// threads (the reference's single-threaded reader is minutes for an
    /// Orkut-class file); results, including which error is reported first, are identical
    /// to the line-by-line reader.  `serial` forces one thread (used by the tests).
    void InitMarket(const std::string &filename, ValueT default_value = 1.0, bool verbose = false, bool serial = false)
    {
        if (verbose) { printf("Reading... "); fflush(stdout); }
        const double t_begin = omp_get_wtime();
        std::string buf;
        {
            std::ifstream ifs(filename.c_str(), std::ifstream::in | std::ifstream::binary);
            if (!ifs.good()) throw MarketError("Error opening file");
            ifs.seekg(0, std::ios::end);
            const std::streamoff size = ifs.tellg();
            ifs.seekg(0, std::ios::beg);
            buf.resize(size > 0 ? (size_t) size : 0);
            if (size > 0) ifs.read(&buf[0], size);
        }
        if (verbose) { printf("Parsing... "); fflush(stdout); }
        const double t_read = omp_get_wtime();
        ingest_times().read_s = t_read - t_begin;
        // ---- lines: only newline-terminated ones count; a line of >= 1024 characters ends the file
        const size_t n = buf.size();
        NoInitVector<size_t> starts;                       // start offset of every terminated line
        {
            // newline positions, found by all threads (chunk c of the file -> local[c]), then laid out in file order
            const int chunks = serial ? 1 : std::max(1, omp_get_max_threads());
            std::vector<std::vector<size_t>> local((size_t) chunks);
#pragma omp parallel for schedule(static, 1) if (!serial)
            for (int c = 0; c < chunks; ++c) {
                const size_t lo = n * (size_t) c / chunks, hi = n * (size_t) (c + 1) / chunks;
                std::vector<size_t> &v = local[(size_t) c];
                for (size_t i = lo; i < hi; ++i) if (buf[i] == '\n') v.push_back(i);
            }
            std::vector<size_t> base((size_t) chunks + 1, 0);
            for (int c = 0; c < chunks; ++c) base[(size_t) c + 1] = base[(size_t) c] + local[(size_t) c].size();
            const size_t lines = base[(size_t) chunks];
            NoInitVector<size_t> newlines(lines);
#pragma omp parallel for schedule(static, 1) if (!serial)
            for (int c = 0; c < chunks; ++c) std::copy(local[(size_t) c].begin(), local[(size_t) c].end(), newlines.begin() + (std::ptrdiff_t) base[(size_t) c]);
            // getline(line, 1024) fails on the first line of >= 1024 characters: parsing stops there
            size_t kept = lines;
#pragma omp parallel for schedule(static) reduction(min : kept) if (!serial)
            for (size_t k = 0; k < lines; ++k) {
                const size_t begin = k == 0 ? 0 : newlines[k - 1] + 1;
                if (newlines[k] - begin >= 1024 && k < kept) kept = k;
            }
            starts.resize(kept);
#pragma omp parallel for schedule(static) if (!serial)
            for (size_t k = 0; k < kept; ++k) {
                starts[k] = k == 0 ? 0 : newlines[k - 1] + 1;
                buf[newlines[k]] = '\0';                   // every kept line is now a C string
            }
        }
        // ---- header (serial): comments / banner until the size line
        bool array = false, symmetric = false, skew = false;
        long long declared = 0;
        size_t li = 0;
        bool have_size = false;
        for (; li < starts.size() && !have_size; ++li) {
            const char *line = &buf[starts[li]];
            if (line[0] == '%') {
                if (line[1] == '%') {
                    symmetric = strstr(line, "symmetric") != nullptr;
                    skew = strstr(line, "skew") != nullptr;
                    array = strstr(line, "array") != nullptr;
                    if (verbose) { printf("(symmetric: %d, skew: %d, array: %d) ", symmetric, skew, array); fflush(stdout); }
                }
                continue;
            }
            int nz = 0;
            const int parsed = sscanf(line, "%d %d %d", &num_rows, &num_cols, &nz);
            if (!array && parsed == 3) declared = symmetric ? 2LL * nz : nz;
            else if (array && parsed == 2) declared = (long long) num_rows * num_cols;
            else throw MarketError(std::string("Error parsing MARKET matrix: invalid problem description: ") + line);
            have_size = true;
        }
        // ---- entry lines, in parallel.  kind: 0 comment, 1 one entry, 2 entry + mirror, <0 error
        const size_t m = starts.size() - li;
        struct Parsed { int kind, r, c; double v; bool array, symmetric, skew; };
        NoInitVector<Parsed> parsed(m);
        auto parse_line = [&](const char *line, bool is_array, bool is_symmetric, bool is_skew) {
            Parsed q{0, 0, 0, 0.0, is_array, is_symmetric, is_skew};
            if (line[0] == '%') return q;
            if (is_array) { q.kind = sscanf(line, "%lf", &q.v) == 1 ? 1 : -3; return q; }
            // strtol(., ., 0) as the reference calls it (:330-345), with a shortcut for what every real file holds: blanks,
            // then a decimal number of at most 9 digits that does not start with 0 -- the same value and end pointer, a
            // tenth of the time (strtol is locale-aware and checks three bases); anything else goes to strtol itself
            auto index = [](char *l, char **t) -> int {
                char *s = l;
                while (*s == ' ' || *s == '\t') ++s;
                if (*s >= '1' && *s <= '9') {
                    int v = 0, digits = 0;          // (stops after 9 digits: 999 999 999 fits an int, a tenth digit would not)
                    while (*s >= '0' && *s <= '9' && digits < 9) { v = v * 10 + (*s - '0'); ++s; ++digits; }
                    if (!(*s >= '0' && *s <= '9')) { *t = s; return v; }
                }
                return (int) strtol(l, t, 0);
            };
            char *l = const_cast<char *>(line), *t = nullptr;
            q.r = index(l, &t);
            if (t == l) { q.kind = -1; return q; }
            l = t;
            q.c = index(l, &t);
            if (t == l) { q.kind = -2; return q; }
            l = t;
            // the value: pattern files have none -- nothing but blanks up to the end of the line is what strtod reports as
            // "no conversion" (t == l) -- so that case does not call it
            {
                const char *e = l;
                while (*e == ' ' || *e == '\t' || *e == '\r') ++e;
                if (*e == '\0') t = l; else q.v = strtod(l, &t);
            }
            if (t == l) q.v = (double) default_value;
            q.kind = (is_symmetric && q.r != q.c) ? 2 : 1;
            return q;
        };
        bool late_banner = false;
#pragma omp parallel for schedule(static) reduction(|| : late_banner) if (!serial)
        for (size_t k = 0; k < m; ++k) {
            const char *line = &buf[starts[li + k]];
            if (line[0] == '%' && line[1] == '%') late_banner = true;
            parsed[k] = parse_line(line, array, symmetric, skew);
        }
        if (late_banner) {
            // a "%%" line after the size line re-evaluates the three flags for the lines that
            // follow it (:262-268 runs for every such line); rare, so replay those in file order
            bool a = array, s = symmetric, w = skew;
            for (size_t k = 0; k < m; ++k) {
                const char *line = &buf[starts[li + k]];
                if (line[0] == '%' && line[1] == '%') {
                    s = strstr(line, "symmetric") != nullptr;
                    w = strstr(line, "skew") != nullptr;
                    a = strstr(line, "array") != nullptr;
                }
                parsed[k] = parse_line(line, a, s, w);
            }
        }
        // ---- sequential semantics: entry counter before each line, first failure in file order.  Coordinate files: the
        //      entries a line yields are known from the parse alone, so the counter is a prefix sum computed chunk-wise by
        //      all threads, and every chunk reports its first failure; array files (position from the counter, symmetric
        //      mirroring decided by the position) go through the loop as one chunk.
        NoInitVector<long long> before(m + 1);
        before[m] = 0;
        {
            bool any_array = false;
#pragma omp parallel for schedule(static) reduction(|| : any_array) if (!serial)
            for (size_t k = 0; k < m; ++k) any_array = any_array || (parsed[k].array && parsed[k].kind == 1);
            const int chunks = (serial || any_array) ? 1 : std::max(1, omp_get_max_threads());
            std::vector<long long> chunk_sum((size_t) chunks + 1, 0);
            if (chunks > 1) {
#pragma omp parallel for schedule(static, 1)
                for (int c = 0; c < chunks; ++c) {
                    long long sum = 0;
                    for (size_t k = m * (size_t) c / chunks; k < m * (size_t) (c + 1) / chunks; ++k) sum += parsed[k].kind > 0 ? parsed[k].kind : 0;
                    chunk_sum[(size_t) c + 1] = sum;
                }
                for (int c = 0; c < chunks; ++c) chunk_sum[(size_t) c + 1] += chunk_sum[(size_t) c];
            }
            std::vector<size_t> first_bad((size_t) chunks, m);          // index of each chunk's first failing line (m: none)
#pragma omp parallel for schedule(static, 1) if (chunks > 1)
            for (int c = 0; c < chunks; ++c) {
                long long cur = chunk_sum[(size_t) c];
                const size_t lo = m * (size_t) c / chunks, hi = m * (size_t) (c + 1) / chunks;
                for (size_t k = lo; k < hi; ++k) {
                    Parsed &q = parsed[k];
                    before[k] = cur;
                    if (q.array && q.kind == 1) {              // array: position comes from the entry counter (:316-324)
                        q.c = num_rows ? (int) (cur / num_rows) : 0;
                        q.r = (int) (cur - (long long) num_rows * q.c);
                        if (q.symmetric && q.r != q.c) q.kind = 2;
                    }
                    if ((q.kind != 0 && cur >= declared) || q.kind < 0) { first_bad[(size_t) c] = k; break; }
                    cur += q.kind > 0 ? q.kind : 0;
                }
                if (first_bad[(size_t) c] == m && hi == m) before[m] = cur;
            }
            size_t bad = m;
            for (int c = 0; c < chunks; ++c) if (first_bad[(size_t) c] < bad) { bad = first_bad[(size_t) c]; break; }
            if (bad < m) {
                const Parsed &q = parsed[bad];
                if (q.kind != 0 && before[bad] >= declared)
                    throw MarketError("Error parsing MARKET matrix: encountered more than " + std::to_string(declared) + " num_nonzeros");
                if (q.kind == -1) throw MarketError("Error parsing MARKET matrix: badly formed row at edge " + std::to_string(before[bad]));
                if (q.kind == -2) throw MarketError("Error parsing MARKET matrix: badly formed col at edge " + std::to_string(before[bad]));
                throw MarketError("Error parsing MARKET matrix: badly formed current_nz: '" + std::string(&buf[starts[li + bad]]) + "'");
            }
        }
        // ---- fill
        const size_t total = (size_t) before[m];
        row.resize(total); col.resize(total); val.resize(total);
#pragma omp parallel for schedule(static) if (!serial)
        for (size_t k = 0; k < m; ++k) {
            const Parsed &q = parsed[k];
            if (q.kind <= 0) continue;
            const size_t at = (size_t) before[k];
            const int shift = q.array ? 0 : 1;              // coordinate entries are 1-based (:357)
            row[at] = q.r - shift; col[at] = q.c - shift; val[at] = (ValueT) q.v;
            if (q.kind == 2) {
                row[at + 1] = q.c - shift; col[at + 1] = q.r - shift;
                val[at + 1] = (ValueT) q.v * (ValueT) (q.skew ? -1 : 1);
            }
        }
        ingest_times().parse_s = omp_get_wtime() - t_read;
        if (verbose) { printf("done. "); fflush(stdout); }
    }
};

/// CSR in the reference's layout (sparse_matrix.h:645-650): row_offsets[rows+1]
/// ([0] = 0, [rows] = nnz, repeated for empty rows), column_indices[nnz]
/// 0-based and sorted by (row, col) with duplicates kept, values[nnz].
template <typename ValueT>
struct CsrMatrix {
    int num_rows = 0, num_cols = 0, num_nonzeros = 0;
    std::vector<int> row_offsets;
    NoInitVector<int> column_indices;
    NoInitVector<ValueT> values;

    CsrMatrix() = default;
    explicit CsrMatrix(const CooMatrix<ValueT> &coo) { Init(coo); }

    /// Binary image of the CSR arrays ("optional binary CSR cache"): parsing and
    /// sorting an Orkut-class .mtx takes seconds even in parallel, loading this takes a read().
    /// Layout: "MSPMVCSR", u32 version = 1, u32 sizeof(ValueT), i32 rows, cols, nnz, then
    /// row_offsets[rows+1], column_indices[nnz], values[nnz].  No reference counterpart.
    bool SaveBinary(const std::string &path) const
    {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) return false;
        const char magic[8] = {'M', 'S', 'P', 'M', 'V', 'C', 'S', 'R'};
        const uint32_t head[2] = {1u, (uint32_t) sizeof(ValueT)};
        const int32_t dims[3] = {num_rows, num_cols, num_nonzeros};
        bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(head, 4, 2, f) == 2 && fwrite(dims, 4, 3, f) == 3;
        ok = ok && fwrite(row_offsets.data(), sizeof(int), row_offsets.size(), f) == row_offsets.size();
        ok = ok && fwrite(column_indices.data(), sizeof(int), column_indices.size(), f) == column_indices.size();
        ok = ok && fwrite(values.data(), sizeof(ValueT), values.size(), f) == values.size();
        ok = (fclose(f) == 0) && ok;
        if (!ok) remove(path.c_str());
        return ok;
    }
    /// false (matrix left empty) when the file is missing, truncated, of another precision or inconsistent
    bool LoadBinary(const std::string &path)
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        char magic[8]; uint32_t head[2]; int32_t dims[3];
        bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "MSPMVCSR", 8) == 0 && fread(head, 4, 2, f) == 2 &&
                  head[0] == 1u && head[1] == sizeof(ValueT) && fread(dims, 4, 3, f) == 3 && dims[0] >= 0 && dims[1] >= 0 &&
                  dims[2] >= 0;
        if (ok) {
            num_rows = dims[0]; num_cols = dims[1]; num_nonzeros = dims[2];
            row_offsets.resize((size_t) num_rows + 1); column_indices.resize((size_t) num_nonzeros); values.resize((size_t) num_nonzeros);
            // the three arrays follow the 28-byte header back to back: every thread preads its share of each (and first
            // touches the pages it fills); the file must end exactly there
            const int fd = fileno(f);
            const uint64_t at_off = 28, at_col = at_off + sizeof(int) * (uint64_t) row_offsets.size(),
                           at_val = at_col + sizeof(int) * (uint64_t) column_indices.size(),
                           at_end = at_val + sizeof(ValueT) * (uint64_t) values.size();
            struct stat st;
            ok = fstat(fd, &st) == 0 && (uint64_t) st.st_size == at_end;
            auto read_all = [&](char *dst, uint64_t bytes, uint64_t file_at) {
                bool good = true;
                const int T = std::max(1, omp_get_max_threads());
#pragma omp parallel for schedule(static, 1) reduction(&& : good)
                for (int t = 0; t < T; ++t) {
                    uint64_t lo = bytes * (uint64_t) t / T, hi = bytes * (uint64_t) (t + 1) / T;
                    while (lo < hi && good) {
                        const ssize_t got = pread(fd, dst + lo, (size_t) std::min<uint64_t>(hi - lo, 1u << 30), (off_t) (file_at + lo));
                        if (got <= 0) good = false; else lo += (uint64_t) got;
                    }
                }
                return good;
            };
            ok = ok && read_all(reinterpret_cast<char *>(row_offsets.data()), at_col - at_off, at_off) &&
                 read_all(reinterpret_cast<char *>(column_indices.data()), at_val - at_col, at_col) &&
                 read_all(reinterpret_cast<char *>(values.data()), at_end - at_val, at_val);
            ok = ok && row_offsets.front() == 0 && row_offsets.back() == num_nonzeros;
            if (ok) {
                bool good = true;
#pragma omp parallel for schedule(static) reduction(&& : good)
                for (size_t r = 0; r < (size_t) num_rows; ++r) good = good && row_offsets[r] <= row_offsets[r + 1];
#pragma omp parallel for schedule(static) reduction(&& : good)
                for (size_t k = 0; k < (size_t) num_nonzeros; ++k) good = good && (unsigned) column_indices[k] < (unsigned) num_cols;
                ok = good;
            }
        }
        fclose(f);
        if (!ok) { num_rows = num_cols = num_nonzeros = 0; row_offsets.clear(); column_indices.clear(); values.clear(); }
        return ok;
    }

    /// COO -> CSR (CsrMatrix::Init, sparse_matrix.h:666-728): same result as a stable
    /// sort of the tuples by (row, col).
    void Init(const CooMatrix<ValueT> &coo)
    {
        num_rows = coo.num_rows; num_cols = coo.num_cols; num_nonzeros = coo.num_nonzeros();
        const size_t n = (size_t) num_nonzeros;
        row_offsets.assign((size_t) num_rows + 1, 0);
        const int T = std::max(1, omp_get_max_threads());
        // ---- row histogram, all threads (relaxed atomic increments: collisions are rare), indices validated on the way
        //      (the reference never checks: a row or column outside the matrix becomes an out-of-bounds access)
        int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
        for (size_t k = 0; k < n; ++k) {
            const int r = coo.row[k];
            if (r < 0 || r >= num_rows) { bad |= 1; continue; }
            if ((unsigned) coo.col[k] >= (unsigned) num_cols) bad |= 2;
#pragma omp atomic
            ++row_offsets[(size_t) r + 1];
        }
        if (bad & 1) throw MarketError("row index out of range");
        if (bad & 2) throw MarketError("col index out of range");
        for (int r = 0; r < num_rows; ++r) row_offsets[(size_t) r + 1] += row_offsets[r];
        // ---- stable scatter by row (keeps emission order inside a row), two levels so that every thread works:
        //      the rows are cut into T blocks of about equal nonzero count; entry chunk c counts its entries per
        //      block, a prefix over (block, chunk) gives every chunk its stable slot range inside every block, the
        //      chunks scatter their entry ids there, and finally each block runs the classic cursor scatter on its own.
        NoInitVector<int> perm(n);
        {
            std::vector<int> block_first_row((size_t) T + 1, num_rows);
            block_first_row[0] = 0;
            for (int b = 1; b < T; ++b) {
                const int target = (int) ((long long) n * b / T);
                block_first_row[(size_t) b] = (int) (std::lower_bound(row_offsets.begin(), row_offsets.end() - 1, target) - row_offsets.begin());
                if (block_first_row[(size_t) b] < block_first_row[(size_t) b - 1]) block_first_row[(size_t) b] = block_first_row[(size_t) b - 1];
            }
            auto block_of = [&](int r) { return (int) (std::upper_bound(block_first_row.begin() + 1, block_first_row.begin() + T, r) - (block_first_row.begin() + 1)); };
            std::vector<size_t> cnt((size_t) T * T, 0);                 // [chunk][block]
            NoInitVector<int> tmp(n);
#pragma omp parallel for schedule(static, 1)
            for (int c = 0; c < T; ++c) {
                const size_t lo = n * (size_t) c / T, hi = n * (size_t) (c + 1) / T;
                size_t *mine = &cnt[(size_t) c * T];
                for (size_t k = lo; k < hi; ++k) ++mine[block_of(coo.row[k])];
            }
            for (int b = 0; b < T; ++b) {
                size_t at = (size_t) row_offsets[(size_t) block_first_row[(size_t) b]];
                for (int cc = 0; cc < T; ++cc) { const size_t v = cnt[(size_t) cc * T + b]; cnt[(size_t) cc * T + b] = at; at += v; }
            }
#pragma omp parallel for schedule(static, 1)
            for (int c = 0; c < T; ++c) {
                const size_t lo = n * (size_t) c / T, hi = n * (size_t) (c + 1) / T;
                size_t *mine = &cnt[(size_t) c * T];
                for (size_t k = lo; k < hi; ++k) tmp[mine[block_of(coo.row[k])]++] = (int) k;
            }
            // block b: cursor scatter of its (already block-local, still emission-ordered) entry ids
#pragma omp parallel for schedule(static, 1)
            for (int blk = 0; blk < T; ++blk) {
                const int r0 = block_first_row[(size_t) blk], r1 = block_first_row[(size_t) blk + 1];
                if (r1 <= r0) continue;
                std::vector<int> cursor(row_offsets.begin() + r0, row_offsets.begin() + r1);
                const size_t a = (size_t) row_offsets[(size_t) r0], e = (size_t) row_offsets[(size_t) r1];
                for (size_t i = a; i < e; ++i) { const int k = tmp[i]; perm[(size_t) cursor[(size_t) (coo.row[(size_t) k] - r0)]++] = k; }
            }
        }
        // stable sort by column inside each row, rows in parallel.  The row's columns are fetched ONCE into (column, entry
        // id) keys -- the ids of a row ascend in emission order, so sorting the 64-bit keys is the stable sort by column --
        // instead of being looked up through the permutation in every comparison (two cache misses each on a 234 M-entry file)
        column_indices.resize(n); values.resize(n);
#pragma omp parallel
        {
            std::vector<unsigned long long> keys;
#pragma omp for schedule(dynamic, 1024)
            for (int r = 0; r < num_rows; ++r) {
                const size_t b = (size_t) row_offsets[r], e = (size_t) row_offsets[(size_t) r + 1];
                bool sorted = true;
                int prev = -1;
                for (size_t i = b; i < e; ++i) {
                    const int c = coo.col[(size_t) perm[i]];
                    column_indices[i] = c;
                    if (c < prev) sorted = false;
                    prev = c;
                }
                if (!sorted) {
                    keys.resize(e - b);
                    for (size_t i = b; i < e; ++i) keys[i - b] = ((unsigned long long) (unsigned) column_indices[i] << 32) | (unsigned) perm[i];
                    std::sort(keys.begin(), keys.end());
                    for (size_t i = b; i < e; ++i) { perm[i] = (int) (unsigned) keys[i - b]; column_indices[i] = (int) (keys[i - b] >> 32); }
                }
                for (size_t i = b; i < e; ++i) values[i] = coo.val[(size_t) perm[i]];
            }
        }
    }

    /// Row-length statistics (CsrMatrix::Stats, sparse_matrix.h:897-910; the
    /// pearson_r it also computes is never printed and is not reproduced).
    GraphStats Stats() const
    {
        GraphStats s;
        s.num_rows = num_rows; s.num_cols = num_cols; s.num_nonzeros = num_nonzeros;
        s.row_length_mean = double(num_nonzeros) / num_rows;
        double variance = 0.0, cube = 0.0;
        for (int r = 0; r < num_rows; ++r) {
            const double delta = double(row_offsets[(size_t) r + 1] - row_offsets[r]) - s.row_length_mean;
            variance += delta * delta;
            cube += delta * delta * delta;
        }
        variance /= num_rows;
        s.row_length_std_dev = std::sqrt(variance);
        s.row_length_skewness = (cube / num_rows) / std::pow(s.row_length_std_dev, 3.0);
        s.row_length_variation = s.row_length_std_dev / s.row_length_mean;
        return s;
    }

    /// Log-10 histogram of row lengths (DisplayHistogram, sparse_matrix.h:919-956),
    /// including its percentage relative to num_COLS (:953).
    void DisplayHistogram(FILE *out = stdout) const
    {
        int log_counts[12] = {0};
        int max_log = -1, max_len = -1;
        for (int r = 0; r < num_rows; ++r) {
            int length = row_offsets[(size_t) r + 1] - row_offsets[r];
            max_len = std::max(max_len, length);
            int lg = -1;
            while (length > 0) { length /= 10; ++lg; }
            max_log = std::max(max_log, lg);
            ++log_counts[lg + 1];
        }
        fprintf(out, "CSR matrix (%d rows, %d columns, %d non-zeros, max-length %d):\n", num_rows, num_cols,
                num_nonzeros, max_len);
        for (int i = -1; i < max_log + 1; ++i)
            fprintf(out, "\tDegree 1e%d: \t%d (%.2f%%)\n", i, log_counts[i + 1],
                    (float) log_counts[i + 1] * 100.0 / num_cols);
        fflush(out);
    }

    /// --v2 dump (CsrMatrix::Display, sparse_matrix.h:962-976).
    void Display(FILE *out = stdout) const
    {
        fprintf(out, "Input Matrix (%d vertices, %d nonzeros):\n", num_rows, num_nonzeros);
        for (int r = 0; r < num_rows; ++r) {
            fprintf(out, "%d [@%d, #%d]: ", r, row_offsets[r], row_offsets[(size_t) r + 1] - row_offsets[r]);
            for (int k = row_offsets[r]; k < row_offsets[(size_t) r + 1]; ++k)
                fprintf(out, "%d (%f), ", column_indices[k], (double) values[k]);
            fprintf(out, "\n");
        }
        fflush(out);
    }
};

/// Sequential gold (SpmvGold, gpu_spmv.cu:72-92 / cpu_spmv.cpp:257-277): accumulates in ValueT.
template <typename ValueT>
void SpmvGold(const CsrMatrix<ValueT> &a, const ValueT *x, const ValueT *y_in, ValueT *y_out, ValueT alpha, ValueT beta)
{
    for (int r = 0; r < a.num_rows; ++r) {
        ValueT partial = beta * y_in[r];
        for (int k = a.row_offsets[r]; k < a.row_offsets[(size_t) r + 1]; ++k)
            partial += alpha * a.values[k] * x[a.column_indices[k]];
        y_out[r] = partial;
    }
}

}  // namespace mspmv_host
