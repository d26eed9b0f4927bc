// host_capi.cpp -- a small C face over the host data model (sparse_matrix.hpp,
// utils.hpp) so the parity tests can drive the PRODUCT's C++ host code through
// ctypes and compare it with golden vectors from the reference's sparse_matrix.h
// and utils.h (tests/test_host_model.py).  Built to libmspmv_host.so; no GPU code.
#include <cstring>
#include <sstream>
#include <string>

#include "driver_common.hpp"
#include "merge_csrmv.hpp"
#include "cpu_bench.hpp"

using namespace mspmv_host;

namespace {
struct Handle {
    CsrMatrix<double> csr64;
    CsrMatrix<float> csr32;
    bool f32 = false;
    std::string text;
};
template <typename V>
int build(Handle *h, CsrMatrix<V> &csr, const char *kind, int a, int b, const char *path)
{
    CooMatrix<V> coo;
    const std::string k = kind;
    try {
        if (k == "dense") coo.InitDense(a, b);
        else if (k == "grid2d") coo.InitGrid2d(a, false);
        else if (k == "grid3d") coo.InitGrid3d(a, false);
        else if (k == "wheel") coo.InitWheel(a);
        else if (k == "mtx") coo.InitMarket(path, (V) 1.0, false);
        else if (k == "mtx_serial") coo.InitMarket(path, (V) 1.0, false, true);
        else return 2;
        csr.Init(coo);
    } catch (const std::exception &e) { h->text = e.what(); return 1; }
    return 0;
}
std::string capture(FILE *tmp)
{
    std::string out; char buf[4096]; rewind(tmp);
    size_t n; while ((n = fread(buf, 1, sizeof(buf), tmp)) > 0) out.append(buf, n);
    fclose(tmp); return out;
}
}  // namespace

extern "C" {

void *mspmv_host_matrix_create(const char *kind, int a, int b, const char *path, int fp32, int *status)
{
    Handle *h = new Handle; h->f32 = fp32 != 0;
    const int st = fp32 ? build(h, h->csr32, kind, a, b, path) : build(h, h->csr64, kind, a, b, path);
    if (status) *status = st;
    return h;
}
void mspmv_host_matrix_destroy(void *p) { delete static_cast<Handle *>(p); }
const char *mspmv_host_matrix_error(void *p) { return static_cast<Handle *>(p)->text.c_str(); }
void mspmv_host_matrix_shape(void *p, int *rows, int *cols, int *nnz)
{
    Handle *h = static_cast<Handle *>(p);
    if (h->f32) { *rows = h->csr32.num_rows; *cols = h->csr32.num_cols; *nnz = h->csr32.num_nonzeros; }
    else { *rows = h->csr64.num_rows; *cols = h->csr64.num_cols; *nnz = h->csr64.num_nonzeros; }
}
void mspmv_host_matrix_copy(void *p, int *row_offsets, int *cols, void *values)
{
    Handle *h = static_cast<Handle *>(p);
    if (h->f32) {
        memcpy(row_offsets, h->csr32.row_offsets.data(), sizeof(int) * h->csr32.row_offsets.size());
        memcpy(cols, h->csr32.column_indices.data(), sizeof(int) * h->csr32.column_indices.size());
        memcpy(values, h->csr32.values.data(), sizeof(float) * h->csr32.values.size());
    } else {
        memcpy(row_offsets, h->csr64.row_offsets.data(), sizeof(int) * h->csr64.row_offsets.size());
        memcpy(cols, h->csr64.column_indices.data(), sizeof(int) * h->csr64.column_indices.size());
        memcpy(values, h->csr64.values.data(), sizeof(double) * h->csr64.values.size());
    }
}
// which: 0 = stats CSV line, 1 = labelled stats, 2 = histogram
const char *mspmv_host_matrix_text(void *p, int which)
{
    Handle *h = static_cast<Handle *>(p);
    FILE *tmp = tmpfile();
    if (which == 2) { if (h->f32) h->csr32.DisplayHistogram(tmp); else h->csr64.DisplayHistogram(tmp); }
    else { const GraphStats s = h->f32 ? h->csr32.Stats() : h->csr64.Stats(); s.Display(which == 1, tmp); }
    h->text = capture(tmp);
    return h->text.c_str();
}
int mspmv_host_compare_reference_rule_f32(const float *a, const float *b, int len) { return CompareResultsReferenceRule(a, b, len, false); }
int mspmv_host_compare_reference_rule_f64(const double *a, const double *b, int len) { return CompareResultsReferenceRule(a, b, len, false); }
int mspmv_host_adaptive_iterations(long long nnz, unsigned long long cap) { return AdaptiveIterations(nnz, cap); }
// parse argv like the drivers do; writes a JSON-ish summary into out
void mspmv_host_parse_args(int argc, char **argv, char *out, int out_len)
{
    CommandLineArgs args(argc, argv);
    int i = -1, grid2d = -1; float alpha = 1.0f; std::string mtx;
    args.GetCmdLineArgument("i", i); args.GetCmdLineArgument("grid2d", grid2d);
    args.GetCmdLineArgument("alpha", alpha); args.GetCmdLineArgument("mtx", mtx);
    snprintf(out, out_len, "{\"quiet\": %d, \"fp32\": %d, \"i\": %d, \"grid2d\": %d, \"alpha\": %.9g, \"mtx\": \"%s\", \"naked\": %d}",
             (int) args.CheckCmdLineFlag("quiet"), (int) args.CheckCmdLineFlag("fp32"), i, grid2d, alpha, mtx.c_str(),
             (int) args.NumNakedArgs());
}

// ---- the product's OpenMP merge-path CsrMV (merge_csrmv.hpp), exported so that
//      tests/test_cpu_product_parity.py can pin it against the oracle and bench.py can time it ----
void mspmv_host_merge_csrmv_f32(int segments, int threads, int rows, int nnz, const int *row_offsets, const int *cols,
                                const float *vals, const float *x, float *y)
{
    std::vector<int> cr; std::vector<float> cv;
    MergeCsrmv<float>(segments, rows, nnz, row_offsets + 1, cols, vals, x, y, cr, cv, threads);
}
void mspmv_host_merge_csrmv_f64(int segments, int threads, int rows, int nnz, const int *row_offsets, const int *cols,
                                const double *vals, const double *x, double *y)
{
    std::vector<int> cr; std::vector<double> cv;
    MergeCsrmv<double>(segments, rows, nnz, row_offsets + 1, cols, vals, x, y, cr, cv, threads);
}
}  // extern "C" (reopened below)

extern "C" {
int mspmv_host_merge_csrmv_bench_f32(int threads, int pin, int rows, int cols, int nnz, const int *row_offsets, const int *col_idx,
                                     const float *vals, const float *x, double budget_s, int max_iters, double *avg_ms,
                                     int *iters_done, int *pinned, int *packages, float *y_out)
{
    return BenchMerge<float>(threads, pin, rows, cols, nnz, row_offsets, col_idx, vals, x, budget_s, max_iters, avg_ms, iters_done,
                             pinned, packages, y_out);
}
int mspmv_host_merge_csrmv_bench_f64(int threads, int pin, int rows, int cols, int nnz, const int *row_offsets, const int *col_idx,
                                     const double *vals, const double *x, double budget_s, int max_iters, double *avg_ms,
                                     int *iters_done, int *pinned, int *packages, double *y_out)
{
    return BenchMerge<double>(threads, pin, rows, cols, nnz, row_offsets, col_idx, vals, x, budget_s, max_iters, avg_ms, iters_done,
                              pinned, packages, y_out);
}
// Write a Matrix Market `coordinate` file (general or symmetric banner) from 0-based index arrays: `pattern` when `values` is
// null, `real` otherwise; `comment` (optional, one line without the leading %) goes right after the banner.  The corpus-scale
// ingest test (tools/c3_ingest.py) and the stand-in corpus (tools/make_standin_mtx.py) write their 10^6..10^8-line inputs with this.
int mspmv_host_write_mtx(const char *path, int rows, int cols, long long n, const int *row, const int *col, const double *values,
                         int symmetric, const char *comment)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    fprintf(f, "%%%%MatrixMarket matrix coordinate %s %s\n", values ? "real" : "pattern", symmetric ? "symmetric" : "general");
    if (comment && comment[0]) fprintf(f, "%%%s\n", comment);
    fprintf(f, "%d %d %lld\n", rows, cols, n);
    const int T = omp_get_max_threads();
    const long long chunk = 1 << 22;                       // entries formatted per round and thread
    std::vector<std::string> out((size_t) T);
    for (long long base = 0; base < n; base += chunk * T) {
#pragma omp parallel num_threads(T)
        {
            const int t = omp_get_thread_num();
            std::string &b = out[(size_t) t]; b.clear();
            const long long lo = std::min(n, base + chunk * t), hi = std::min(n, lo + chunk);
            char tmp[64];
            for (long long k = lo; k < hi; ++k) {
                int len = values ? snprintf(tmp, sizeof(tmp), "%d %d %.17g\n", row[k] + 1, col[k] + 1, values[k])
                                 : snprintf(tmp, sizeof(tmp), "%d %d\n", row[k] + 1, col[k] + 1);
                b.append(tmp, (size_t) len);
            }
        }
        for (int t = 0; t < T; ++t) if (!out[(size_t) t].empty() && fwrite(out[(size_t) t].data(), 1, out[(size_t) t].size(), f) != out[(size_t) t].size()) { fclose(f); return 2; }
    }
    return fclose(f) == 0 ? 0 : 2;
}
int mspmv_host_write_pattern_mtx(const char *path, int rows, int cols, long long n, const int *row, const int *col, int symmetric)
{
    return mspmv_host_write_mtx(path, rows, cols, n, row, col, nullptr, symmetric, nullptr);
}

// the thread count the drivers default to: hardware threads capped by the cgroup CPU quota (utils.hpp)
int mspmv_host_usable_cpus(void) { return UsableCpus(); }
int mspmv_host_hardware_threads(void) { return omp_get_num_procs(); }
}
