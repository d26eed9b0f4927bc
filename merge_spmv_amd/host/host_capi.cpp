// host_capi.cpp -- a small C face over the host data model (sparse_matrix.hpp,
// utils.hpp) so the parity tests can drive the PRODUCT's C++ host code through
// ctypes and compare it with golden vectors from the reference's sparse_matrix.h
// and utils.h (tests/test_host_model.py).  Built to libmspmv_host.so; no GPU code.
#include <cstring>
#include <sstream>
#include <string>

#include "driver_common.hpp"

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
}
