// TEST INFRASTRUCTURE: builds the REFERENCE's own host data model
// (sparse_matrix.h: CooMatrix generators / InitMarket / CsrMatrix::Init /
// Stats / DisplayHistogram) and utils.h (CommandLineArgs, CompareResults)
// straight from the read-only reference tree (no CUB_MKL, no __NVCC__: both
// headers are self-contained C++ in that configuration).  Used by
// oracle/make_golden.py to generate tests/golden/*.json and to pin the numpy
// restatements in oracle/oracle.py.  Output binary goes to oracle/_ref/ only.
//
// usage:
//   ref_host csr <f32|f64> dense <rows> <cols> | grid2d <w> | grid3d <w> |
//                          wheel <spokes> | mtx <file>
//        -> JSON {rows, cols, nnz, row_offsets, column_indices, values, stats{...}}
//   ref_host hist <f32|f64> <kind...>      -> the DisplayHistogram text
//   ref_host cmp <f32|f64> <file>          -> "0"/"1": CompareResults verdict
//        file: len, then len computed values, then len reference values
//   ref_host args <argv...>                -> parsed view of CommandLineArgs
#include <cmath>
#include <cstring>
#include <string>
#include "sparse_matrix.h"
#include "utils.h"

template <typename V>
static int build(CooMatrix<V, int>& coo, int argc, char** argv, int at)
{
    std::string kind = argv[at];
    if (kind == "dense")       coo.InitDense(atoi(argv[at + 1]), atoi(argv[at + 2]));
    else if (kind == "grid2d") coo.InitGrid2d(atoi(argv[at + 1]), false);
    else if (kind == "grid3d") coo.InitGrid3d(atoi(argv[at + 1]), false);
    else if (kind == "wheel")  coo.InitWheel(atoi(argv[at + 1]));
    else if (kind == "mtx")    coo.InitMarket(argv[at + 1], 1.0, false);
    else return 1;
    return 0;
}

static void pnum(double v)
{
    if (std::isnan(v)) printf("\"nan\""); else if (std::isinf(v)) printf(v > 0 ? "\"inf\"" : "\"-inf\"");
    else printf("%.17g", v);
}

template <typename V>
static int do_csr(int argc, char** argv, bool hist)
{
    CooMatrix<V, int> coo;
    if (build(coo, argc, argv, 3)) return 1;
    CsrMatrix<V, int> csr(coo);
    if (hist) { csr.DisplayHistogram(); return 0; }
    GraphStats st = csr.Stats();
    printf("{\"rows\": %d, \"cols\": %d, \"nnz\": %d,\n \"row_offsets\": [", csr.num_rows, csr.num_cols, csr.num_nonzeros);
    for (int i = 0; i <= csr.num_rows; ++i) printf("%s%d", i ? "," : "", csr.row_offsets[i]);
    printf("],\n \"column_indices\": [");
    for (int i = 0; i < csr.num_nonzeros; ++i) printf("%s%d", i ? "," : "", csr.column_indices[i]);
    printf("],\n \"values\": [");
    for (int i = 0; i < csr.num_nonzeros; ++i) { if (i) printf(","); pnum((double) csr.values[i]); }
    printf("],\n \"stats\": {\"row_length_mean\": "); pnum(st.row_length_mean);
    printf(", \"row_length_std_dev\": "); pnum(st.row_length_std_dev);
    printf(", \"row_length_variation\": "); pnum(st.row_length_variation);
    printf(", \"row_length_skewness\": "); pnum(st.row_length_skewness);
    printf(", \"pearson_r\": "); pnum(st.pearson_r);
    printf("},\n \"stats_csv\": \"");
    fflush(stdout); st.Display(false); fflush(stdout);
    printf("\"}\n");
    return 0;
}

template <typename V>
static int do_cmp(const char* file)
{
    FILE* f = fopen(file, "r"); if (!f) return 2;
    int len; if (fscanf(f, "%d", &len) != 1) return 2;
    std::vector<V> a(len), b(len);
    for (int i = 0; i < len; ++i) { double v; if (fscanf(f, "%lf", &v) != 1) return 2; a[i] = (V) v; }
    for (int i = 0; i < len; ++i) { double v; if (fscanf(f, "%lf", &v) != 1) return 2; b[i] = (V) v; }
    printf("%d\n", CompareResults(a.data(), b.data(), len, false));
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    std::string cmd = argv[1];
    if (cmd == "csr" || cmd == "hist") {
        if (argc < 5) return 1;
        bool f32 = !strcmp(argv[2], "f32");
        return f32 ? do_csr<float>(argc, argv, cmd == "hist") : do_csr<double>(argc, argv, cmd == "hist");
    }
    if (cmd == "cmp") {
        if (argc < 4) return 1;
        return !strcmp(argv[2], "f32") ? do_cmp<float>(argv[3]) : do_cmp<double>(argv[3]);
    }
    if (cmd == "args") {
        CommandLineArgs args(argc - 1, argv + 1);
        int i = -1, grid2d = -1; float alpha = 1.0f; std::string mtx;
        args.GetCmdLineArgument("i", i);
        args.GetCmdLineArgument("grid2d", grid2d);
        args.GetCmdLineArgument("alpha", alpha);
        args.GetCmdLineArgument("mtx", mtx);
        printf("{\"quiet\": %d, \"fp32\": %d, \"i\": %d, \"grid2d\": %d, \"alpha\": %.9g, \"mtx\": \"%s\", \"naked\": %d}\n",
               (int) args.CheckCmdLineFlag("quiet"), (int) args.CheckCmdLineFlag("fp32"), i, grid2d, alpha, mtx.c_str(),
               (int) args.args.size());
        return 0;
    }
    return 1;
}
