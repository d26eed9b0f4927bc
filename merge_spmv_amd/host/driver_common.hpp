// driver_common.hpp -- what the cpu_spmv and gpu_spmv drivers share: the
// reference's command-line surface, input selection
// and labels (gpu_spmv.cu:598-664, cpu_spmv.cpp:537-592), the matrix report
// (gpu_spmv.cu:503-516) and the perf lines (gpu_spmv.cu:445-474,
// cpu_spmv.cpp:498-523).  Output formats are the reference's, byte for byte,
// so eval_csrmv.sh-style tooling keeps working.
#pragma once

#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "sparse_matrix.hpp"
#include "utils.hpp"

namespace mspmv_host {

struct RunConfig {
    bool quiet = false, verbose = false, verbose2 = false, fp32 = false, strict = true;
    std::string mtx;
    int grid2d = -1, grid3d = -1, wheel = -1, dense = -1;
    int size = 1 << 24;                    // --size (gpu_spmv.cu:645-646); the CPU driver fixes 1<<24 (cpu_spmv.cpp:584)
    int timing_iterations = -1;
    float alpha = 1.0f, beta = 0.0f;
    int threads = -1;                      // --threads (cpu_spmv.cpp:732)
    int device = 0;                        // --device  (utils.h:465-472)
    double peak_gbs = -1;                  // --peak-gbs: overrides the bus-width formula of utils.h:491
    bool cache = false;                    // --cache: keep / reuse <mtx>.<fp32|fp64>.csrbin next to a Matrix Market file
    bool timing = false;                   // --timing: print the wall-clock seconds of the ingest phases (non-quiet)
    bool check = false;                    // --check: with --quiet, the strict check of the merge-based result all the same: one line on STDERR (the CSV stays the reference's), exit code 2 on a violation
};

inline RunConfig ParseCommon(const CommandLineArgs &args, bool gpu_driver)
{
    RunConfig c;
    c.verbose = args.CheckCmdLineFlag("v");
    c.verbose2 = args.CheckCmdLineFlag("v2");
    c.quiet = args.CheckCmdLineFlag("quiet");
    c.fp32 = args.CheckCmdLineFlag("fp32");
    c.strict = !args.CheckCmdLineFlag("no-strict");
    args.GetCmdLineArgument("i", c.timing_iterations);
    args.GetCmdLineArgument("mtx", c.mtx);
    c.cache = args.CheckCmdLineFlag("cache");
    c.timing = args.CheckCmdLineFlag("timing");
    c.check = args.CheckCmdLineFlag("check");
    args.GetCmdLineArgument("grid2d", c.grid2d);
    args.GetCmdLineArgument("grid3d", c.grid3d);
    args.GetCmdLineArgument("wheel", c.wheel);      // parsed by gpu_spmv.cu:719 only; cpu_spmv.cpp forgot it
    args.GetCmdLineArgument("dense", c.dense);
    args.GetCmdLineArgument("alpha", c.alpha);
    args.GetCmdLineArgument("beta", c.beta);
    args.GetCmdLineArgument("peak-gbs", c.peak_gbs);
    if (gpu_driver) {
        args.GetCmdLineArgument("size", c.size);
        args.GetCmdLineArgument("device", c.device);
    } else {
        args.GetCmdLineArgument("threads", c.threads);
    }
    return c;
}

/// Build the input in COO form and print its label exactly as the reference
/// does (including `wheel_<grid2d>`, gpu_spmv.cu:639).  exit(0) on a trivial
/// Matrix Market dataset, exit(1) when no input was named.
template <typename ValueT>
void BuildInput(const RunConfig &c, CooMatrix<ValueT> &coo)
{
    if (!c.mtx.empty()) {
        try { coo.InitMarket(c.mtx, (ValueT) 1.0, !c.quiet); }
        catch (const MarketError &e) { fprintf(stderr, "%s\n", e.what()); exit(1); }
        if (coo.num_rows == 1 || coo.num_cols == 1 || coo.num_nonzeros() == 1) {
            if (!c.quiet) printf("Trivial dataset\n");
            exit(0);
        }
        printf("%s, ", c.mtx.c_str());
    } else if (c.grid2d > 0) {
        printf("grid2d_%d, ", c.grid2d);
        coo.InitGrid2d(c.grid2d, false);
    } else if (c.grid3d > 0) {
        printf("grid3d_%d, ", c.grid3d);
        coo.InitGrid3d(c.grid3d, false);
    } else if (c.wheel > 0) {
        printf("wheel_%d, ", c.grid2d);
        coo.InitWheel(c.wheel);
    } else if (c.dense > 0) {
        const int rows = c.size / c.dense;
        printf("dense_%d_x_%d, ", rows, c.dense);
        coo.InitDense(rows, c.dense);
    } else {
        fprintf(stderr, "No graph type specified.\n");
        exit(1);
    }
    fflush(stdout);
}

/// COO -> CSR for the chosen input; with --cache a Matrix Market input is served from / saved to a
/// binary CSR image beside it (newer than the .mtx), skipping the parse and the sort.
/// Returns the nonzero count (for the iteration count the GPU driver prints before converting).
template <typename ValueT>
void BuildCsr(const RunConfig &c, CsrMatrix<ValueT> &csr, void (*before_convert)(const RunConfig &, int) = nullptr)
{
    std::string bin;
    if (c.cache && !c.mtx.empty()) {
        bin = c.mtx + (sizeof(ValueT) == 4 ? ".fp32.csrbin" : ".fp64.csrbin");
        struct stat sm, sb;
        const double t0 = omp_get_wtime();
        if (stat(c.mtx.c_str(), &sm) == 0 && stat(bin.c_str(), &sb) == 0 && sb.st_mtime >= sm.st_mtime && csr.LoadBinary(bin)) {
            ingest_times().cache_load_s = omp_get_wtime() - t0;
            if (!c.quiet) { printf("Reading binary CSR image... done. "); fflush(stdout); }
            if (csr.num_rows == 1 || csr.num_cols == 1 || csr.num_nonzeros == 1) { if (!c.quiet) printf("Trivial dataset\n"); exit(0); }
            printf("%s, ", c.mtx.c_str()); fflush(stdout);
            if (before_convert) before_convert(c, csr.num_nonzeros);
            return;
        }
    }
    CooMatrix<ValueT> coo;
    BuildInput(c, coo);
    if (before_convert) before_convert(c, coo.num_nonzeros());
    double t0 = omp_get_wtime();
    csr.Init(coo);
    ingest_times().convert_s = omp_get_wtime() - t0;
    t0 = omp_get_wtime();
    if (!bin.empty() && !csr.SaveBinary(bin) && !c.quiet) fprintf(stderr, "(could not write %s)\n", bin.c_str());
    if (!bin.empty()) ingest_times().cache_save_s = omp_get_wtime() - t0;
}

/// Stats line / block, histogram and optional dump (gpu_spmv.cu:503-516).
template <typename ValueT>
void ReportMatrix(const RunConfig &c, const CsrMatrix<ValueT> &csr)
{
    csr.Stats().Display(!c.quiet);
    if (!c.quiet && c.timing) {
        const IngestTimes &t = ingest_times();
        printf("\n\t ingest seconds: read %.3f, parse %.3f, COO->CSR %.3f, cache save %.3f, cache load %.3f (%d threads)\n",
               t.read_s, t.parse_s, t.convert_s, t.cache_save_s, t.cache_load_s, omp_get_max_threads());
    }
    if (!c.quiet) {
        printf("\n");
        csr.DisplayHistogram();
        printf("\n");
        if (c.verbose2) csr.Display();
        printf("\n");
    }
    fflush(stdout);
}

/// "run 16 billion nonzeros through": clamp(2^34 / nnz, 100, cap)
/// (gpu_spmv.cu:492-493 cap 50000, cpu_spmv.cpp:611-616 cap 200000).
inline int AdaptiveIterations(long long nnz, unsigned long long cap)
{
    const unsigned long long want = nnz > 0 ? (16ull << 30) / (unsigned long long) nnz : cap;
    return (int) std::min(cap, std::max(100ull, want));
}

/// Perf line (gpu_spmv.cu:445-474 with peak, cpu_spmv.cpp:498-523 without).
inline void DisplayPerf(bool quiet, int value_bytes, double setup_ms, double avg_ms, long long rows, long long nnz,
                        double device_giga_bandwidth /* <= 0: CPU form */)
{
    const double total_bytes = double(nnz) * (value_bytes * 2 + 4) + double(rows) * (4 + value_bytes);
    const double nz_throughput = double(nnz) / avg_ms / 1.0e6;
    const double effective_bandwidth = total_bytes / avg_ms / 1.0e6;
    if (quiet)
        printf("%.5f, %.5f, %.6f, %.3lf, ", setup_ms, avg_ms, 2 * nz_throughput, effective_bandwidth);
    else if (device_giga_bandwidth > 0)
        printf("fp%d: %.4f setup ms, %.4f avg ms, %.5f gflops, %.3lf effective GB/s (%.2f%% peak)\n", value_bytes * 8,
               setup_ms, avg_ms, 2 * nz_throughput, effective_bandwidth,
               effective_bandwidth / device_giga_bandwidth * 100);
    else
        printf("fp%d: %.4f setup ms, %.4f avg ms, %.5f gflops, %.3lf effective GB/s\n", value_bytes * 8, setup_ms,
               avg_ms, 2 * nz_throughput, effective_bandwidth);
    fflush(stdout);
}

/// Extra line of this project (non-quiet only): compulsory-traffic bandwidth against the
/// roofline: B_alg = nnz*(sizeof V + 4) + (rows+1)*4 + rows*sizeof V + cols*sizeof V.
inline void DisplayRoofline(int value_bytes, double avg_ms, long long rows, long long cols, long long nnz,
                            double peak_gbs)
{
    const double b_alg = double(nnz) * (value_bytes + 4) + double(rows + 1) * 4 + double(rows) * value_bytes +
                         double(cols) * value_bytes;
    const double gbs = b_alg / avg_ms / 1.0e6;
    printf("\tcompulsory traffic %.1f MB -> %.1f GB/s", b_alg / 1e6, gbs);
    if (peak_gbs > 0) printf(" = %.2f%% of the %.0f GB/s HBM roofline", gbs / peak_gbs * 100, peak_gbs);
    printf("\n");
    fflush(stdout);
}

}  // namespace mspmv_host
