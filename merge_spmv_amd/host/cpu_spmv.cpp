// cpu_spmv.cpp -- OpenMP merge-path CsrMV driver: the reference's cpu_spmv
// command line and report (cpu_spmv.cpp:682-747) without MKL.  This is the
// PRODUCT CPU path (timed beside the GPU one); it shares no
// code with the oracle.
//
//   cpu_spmv [--quiet] [--v] [--v2] [--i=<iterations>] [--fp32] [--threads=<n>]
//            [--alpha=<a>] [--beta=<b>] [--pin] [--timing] [--cache]
//            --mtx=<file> | --dense=<cols> | --grid2d=<w> | --grid3d=<w> | --wheel=<spokes>
//
// Differences from the reference, all deliberate:
//  * the per-thread carry arrays are sized by the thread count (the reference
//    uses fixed [256] stack arrays, cpu_spmv.cpp:302-303, and smashes its stack
//    beyond 256 threads);
//  * the comparison column is a plain row-parallel OpenMP CsrMV ("OMP-row
//    CsrMV") where the reference calls MKL (cpu_spmv.cpp:417-491);
//  * besides the reference's vacuous PASS/FAIL rule, a strict per-row tolerance
//    check is printed (--no-strict to skip);
//  * --wheel is honoured (the reference never parses it, cpu_spmv.cpp:721-732);
//  * --pin adds a "Merge CsrMV (socket 0, first touch)" line: the same kernel timed on private copies of the arrays
//    first-touched by the threads that stream them, one thread per physical core of socket 0 (BASELINE config 1:
//    "single socket"; the reference relies on numactl / KMP_AFFINITY outside the program for that).
#include <omp.h>

#include <cstring>
#include <vector>

#include "driver_common.hpp"
#include "merge_csrmv.hpp"
#include "cpu_bench.hpp"

using namespace mspmv_host;

namespace {

template <typename V>
void RowParallelCsrmv(int threads, int rows, const int *row_offsets, const int *cols, const V *vals, const V *x, V *y)
{
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int r = 0; r < rows; ++r) {
        V sum = 0;
        for (int k = row_offsets[r]; k < row_offsets[r + 1]; ++k) sum += vals[k] * x[cols[k]];
        y[r] = sum;
    }
}

template <typename V, typename Fn>
float TimeMethod(const RunConfig &c, const CsrMatrix<V> &a, const V *x, const V *gold, V *y, int iterations, int ipt_for_bound,
                 Fn &&spmv)
{
    memset(y, -1, sizeof(V) * a.num_rows);            // NaN sentinel for unwritten rows (cpu_spmv.cpp:380)
    spmv();
    if (!c.quiet) {
        const int bad = CompareResultsReferenceRule(y, gold, a.num_rows, true);
        printf("\t%s\n", bad ? "FAIL" : "PASS");
        if (c.strict) {
            double worst = 0;
            const long long v = StrictCheck(a.num_rows, a.row_offsets.data(), a.column_indices.data(), a.values.data(), x, y,
                                            ipt_for_bound, &worst);
            printf("\tstrict check: %s (%lld rows outside tolerance, worst ratio %.3g)\n", v ? "FAIL" : "PASS", v, worst);
        }
        fflush(stdout);
    }
    spmv(); spmv(); spmv();                           // re-populate caches (cpu_spmv.cpp:390-392)
    CpuTimer timer;
    timer.Start();
    for (int it = 0; it < iterations; ++it) spmv();
    timer.Stop();
    return timer.ElapsedMillis() / iterations;
}

template <typename V>
void Run(const RunConfig &c, bool pin)
{
    CsrMatrix<V> csr;
    BuildCsr(c, csr);
    ReportMatrix(c, csr);

    int iterations = c.timing_iterations;
    if (iterations == -1) {
        iterations = AdaptiveIterations(csr.num_nonzeros, 200000ull);
        if (!c.quiet) printf("\t%d timing iterations\n", iterations);
    }
    std::vector<V> x((size_t) csr.num_cols, (V) 1.0), y_in((size_t) csr.num_rows, (V) 1.0), gold((size_t) csr.num_rows),
        y((size_t) csr.num_rows);
    SpmvGold(csr, x.data(), y_in.data(), gold.data(), (V) c.alpha, (V) c.beta);

    const int threads = c.threads > 0 ? c.threads : UsableCpus();
    const int *row_end = csr.row_offsets.data() + 1;

    if (!c.quiet) printf("\n\n");
    printf("OMP-row CsrMV, "); fflush(stdout);
    float avg = TimeMethod(c, csr, x.data(), gold.data(), y.data(), iterations, 8, [&] {
        RowParallelCsrmv(threads, csr.num_rows, csr.row_offsets.data(), csr.column_indices.data(), csr.values.data(), x.data(), y.data());
    });
    DisplayPerf(c.quiet, (int) sizeof(V), 0.0, avg, csr.num_rows, csr.num_nonzeros, -1);

    if (!c.quiet) printf("\n\n");
    printf("Merge CsrMV, "); fflush(stdout);
    if (!c.quiet) printf("\tUsing %d threads on %d procs\n", threads, omp_get_num_procs());
    std::vector<int> carry_row; std::vector<V> carry_val;
    avg = TimeMethod(c, csr, x.data(), gold.data(), y.data(), iterations, 8, [&] {
        MergeCsrmv(threads, csr.num_rows, csr.num_nonzeros, row_end, csr.column_indices.data(), csr.values.data(), x.data(),
                   y.data(), carry_row, carry_val);
    });
    DisplayPerf(c.quiet, (int) sizeof(V), 0.0, avg, csr.num_rows, csr.num_nonzeros, -1);
    if (!c.quiet) DisplayRoofline((int) sizeof(V), avg, csr.num_rows, csr.num_cols, csr.num_nonzeros, -1);
    if (pin && !c.quiet) {                    // extra method line, never in the CSV (its columns are the reference's)
        printf("\n\nMerge CsrMV (socket 0, first touch), "); fflush(stdout);
        double avg_ms = 0; int done = 0, pinned = 0, packages = 0;
        const int st = BenchMerge<V>(threads, 1, csr.num_rows, csr.num_cols, csr.num_nonzeros, csr.row_offsets.data(),
                                     csr.column_indices.data(), csr.values.data(), x.data(), 1e9, iterations, &avg_ms, &done, &pinned,
                                     &packages, y.data());
        if (st != 0) { printf("\tskipped (allocation failed)\n"); return; }
        printf("\t%d threads, %s (%d socket(s) visible)\n", threads,
               pinned ? "one per physical core of socket 0, spread evenly over its cores" : "NOT bound (socket 0 has fewer allowed cores, or affinity refused)", packages);
        const int bad = CompareResultsReferenceRule(y.data(), gold.data(), csr.num_rows, true);
        printf("\t%s\n", bad ? "FAIL" : "PASS");
        DisplayPerf(c.quiet, (int) sizeof(V), 0.0, avg_ms, csr.num_rows, csr.num_nonzeros, -1);
    }
}

}  // namespace

int main(int argc, char **argv)
{
    CommandLineArgs args(argc, argv);
    if (args.CheckCmdLineFlag("help")) {
        printf("%s [--quiet] [--v] [--v2] [--threads=<OMP threads>] [--i=<timing iterations>] [--fp32] "
               "[--alpha=<alpha scalar (default: 1.0)>] [--beta=<beta scalar (default: 0.0)>] [--no-strict] [--cache] [--pin] [--timing]\n"
               "\t--mtx=<matrix market file>\n\t--dense=<cols>\n\t--grid2d=<width>\n\t--grid3d=<width>\n\t--wheel=<spokes>\n",
               argv[0]);
        return 0;
    }
    const RunConfig c = ParseCommon(args, false);
    omp_set_num_threads(c.threads > 0 ? c.threads : UsableCpus());     // also sizes the matrix-building regions
    const bool pin = args.CheckCmdLineFlag("pin");
    if (c.fp32) Run<float>(c, pin); else Run<double>(c, pin);
    printf("\n");
    return 0;
}
