// gpu_spmv.cpp -- the MI355X driver with the reference's gpu_spmv command line
// and report (gpu_spmv.cu:671-741): builds the CSR matrix on the host, uploads
// it, runs the merge-based CsrMV through the drop-in boundary
// (mspmv::DeviceSpmv::CsrMV -> include/mspmv.h) with the reference's protocol
// (size query, allocate, reset y, one warm-up + verification, events around N
// back-to-back calls; gpu_spmv.cu:376-435) and then the vendor library for
// comparison: rocSPARSE CsrMV (analysis time reported as setup) and rocSPARSE
// HybMV, where the reference ran cuSPARSE (gpu_spmv.cu:106-364,565-578).
//
//   gpu_spmv [--device=<id>] [--quiet] [--v] [--v2] [--i=<iterations>] [--fp32]
//            [--alpha=<a>] [--beta=<b>] [--peak-gbs=<GB/s>] [--no-strict] [--no-vendor] [--no-hyb] [--check]
//            [--prepared] [--plan[=<bands>]] [--gpus=<G>[,<G2>...]] [--mg-one-device] [--mg-exchange=peer|rccl]
//            --mtx=<file> | --dense=<cols> [--size=<nnz>] | --grid2d=<w> | --grid3d=<w> | --wheel=<spokes>
//
// Extra method lines of this project (non-quiet only; the CSV keeps the reference's columns):
//   --prepared   the stateless call with the tile coordinates found once (mspmv_csrmv_prepare)
//   --plan       the prepared band-major plan (mspmv_csrmv_plan_*): set-up = the plan build
//   --hotcols    the hot-column plan (mspmv_csrmv_hotcols_*): set-up = ranking the columns by reference count
//   --gpus=G     the matrix merge-partitioned over G GPUs of this node through the C multi-GPU operator
//                (mspmv_mg_plan_*; the reference has a single --device, utils.h:465-474), one line per G;
//                --mg-one-device runs all parts on --device (a functional run on a 1-GPU box)
//
// alpha/beta: the reference parses them but its CsrMV forces 1/0, so any other
// value makes the reference print FAIL; here they are passed to the
// y = alpha*A*x + beta*y extension of the C ABI.
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <cstring>
#include <vector>

#include "device_spmv.hpp"
#include "driver_common.hpp"

using namespace mspmv_host;

#define HIP_OK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int) e_, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)
#define ROCSPARSE_OK(expr)                                                                            \
    do {                                                                                              \
        rocsparse_status s_ = (expr);                                                                 \
        if (s_ != rocsparse_status_success) {                                                         \
            fprintf(stderr, "rocSPARSE error %d at %s:%d\n", (int) s_, __FILE__, __LINE__);          \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)

namespace {

struct GpuTimer {                                       // utils.h:624-658 with hipEvents
    hipEvent_t start, stop;
    GpuTimer() { HIP_OK(hipEventCreate(&start)); HIP_OK(hipEventCreate(&stop)); }
    ~GpuTimer() { (void) hipEventDestroy(start); (void) hipEventDestroy(stop); }
    void Start() { HIP_OK(hipEventRecord(start, 0)); }
    void Stop() { HIP_OK(hipEventRecord(stop, 0)); }
    float ElapsedMillis() { float ms; HIP_OK(hipEventSynchronize(stop)); HIP_OK(hipEventElapsedTime(&ms, start, stop)); return ms; }
};

// --chunk-times=<n> (diagnostic, non-quiet): after a method's timed loop, the same loop again with an event every n calls --
// the time line of the loop, us per call and chunk (where a loop average that moves from run to run comes from: a clock ramp
// at its start, a step in the middle, or isolated long calls)
int g_chunk = 0;
template <typename Call>
void ChunkTimes(int iterations, Call &&call)
{
    if (g_chunk <= 0 || iterations <= 0) return;
    const int chunks = (iterations + g_chunk - 1) / g_chunk;
    std::vector<hipEvent_t> ev((size_t) chunks + 1);
    for (auto &e : ev) HIP_OK(hipEventCreate(&e));
    HIP_OK(hipEventRecord(ev[0], 0));
    for (int k = 0, it = 0; k < chunks; ++k) {
        for (int j = 0; j < g_chunk && it < iterations; ++j, ++it) call();
        HIP_OK(hipEventRecord(ev[(size_t) k + 1], 0));
    }
    HIP_OK(hipEventSynchronize(ev[(size_t) chunks]));
    printf("\tchunk times (us per call, %d calls per chunk):", g_chunk);
    float lo = 1e30f, hi = 0;
    std::vector<float> all;
    for (int k = 0, it = 0; k < chunks; ++k) {
        const int n = std::min(g_chunk, iterations - it); it += n;
        float ms = 0; HIP_OK(hipEventElapsedTime(&ms, ev[(size_t) k], ev[(size_t) k + 1]));
        const float us = ms * 1000.0f / n;
        lo = std::min(lo, us); hi = std::max(hi, us); all.push_back(us);
        printf(" %.2f", us);
    }
    std::sort(all.begin(), all.end());
    printf("\n\tchunk min %.2f us, max %.2f us, median %.2f us\n", lo, hi, all[all.size() / 2]);
    for (auto &e : ev) (void) hipEventDestroy(e);
}

struct Device {
    hipDeviceProp_t prop;
    double giga_bandwidth = 0;
    int id = 0;
};

// args.DeviceInit() (utils.h:451-515): out-of-range ids fall back to device 0.
Device DeviceInit(const RunConfig &c)
{
    Device d;
    int count = 0;
    HIP_OK(hipGetDeviceCount(&count));
    if (count == 0) { fprintf(stderr, "No devices supporting HIP.\n"); exit(1); }
    d.id = (c.device >= 0 && c.device < count) ? c.device : 0;
    HIP_OK(hipSetDevice(d.id));
    HIP_OK(hipGetDeviceProperties(&d.prop, d.id));
    size_t free_b = 0, total_b = 0;
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    // the reference's formula (utils.h:491) underestimates HBM3E; MI355X datasheet peak is 8 TB/s
    d.giga_bandwidth = double(d.prop.memoryBusWidth) * d.prop.memoryClockRate * 2 / 8 / 1000 / 1000;
    if (strstr(d.prop.gcnArchName, "gfx950")) d.giga_bandwidth = 8000.0;
    if (c.peak_gbs > 0) d.giga_bandwidth = c.peak_gbs;
    if (!c.quiet) {
        printf("Using device %d: %s (%s, %d CUs, %lld free / %lld total MB physmem, %.3f GB/s @ %d kHz mem clock, ECC %s)\n",
               d.id, d.prop.name, d.prop.gcnArchName, d.prop.multiProcessorCount, (long long) free_b / 1024 / 1024,
               (long long) total_b / 1024 / 1024, d.giga_bandwidth, d.prop.memoryClockRate, d.prop.ECCEnabled ? "on" : "off");
        fflush(stdout);
    }
    return d;
}

template <typename V>
struct DeviceProblem {
    V *d_values = nullptr, *d_x = nullptr, *d_y = nullptr;
    int *d_row_offsets = nullptr, *d_cols = nullptr;
    int rows = 0, cols = 0, nnz = 0;
    void Upload(const CsrMatrix<V> &a, const std::vector<V> &x)
    {
        rows = a.num_rows; cols = a.num_cols; nnz = a.num_nonzeros;
        HIP_OK(hipMalloc(&d_values, sizeof(V) * std::max(nnz, 1)));
        HIP_OK(hipMalloc(&d_row_offsets, sizeof(int) * (rows + 1)));
        HIP_OK(hipMalloc(&d_cols, sizeof(int) * std::max(nnz, 1)));
        HIP_OK(hipMalloc(&d_x, sizeof(V) * std::max(cols, 1)));
        HIP_OK(hipMalloc(&d_y, sizeof(V) * std::max(rows, 1)));
        HIP_OK(hipMemcpy(d_values, a.values.data(), sizeof(V) * nnz, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_row_offsets, a.row_offsets.data(), sizeof(int) * (rows + 1), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_cols, a.column_indices.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_x, x.data(), sizeof(V) * cols, hipMemcpyHostToDevice));
    }
    ~DeviceProblem()
    {
        (void) hipFree(d_values); (void) hipFree(d_row_offsets); (void) hipFree(d_cols); (void) hipFree(d_x); (void) hipFree(d_y);
    }
};

template <typename V>
void Verify(const RunConfig &c, const CsrMatrix<V> &a, const std::vector<V> &x, const std::vector<V> &gold, const V *d_y,
            bool alpha_beta_default)
{
    std::vector<V> y((size_t) a.num_rows);
    HIP_OK(hipMemcpy(y.data(), d_y, sizeof(V) * a.num_rows, hipMemcpyDeviceToHost));
    if (c.verbose) {
        printf("\nReference / computed:\n");
        for (int r = 0; r < a.num_rows; ++r) printf("[%d] %g %g\n", r, (double) gold[r], (double) y[r]);
    }
    const int bad = CompareResultsReferenceRule(y.data(), gold.data(), a.num_rows, true);
    printf("\t%s\n", bad ? "FAIL" : "PASS");
    if (c.strict && alpha_beta_default) {
        double worst = 0;
        const long long v = StrictCheck(a.num_rows, a.row_offsets.data(), a.column_indices.data(), a.values.data(), x.data(),
                                        y.data(), 16, &worst);
        printf("\tstrict check: %s (%lld rows outside tolerance, worst ratio %.3g)\n", v ? "FAIL" : "PASS", v, worst);
    }
    fflush(stdout);
}

// --quiet --check (corpus sweeps, tools/corpus_sweep.sh): the strict check without touching the CSV on stdout
static int g_check_failures = 0;
template <typename V>
void QuietCheck(const RunConfig &c, const CsrMatrix<V> &a, const std::vector<V> &x, const V *d_y)
{
    std::vector<V> y((size_t) a.num_rows);
    HIP_OK(hipMemcpy(y.data(), d_y, sizeof(V) * a.num_rows, hipMemcpyDeviceToHost));
    double worst = 0;
    const long long v = StrictCheck(a.num_rows, a.row_offsets.data(), a.column_indices.data(), a.values.data(), x.data(), y.data(), 16, &worst);
    fprintf(stderr, "strict-check, %s, %s, %s, %lld, %.4g\n", c.mtx.empty() ? "generated" : c.mtx.c_str(), sizeof(V) > 4 ? "fp64" : "fp32", v ? "FAIL" : "PASS", v, worst);
    if (v) ++g_check_failures;
}

// TestGpuMergeCsrmv (gpu_spmv.cu:376-435)
template <typename V>
float TestMerge(const RunConfig &c, const CsrMatrix<V> &a, const std::vector<V> &x, const std::vector<V> &y_in,
                const std::vector<V> &gold, DeviceProblem<V> &p, int iterations, float &setup_ms, bool prepared = false)
{
    setup_ms = 0;
    const bool plain = c.alpha == 1.0f && c.beta == 0.0f;
    auto call = [&](void *temp, size_t &bytes, bool debug) {
        // --prepared (extension): the tile coordinates were found once, outside the timed loop
        if (prepared && temp)
            return mspmv::DeviceSpmv::CsrMVPrepared(temp, bytes, p.d_values, p.d_row_offsets, p.d_cols, p.d_x, p.d_y, p.rows,
                                                    p.cols, p.nnz, (V) c.alpha, (V) c.beta, (hipStream_t) 0, debug);
        return plain ? mspmv::DeviceSpmv::CsrMV(temp, bytes, p.d_values, p.d_row_offsets, p.d_cols, p.d_x, p.d_y, p.rows, p.cols,
                                                p.nnz, (hipStream_t) 0, debug)
                     : mspmv::DeviceSpmv::CsrMV(temp, bytes, p.d_values, p.d_row_offsets, p.d_cols, p.d_x, p.d_y, p.rows, p.cols,
                                                p.nnz, (V) c.alpha, (V) c.beta, (hipStream_t) 0, debug);
    };
    size_t temp_bytes = 0;
    void *d_temp = nullptr;
    HIP_OK(call(nullptr, temp_bytes, false));
    HIP_OK(hipMalloc(&d_temp, temp_bytes));
    if (prepared) {
        GpuTimer setup; setup.Start();
        HIP_OK(mspmv::DeviceSpmv::CsrMVPrepare<V>(d_temp, temp_bytes, p.d_row_offsets, p.rows, p.nnz));
        setup.Stop(); setup_ms = setup.ElapsedMillis();
    }
    HIP_OK(hipMemcpy(p.d_y, y_in.data(), sizeof(V) * p.rows, hipMemcpyHostToDevice));
    HIP_OK(call(d_temp, temp_bytes, !c.quiet));                      // warm-up (+ launch log, like debug_synchronous)
    if (!c.quiet) Verify(c, a, x, gold, p.d_y, plain);
    else if (c.check && plain && !prepared) QuietCheck(c, a, x, p.d_y);
    GpuTimer timer;
    timer.Start();
    for (int it = 0; it < iterations; ++it) HIP_OK(call(d_temp, temp_bytes, false));
    timer.Stop();
    const float ms = timer.ElapsedMillis() / iterations;
    if (!c.quiet) ChunkTimes(iterations, [&]() { HIP_OK(call(d_temp, temp_bytes, false)); });
    HIP_OK(hipFree(d_temp));
    return ms;
}

// the prepared band-major plan (extension): set-up = mspmv_csrmv_plan_build_*, then the same protocol
template <typename V>
float TestPlan(const RunConfig &c, const CsrMatrix<V> &a, const std::vector<V> &x, const std::vector<V> &y_in,
               const std::vector<V> &gold, DeviceProblem<V> &p, int iterations, float &setup_ms, int bands, int &bands_used)
{
    size_t bytes = 0;
    HIP_OK(mspmv::DeviceSpmv::PlanSize<V>(p.rows, p.cols, p.nnz, bands, bytes, bands_used));
    void *d_plan = nullptr;
    HIP_OK(hipMalloc(&d_plan, bytes));
    GpuTimer setup; setup.Start();
    HIP_OK(mspmv::DeviceSpmv::PlanBuild(d_plan, bytes, p.d_values, p.d_row_offsets, p.d_cols, p.rows, p.cols, p.nnz, bands_used));
    setup.Stop(); setup_ms = setup.ElapsedMillis();
    HIP_OK(hipMemcpy(p.d_y, y_in.data(), sizeof(V) * p.rows, hipMemcpyHostToDevice));
    HIP_OK(mspmv::DeviceSpmv::PlanApply(d_plan, bytes, p.d_x, p.d_y, p.rows, p.cols, p.nnz, bands_used, (V) c.alpha, (V) c.beta, 0, !c.quiet));
    if (!c.quiet) Verify(c, a, x, gold, p.d_y, c.alpha == 1.0f && c.beta == 0.0f);
    GpuTimer timer; timer.Start();
    for (int it = 0; it < iterations; ++it)
        HIP_OK(mspmv::DeviceSpmv::PlanApply(d_plan, bytes, p.d_x, p.d_y, p.rows, p.cols, p.nnz, bands_used, (V) c.alpha, (V) c.beta));
    timer.Stop();
    const float ms = timer.ElapsedMillis() / iterations;
    HIP_OK(hipFree(d_plan));
    return ms;
}

// the hot-column plan (extension): set-up = mspmv_csrmv_hotcols_build, then the same protocol (x permuted inside every timed call)
template <typename V>
float TestHotColumns(const RunConfig &c, const CsrMatrix<V> &a, const std::vector<V> &x, const std::vector<V> &y_in,
                     const std::vector<V> &gold, DeviceProblem<V> &p, int iterations, float &setup_ms)
{
    size_t bytes = 0;
    HIP_OK((hipError_t) mspmv_csrmv_hotcols_size(p.rows, p.cols, p.nnz, (int) sizeof(V), &bytes));
    void *d_plan = nullptr;
    HIP_OK(hipMalloc(&d_plan, bytes));
    auto apply = [&](int debug) {
        if constexpr (sizeof(V) == 4) return (hipError_t) mspmv_csrmv_hotcols_apply_f32(d_plan, bytes, p.d_values, p.d_row_offsets, p.d_x, p.d_y, p.rows, p.cols, p.nnz, (float) c.alpha, (float) c.beta, nullptr, debug);
        else return (hipError_t) mspmv_csrmv_hotcols_apply_f64(d_plan, bytes, p.d_values, p.d_row_offsets, p.d_x, p.d_y, p.rows, p.cols, p.nnz, (double) c.alpha, (double) c.beta, nullptr, debug);
    };
    GpuTimer setup; setup.Start();
    HIP_OK((hipError_t) mspmv_csrmv_hotcols_build(d_plan, bytes, p.d_row_offsets, p.d_cols, p.rows, p.cols, p.nnz, (int) sizeof(V), nullptr, 0));
    setup.Stop(); setup_ms = setup.ElapsedMillis();
    HIP_OK(hipMemcpy(p.d_y, y_in.data(), sizeof(V) * p.rows, hipMemcpyHostToDevice));
    HIP_OK(apply(!c.quiet));
    if (!c.quiet) Verify(c, a, x, gold, p.d_y, c.alpha == 1.0f && c.beta == 0.0f);
    GpuTimer timer; timer.Start();
    for (int it = 0; it < iterations; ++it) HIP_OK(apply(0));
    timer.Stop();
    const float ms = timer.ElapsedMillis() / iterations;
    HIP_OK(hipFree(d_plan));
    return ms;
}

// the matrix cut into G swaths of the merge path, one per GPU (or all on one device), through the C multi-GPU
// operator: y = A*x per step = every part's CsrMV + one carry exchange, below the C ABI.  Timed on the host
// clock around N back-to-back steps + a plan-wide synchronise (events of one device cannot bracket G streams).
template <typename V>
float TestMultiGpu(const RunConfig &c, const CsrMatrix<V> &a, const std::vector<V> &x, const std::vector<V> &gold, int parts,
                   bool one_device, int exchange, int iterations, float &setup_ms, int &exchange_used)
{
    int ndev = 0; HIP_OK(hipGetDeviceCount(&ndev));
    if (!one_device && parts > ndev) { printf("\t(skipped: %d GPUs visible)\n", ndev); setup_ms = 0; return -1.f; }
    CpuTimer setup; setup.Start();
    std::vector<int64_t> off64((size_t) a.num_rows + 1), row_split((size_t) parts + 1), nz_split((size_t) parts + 1);
    for (size_t i = 0; i < off64.size(); ++i) off64[i] = a.row_offsets[i];
    HIP_OK((hipError_t) mspmv_mg_partition(off64.data(), a.num_rows, a.num_nonzeros, parts, row_split.data(), nz_split.data()));
    std::vector<int32_t> ids((size_t) parts), devs((size_t) parts);
    for (int g = 0; g < parts; ++g) { ids[g] = g; devs[g] = one_device ? c.device : g; }
    mspmv_mg_plan_t *plan = nullptr;
    HIP_OK((hipError_t) mspmv_mg_plan_create(&plan, parts, parts, ids.data(), devs.data(), row_split.data(), nz_split.data(), a.num_cols,
                                             (int) sizeof(V), exchange, nullptr));
    std::vector<void *> owned;                       // the parts' CSR arrays (caller-owned, per the plan's contract)
    for (int g = 0; g < parts; ++g) {
        HIP_OK(hipSetDevice(devs[g]));
        const int64_t lr = row_split[g + 1] - row_split[g] + 1, ln = nz_split[g + 1] - nz_split[g];
        std::vector<int32_t> lo((size_t) lr + 1);
        HIP_OK((hipError_t) mspmv_mg_local_offsets(off64.data(), a.num_rows, row_split[g], row_split[g + 1], nz_split[g], nz_split[g + 1], lo.data()));
        void *d_off = nullptr, *d_col = nullptr, *d_val = nullptr;
        HIP_OK(hipMalloc(&d_off, sizeof(int32_t) * lo.size()));
        HIP_OK(hipMalloc(&d_col, sizeof(int32_t) * std::max<int64_t>(ln, 1)));
        HIP_OK(hipMalloc(&d_val, sizeof(V) * std::max<int64_t>(ln, 1)));
        HIP_OK(hipMemcpy(d_off, lo.data(), sizeof(int32_t) * lo.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_col, a.column_indices.data() + nz_split[g], sizeof(int32_t) * ln, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_val, a.values.data() + nz_split[g], sizeof(V) * ln, hipMemcpyHostToDevice));
        HIP_OK((hipError_t) mspmv_mg_plan_set_part(plan, g, d_val, (const int32_t *) d_off, (const int32_t *) d_col));
        HIP_OK(hipMemcpy(mspmv_mg_plan_x(plan, g), x.data(), sizeof(V) * a.num_cols, hipMemcpyHostToDevice));
        owned.push_back(d_off); owned.push_back(d_col); owned.push_back(d_val);
    }
    setup.Stop(); setup_ms = setup.ElapsedMillis();
    mspmv_mg_info_t info; HIP_OK((hipError_t) mspmv_mg_plan_info(plan, &info)); exchange_used = info.exchange;
    HIP_OK((hipError_t) mspmv_mg_csrmv(plan));
    HIP_OK((hipError_t) mspmv_mg_synchronize(plan));
    if (!c.quiet) {
        std::vector<V> y((size_t) a.num_rows);
        for (int g = 0; g < parts; ++g) {
            HIP_OK(hipSetDevice(devs[g]));
            HIP_OK(hipMemcpy(y.data() + row_split[g], mspmv_mg_plan_y(plan, g), sizeof(V) * (row_split[g + 1] - row_split[g]), hipMemcpyDeviceToHost));
        }
        const int bad = CompareResultsReferenceRule(y.data(), gold.data(), a.num_rows, true);
        printf("\t%s\n", bad ? "FAIL" : "PASS");
        if (c.strict) {
            double worst = 0;
            const long long v = StrictCheck(a.num_rows, a.row_offsets.data(), a.column_indices.data(), a.values.data(), x.data(), y.data(),
                                            16 + parts, &worst);
            printf("\tstrict check: %s (%lld rows outside tolerance, worst ratio %.3g)\n", v ? "FAIL" : "PASS", v, worst);
        }
    }
    CpuTimer timer; timer.Start();
    for (int it = 0; it < iterations; ++it) HIP_OK((hipError_t) mspmv_mg_csrmv(plan));
    HIP_OK((hipError_t) mspmv_mg_synchronize(plan));
    timer.Stop();
    const float ms = timer.ElapsedMillis() / iterations;
    HIP_OK((hipError_t) mspmv_mg_plan_destroy(plan));
    for (size_t i = 0; i < owned.size(); ++i) { HIP_OK(hipSetDevice(devs[i / 3])); HIP_OK(hipFree(owned[i])); }
    HIP_OK(hipSetDevice(c.device));
    return ms;
}

template <typename V> struct Roc;
template <> struct Roc<float> {
    static constexpr auto analysis = rocsparse_scsrmv_analysis; static constexpr auto csrmv = rocsparse_scsrmv;
    static constexpr auto csr2hyb = rocsparse_scsr2hyb; static constexpr auto hybmv = rocsparse_shybmv;
};
template <> struct Roc<double> {
    static constexpr auto analysis = rocsparse_dcsrmv_analysis; static constexpr auto csrmv = rocsparse_dcsrmv;
    static constexpr auto csr2hyb = rocsparse_dcsr2hyb; static constexpr auto hybmv = rocsparse_dhybmv;
};

// rocSPARSE CsrMV, the counterpart of TestCusparseCsrmv (gpu_spmv.cu:262-364); setup = analysis
template <typename V>
float TestRocsparseCsrmv(const RunConfig &c, const CsrMatrix<V> &a, const std::vector<V> &x, const std::vector<V> &y_in,
                         const std::vector<V> &gold, DeviceProblem<V> &p, int iterations, float &setup_ms,
                         rocsparse_handle handle)
{
    rocsparse_mat_descr descr; rocsparse_mat_info info;
    ROCSPARSE_OK(rocsparse_create_mat_descr(&descr));
    ROCSPARSE_OK(rocsparse_create_mat_info(&info));
    CpuTimer setup; HIP_OK(hipDeviceSynchronize()); setup.Start();
    ROCSPARSE_OK(Roc<V>::analysis(handle, rocsparse_operation_none, p.rows, p.cols, p.nnz, descr, p.d_values, p.d_row_offsets,
                                  p.d_cols, info));
    HIP_OK(hipDeviceSynchronize()); setup.Stop();
    setup_ms = setup.ElapsedMillis();
    const V alpha = (V) c.alpha, beta = (V) c.beta;
    HIP_OK(hipMemcpy(p.d_y, y_in.data(), sizeof(V) * p.rows, hipMemcpyHostToDevice));
    ROCSPARSE_OK(Roc<V>::csrmv(handle, rocsparse_operation_none, p.rows, p.cols, p.nnz, &alpha, descr, p.d_values,
                               p.d_row_offsets, p.d_cols, info, p.d_x, &beta, p.d_y));
    if (!c.quiet) Verify(c, a, x, gold, p.d_y, c.alpha == 1.0f && c.beta == 0.0f);
    GpuTimer timer; timer.Start();
    for (int it = 0; it < iterations; ++it)
        ROCSPARSE_OK(Roc<V>::csrmv(handle, rocsparse_operation_none, p.rows, p.cols, p.nnz, &alpha, descr, p.d_values,
                                   p.d_row_offsets, p.d_cols, info, p.d_x, &beta, p.d_y));
    timer.Stop();
    const float ms = timer.ElapsedMillis() / iterations;
    if (!c.quiet) ChunkTimes(iterations, [&]() {
        ROCSPARSE_OK(Roc<V>::csrmv(handle, rocsparse_operation_none, p.rows, p.cols, p.nnz, &alpha, descr, p.d_values,
                                   p.d_row_offsets, p.d_cols, info, p.d_x, &beta, p.d_y)); });
    ROCSPARSE_OK(rocsparse_destroy_mat_info(info));
    ROCSPARSE_OK(rocsparse_destroy_mat_descr(descr));
    return ms;
}

// rocSPARSE HybMV, the counterpart of TestCusparseHybmv (gpu_spmv.cu:106-257); setup = CSR -> HYB
template <typename V>
float TestRocsparseHybmv(const RunConfig &c, const CsrMatrix<V> &a, const std::vector<V> &x, const std::vector<V> &y_in,
                         const std::vector<V> &gold, DeviceProblem<V> &p, int iterations, float &setup_ms,
                         rocsparse_handle handle)
{
    rocsparse_mat_descr descr; rocsparse_hyb_mat hyb;
    ROCSPARSE_OK(rocsparse_create_mat_descr(&descr));
    ROCSPARSE_OK(rocsparse_create_hyb_mat(&hyb));
    CpuTimer setup; HIP_OK(hipDeviceSynchronize()); setup.Start();
    ROCSPARSE_OK(Roc<V>::csr2hyb(handle, p.rows, p.cols, descr, p.d_values, p.d_row_offsets, p.d_cols, hyb, 0,
                                 rocsparse_hyb_partition_auto));
    HIP_OK(hipDeviceSynchronize()); setup.Stop();
    setup_ms = setup.ElapsedMillis();
    const V alpha = (V) c.alpha, beta = (V) c.beta;
    HIP_OK(hipMemcpy(p.d_y, y_in.data(), sizeof(V) * p.rows, hipMemcpyHostToDevice));   // (the reference copies sizeof(float) here, gpu_spmv.cu:213)
    ROCSPARSE_OK(Roc<V>::hybmv(handle, rocsparse_operation_none, &alpha, descr, hyb, p.d_x, &beta, p.d_y));
    if (!c.quiet) Verify(c, a, x, gold, p.d_y, c.alpha == 1.0f && c.beta == 0.0f);
    GpuTimer timer; timer.Start();
    for (int it = 0; it < iterations; ++it)
        ROCSPARSE_OK(Roc<V>::hybmv(handle, rocsparse_operation_none, &alpha, descr, hyb, p.d_x, &beta, p.d_y));
    timer.Stop();
    const float ms = timer.ElapsedMillis() / iterations;
    ROCSPARSE_OK(rocsparse_destroy_hyb_mat(hyb));
    ROCSPARSE_OK(rocsparse_destroy_mat_descr(descr));
    return ms;
}

struct Extras { bool vendor = true, hyb = true, prepared = false, plan = false, hotcols = false, mg_one_device = false; int plan_bands = 0, mg_exchange = MSPMV_MG_EXCHANGE_AUTO; std::vector<int> gpus; };

template <typename V>
void Run(const RunConfig &c, const Device &dev, const Extras &ex)
{
    const bool vendor = ex.vendor, prepared_too = ex.prepared;
    CsrMatrix<V> csr;
    BuildCsr<V>(c, csr, [](const RunConfig &cc, int nnz) {
        const int it = cc.timing_iterations == -1 ? AdaptiveIterations(nnz, 50000ull) : cc.timing_iterations;
        if (!cc.quiet) printf("\t%d timing iterations\n", it);          // gpu_spmv.cu:495-496 (printed before the conversion)
    });
    const int iterations = c.timing_iterations == -1 ? AdaptiveIterations(csr.num_nonzeros, 50000ull) : c.timing_iterations;
    ReportMatrix(c, csr);

    std::vector<V> x((size_t) csr.num_cols, (V) 1.0), y_in((size_t) csr.num_rows, (V) 1.0), gold((size_t) csr.num_rows);
    SpmvGold(csr, x.data(), y_in.data(), gold.data(), (V) c.alpha, (V) c.beta);
    if (c.quiet) {   // gpu_spmv.cu:532-534 (the marketing name can be empty on headless boxes: fall back to the arch)
        printf("%s, %s, ", dev.prop.name[0] ? dev.prop.name : dev.prop.gcnArchName, sizeof(V) > 4 ? "fp64" : "fp32");
        fflush(stdout);
    }

    DeviceProblem<V> p;
    p.Upload(csr, x);
    float setup_ms = 0, avg_ms = 0;

    if (!c.quiet) printf("\n\n");
    printf("Merge-based CsrMV, "); fflush(stdout);
    avg_ms = TestMerge(c, csr, x, y_in, gold, p, iterations, setup_ms);
    DisplayPerf(c.quiet, (int) sizeof(V), setup_ms, avg_ms, csr.num_rows, csr.num_nonzeros, dev.giga_bandwidth);
    if (!c.quiet) DisplayRoofline((int) sizeof(V), avg_ms, csr.num_rows, csr.num_cols, csr.num_nonzeros, dev.giga_bandwidth);
    if (prepared_too && !c.quiet) {            // extra method line, never in the CSV (its column layout is the reference's)
        printf("\n\nMerge-based CsrMV (prepared: coordinates found once), "); fflush(stdout);
        avg_ms = TestMerge(c, csr, x, y_in, gold, p, iterations, setup_ms, true);
        DisplayPerf(c.quiet, (int) sizeof(V), setup_ms, avg_ms, csr.num_rows, csr.num_nonzeros, dev.giga_bandwidth);
    }

    if (ex.plan && !c.quiet) {                  // extra method line, never in the CSV
        int used = 0;
        printf("\n\nMerge-based CsrMV (prepared band-major plan), "); fflush(stdout);
        avg_ms = TestPlan(c, csr, x, y_in, gold, p, iterations, setup_ms, ex.plan_bands, used);
        printf("\t%d column band(s)\n", used);
        DisplayPerf(c.quiet, (int) sizeof(V), setup_ms, avg_ms, csr.num_rows, csr.num_nonzeros, dev.giga_bandwidth);
        DisplayRoofline((int) sizeof(V), avg_ms, csr.num_rows, csr.num_cols, csr.num_nonzeros, dev.giga_bandwidth);
    }
    if (ex.hotcols && !c.quiet) {               // extra method line, never in the CSV
        printf("\n\nMerge-based CsrMV (hot-column plan: columns renumbered by reference count), "); fflush(stdout);
        avg_ms = TestHotColumns(c, csr, x, y_in, gold, p, iterations, setup_ms);
        DisplayPerf(c.quiet, (int) sizeof(V), setup_ms, avg_ms, csr.num_rows, csr.num_nonzeros, dev.giga_bandwidth);
        DisplayRoofline((int) sizeof(V), avg_ms, csr.num_rows, csr.num_cols, csr.num_nonzeros, dev.giga_bandwidth);
    }
    if (!c.quiet && c.alpha == 1.0f && c.beta == 0.0f)
        for (int parts : ex.gpus) {
            int used = 0;
            printf("\n\nMerge-based CsrMV (%d GPU%s%s), ", parts, parts == 1 ? "" : "s", ex.mg_one_device ? ", all parts on one device" : ""); fflush(stdout);
            avg_ms = TestMultiGpu(c, csr, x, gold, parts, ex.mg_one_device, ex.mg_exchange, iterations, setup_ms, used);
            if (avg_ms < 0) continue;
            printf("\tcarry exchange: %s, %d bytes per step\n", used == MSPMV_MG_EXCHANGE_PEER ? "peer reads" : "RCCL all-gather",
                   parts * (int) sizeof(V));
            DisplayPerf(c.quiet, (int) sizeof(V), setup_ms, avg_ms, csr.num_rows, csr.num_nonzeros, dev.giga_bandwidth * parts);
        }

    if (vendor) {
        rocsparse_handle handle;
        ROCSPARSE_OK(rocsparse_create_handle(&handle));
        if (!c.quiet) printf("\n\n");
        printf("rocSPARSE CsrMV, "); fflush(stdout);
        avg_ms = TestRocsparseCsrmv(c, csr, x, y_in, gold, p, iterations, setup_ms, handle);
        DisplayPerf(c.quiet, (int) sizeof(V), setup_ms, avg_ms, csr.num_rows, csr.num_nonzeros, dev.giga_bandwidth);
        if (!c.quiet) printf("\n\n");
        if (ex.hyb) {
            printf("rocSPARSE HybMV, "); fflush(stdout);
            avg_ms = TestRocsparseHybmv(c, csr, x, y_in, gold, p, iterations, setup_ms, handle);
            DisplayPerf(c.quiet, (int) sizeof(V), setup_ms, avg_ms, csr.num_rows, csr.num_nonzeros, dev.giga_bandwidth);
        }
        ROCSPARSE_OK(rocsparse_destroy_handle(handle));
    }
}

}  // namespace

int main(int argc, char **argv)
{
    omp_set_num_threads(UsableCpus());      // host-side matrix construction and the gold SpMV
    CommandLineArgs args(argc, argv);
    if (args.CheckCmdLineFlag("help")) {
        printf("%s [--csrmv | --hybmv | --bsrmv ] [--device=<device-id>] [--quiet] [--v] [--i=<timing iterations>] [--fp32] "
               "[--alpha=<alpha scalar (default: 1.0)>] [--beta=<beta scalar (default: 0.0)>] [--peak-gbs=<GB/s>] "
               "[--no-strict] [--no-vendor] [--no-hyb] [--check] [--cache] [--prepared] [--plan[=<bands>]] [--hotcols] [--gpus=<G>[,<G2>...]] [--mg-one-device] "
               "[--mg-exchange=peer|rccl] [--chunk-times=<calls per chunk>]\n"
               "\t--mtx=<matrix market file> \n\t--dense=<cols>\n\t--grid2d=<width>\n\t--grid3d=<width>\n\t--wheel=<spokes>\n",
               argv[0]);
        return 0;
    }
    const RunConfig c = ParseCommon(args, true);
    const Device dev = DeviceInit(c);
    Extras ex;
    ex.vendor = !args.CheckCmdLineFlag("no-vendor");
    ex.hyb = !args.CheckCmdLineFlag("no-hyb");          // (the CSR -> HYB conversion takes seconds on some small matrices)
    ex.prepared = args.CheckCmdLineFlag("prepared");
    ex.hotcols = args.CheckCmdLineFlag("hotcols");
    ex.plan = args.CheckCmdLineFlag("plan");
    args.GetCmdLineArgument("plan", ex.plan_bands);
    ex.mg_one_device = args.CheckCmdLineFlag("mg-one-device");
    args.GetCmdLineArgument("chunk-times", g_chunk);
    std::string gpus, exchange;
    args.GetCmdLineArgument("gpus", gpus);
    args.GetCmdLineArgument("mg-exchange", exchange);
    if (exchange == "peer") ex.mg_exchange = MSPMV_MG_EXCHANGE_PEER; else if (exchange == "rccl") ex.mg_exchange = MSPMV_MG_EXCHANGE_RCCL;
    for (size_t i = 0; i < gpus.size();) {
        size_t j = gpus.find(',', i); if (j == std::string::npos) j = gpus.size();
        const int g = atoi(gpus.substr(i, j - i).c_str());
        if (g >= 1 && g <= MSPMV_MG_MAX_PARTS) ex.gpus.push_back(g);
        i = j + 1;
    }
    if (c.fp32) Run<float>(c, dev, ex); else Run<double>(c, dev, ex);
    HIP_OK(hipDeviceSynchronize());
    printf("\n");
    return g_check_failures ? 2 : 0;
}
