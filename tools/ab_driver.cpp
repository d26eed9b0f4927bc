// tools/ab_driver.cpp -- same-process, same-box A/B of the per-call latency of small CsrMV calls in the reference's timing protocol
// (N back-to-back calls between two events, gpu_spmv.cu:418-434): two builds of libmspmv.so (dlopen'ed side by side) and rocSPARSE's
// analysed csrmv on the SAME device arrays, loops interleaved A B R A B R ..., the median and the minimum of the loop averages
// reported.  Boxes differ by 0.1-0.3 us on these figures and single loops catch 10 ms stalls of the box (profiles/r05_small_chunk_times*):
// a change of 0.1 us can only be read from interleaved loops in one process.
//   build: make -C tools ab_driver
//   run:   tools/ab_driver <libA.so> <libB.so> [--fp32] [--loops=9] [--calls=2000] [--tune-b=<compact tiles for B: -1 never>] [--env-a=N=V] [--env-b=N=V] [sizes ...]
// With <libB.so> = <libA.so> and --tune-b=-1 it compares the compact front end with the general kernel inside one library.
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define RK(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { fprintf(stderr, "%s: rocsparse %d\n", #x, (int) s_); exit(1); } } while (0)

template <typename V> struct Api {
    int (*csrmv)(void *, size_t *, const V *, const int *, const int *, const V *, V *, int, int, int, void *, int) = nullptr;
    int (*set_compact)(int) = nullptr;
    int (*set_tuning)(int, int, int, int) = nullptr;
};
template <typename V>
static Api<V> load(const char *path, bool f32)
{
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(1); }
    Api<V> a;
    a.csrmv = (decltype(a.csrmv)) dlsym(h, f32 ? "mspmv_csrmv_f32" : "mspmv_csrmv_f64");
    a.set_compact = (decltype(a.set_compact)) dlsym(h, "mspmv_set_compact_tiles");      // (absent in builds before round 5)
    a.set_tuning = (decltype(a.set_tuning)) dlsym(h, "mspmv_set_tuning");
    if (!a.csrmv) { fprintf(stderr, "%s: no mspmv_csrmv\n", path); exit(1); }
    return a;
}

template <typename V> struct Roc;
template <> struct Roc<float> { static constexpr auto analysis = rocsparse_scsrmv_analysis; static constexpr auto csrmv = rocsparse_scsrmv; };
template <> struct Roc<double> { static constexpr auto analysis = rocsparse_dcsrmv_analysis; static constexpr auto csrmv = rocsparse_dcsrmv; };

static double g_host_us = 0;           // host time per call of the last loop: first call entered -> last call returned (before the sync)
template <typename F>
static float loop_us(int calls, F &&f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    const auto h0 = std::chrono::steady_clock::now();
    for (int i = 0; i < calls; ++i) f();
    g_host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / calls;
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1000.0f / calls;
}
static void stats(std::vector<float> v, float &med, float &mn) { std::sort(v.begin(), v.end()); med = v[v.size() / 2]; mn = v[0]; }

template <typename V>
static void run(const char *pa, const char *pb, bool f32, int loops, int calls, int tune_b, int ipt_b, const std::vector<int> &sizes, const char *env_a, const char *env_b)
{
    Api<V> A = load<V>(pa, f32), B = load<V>(pb, f32);
    // --ipt-b=<n>: B runs the 256 x n tile shape whatever the size (mspmv_set_tuning; B must be another FILE than A: the tuning is per library instance)
    if (ipt_b > 0 && B.set_tuning && strcmp(pa, pb) != 0) B.set_tuning((int) sizeof(V), 256, ipt_b, 0);
    rocsparse_handle handle; RK(rocsparse_create_handle(&handle));
    if (env_a || env_b) printf("# environment at first use: A %s | B %s\n", env_a ? env_a : "-", env_b ? env_b : "-");
    printf("# A = %s | B = %s%s | R = rocSPARSE csrmv after analysis; %s, %d loops of %d calls each, interleaved; us per call: median (min)\n", pa, pb,
           tune_b ? (std::string(" with mspmv_set_compact_tiles(") + std::to_string(tune_b) + ")").c_str() : "", f32 ? "fp32" : "fp64", loops, calls);
    for (int w : sizes) {
        // the reference's --grid2d=<w> (sparse_matrix.h:461-526): w^2 vertices, neighbours W, E, N, S, no self loop, values 1
        const int rows = w * w;
        std::vector<int> off(rows + 1), col; col.reserve((size_t) rows * 4);
        for (int i = 0; i < w; ++i) for (int j = 0; j < w; ++j) {
            const int me = i * w + j; off[me] = (int) col.size();
            if (j > 0) col.push_back(me - 1);
            if (j + 1 < w) col.push_back(me + 1);
            if (i > 0) col.push_back(me - w);
            if (i + 1 < w) col.push_back(me + w);
        }
        off[rows] = (int) col.size();
        const int nnz = (int) col.size();
        std::vector<V> val((size_t) nnz, (V) 1), x((size_t) rows);
        for (int i = 0; i < rows; ++i) x[i] = (V) (1.0 + (i % 7) * 0.125);
        int *d_off, *d_col; V *d_val, *d_x, *d_y[3];
        CK(hipMalloc(&d_off, (rows + 1) * 4)); CK(hipMalloc(&d_col, nnz * 4)); CK(hipMalloc(&d_val, nnz * sizeof(V))); CK(hipMalloc(&d_x, rows * sizeof(V)));
        for (auto &p : d_y) CK(hipMalloc(&p, rows * sizeof(V)));
        CK(hipMemcpy(d_off, off.data(), (rows + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_val, val.data(), nnz * sizeof(V), hipMemcpyHostToDevice)); CK(hipMemcpy(d_x, x.data(), rows * sizeof(V), hipMemcpyHostToDevice));
        size_t ta = 0, tb = 0; void *wa, *wb;
        // (the temp storages sit between guard zones of GUARD bytes of 0xA5, checked after the loops: a write outside [wa, wa + ta) shows)
        const size_t GUARD = 4096; char *ga, *gb;
        CK((hipError_t) A.csrmv(nullptr, &ta, d_val, d_off, d_col, d_x, d_y[0], rows, rows, nnz, nullptr, 0)); CK(hipMalloc(&ga, ta + 2 * GUARD));
        CK((hipError_t) B.csrmv(nullptr, &tb, d_val, d_off, d_col, d_x, d_y[1], rows, rows, nnz, nullptr, 0)); CK(hipMalloc(&gb, tb + 2 * GUARD));
        CK(hipMemset(ga, 0xA5, ta + 2 * GUARD)); CK(hipMemset(gb, 0xA5, tb + 2 * GUARD));
        wa = ga + GUARD; wb = gb + GUARD;
        std::vector<V> yh((size_t) rows);                      // the sequential definition on the host
        for (int r = 0; r < rows; ++r) { V t = 0; for (int k = off[r]; k < off[r + 1]; ++k) t += val[k] * x[col[k]]; yh[r] = t; }
        rocsparse_mat_descr descr; rocsparse_mat_info info; RK(rocsparse_create_mat_descr(&descr)); RK(rocsparse_create_mat_info(&info));
        RK(Roc<V>::analysis(handle, rocsparse_operation_none, rows, rows, nnz, descr, d_val, d_off, d_col, info));
        const V one = 1, zero = 0;
        auto fa = [&] { A.csrmv(wa, &ta, d_val, d_off, d_col, d_x, d_y[0], rows, rows, nnz, nullptr, 0); };
        auto fb = [&] { B.csrmv(wb, &tb, d_val, d_off, d_col, d_x, d_y[1], rows, rows, nnz, nullptr, 0); };
        auto fr = [&] { Roc<V>::csrmv(handle, rocsparse_operation_none, rows, rows, nnz, &one, descr, d_val, d_off, d_col, info, d_x, &zero, d_y[2]); };
        {   // rocSPARSE BEFORE any call of ours has run on this matrix
            fr(); std::vector<V> y0(rows); CK(hipMemcpy(y0.data(), d_y[2], rows * sizeof(V), hipMemcpyDeviceToHost));
            for (int i = 0; i < rows; ++i) if (std::abs((double) y0[i] - (double) yh[i]) > 1e-3 * std::abs((double) yh[i])) {
                printf("# rocSPARSE's FIRST call after its analysis, before any call of ours: row %d = %.6g, the sequential sum is %.6g\n", i, (double) y0[i], (double) yh[i]); break; }
        }
        // --env-a=NAME=VALUE / --env-b=...: put into the environment right before that library's FIRST launching call (the libraries
        // read their re-tuning variables once, at first use; A and B must be two FILES then -- two instances)
        auto put = [](const char *kv) { if (!kv) return; std::string t(kv); const size_t eq = t.find('='); if (eq != std::string::npos) setenv(t.substr(0, eq).c_str(), t.substr(eq + 1).c_str(), 1); };
        put(env_a); fa(); put(env_b); fb();
        for (int i = 0; i < 50; ++i) { fa(); fb(); fr(); }
        CK(hipDeviceSynchronize());
        std::vector<float> va, vb, vr;
        double ha = 0, hb = 0, hr = 0;      // host enqueue time per call (the last loop's): a loop whose GPU time equals it is bound by the host
        // (the tuning is per host thread AND per library instance: with A and B the same file -- one instance -- it is switched around every loop)
        for (int l = 0; l < loops; ++l) {
            if (tune_b && A.set_compact) A.set_compact(0);
            va.push_back(loop_us(calls, fa)); ha = g_host_us;
            if (tune_b && B.set_compact) B.set_compact(tune_b);
            vb.push_back(loop_us(calls, fb)); hb = g_host_us;
            vr.push_back(loop_us(calls, fr)); hr = g_host_us;
        }
        std::vector<V> ya(rows), yb(rows), yr(rows);
        CK(hipMemcpy(ya.data(), d_y[0], rows * sizeof(V), hipMemcpyDeviceToHost)); CK(hipMemcpy(yb.data(), d_y[1], rows * sizeof(V), hipMemcpyDeviceToHost));
        CK(hipMemcpy(yr.data(), d_y[2], rows * sizeof(V), hipMemcpyDeviceToHost));
        const bool same = memcmp(ya.data(), yb.data(), rows * sizeof(V)) == 0;
        double dmax = 0; int imax = 0;
        for (int i = 0; i < rows; ++i) { const double d = std::abs((double) ya[i] - (double) yr[i]); if (d > dmax) { dmax = d; imax = i; } }
        {
            std::vector<unsigned char> g(ta + 2 * GUARD); CK(hipMemcpy(g.data(), ga, g.size(), hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < GUARD; ++i) bad += (g[i] != 0xA5) + (g[GUARD + ta + i] != 0xA5);
            std::vector<unsigned char> h(tb + 2 * GUARD); CK(hipMemcpy(h.data(), gb, h.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < GUARD; ++i) bad += (h[i] != 0xA5) + (h[GUARD + tb + i] != 0xA5);
            if (bad) printf("# GUARD ZONES AROUND THE TEMP STORAGE TOUCHED: %zu bytes\n", bad);
            double ea = 0, er = 0; for (int i = 0; i < rows; ++i) { ea = std::max(ea, (double) std::abs(ya[i] - yh[i])); er = std::max(er, (double) std::abs(yr[i] - yh[i])); }
            if (ea > 1e-3 || er > 1e-3) printf("# against the sequential sum on the host: max |A - host| %.3g, max |R - host| %.3g\n", ea, er);
        }
        if (dmax > 1e-3) printf("# LARGE DIFFERENCE at row %d of %d: A %.6g  B %.6g  R %.6g\n", imax, rows, (double) ya[imax], (double) yb[imax], (double) yr[imax]);
        float ma, na, mb, nb, mr, nr; stats(va, ma, na); stats(vb, mb, nb); stats(vr, mr, nr);
        printf("grid2d_%-5d nnz %9d: A %.2f (%.2f)  B %.2f (%.2f)  R %.2f (%.2f)   host enqueue A %.2f B %.2f R %.2f   A == B bitwise: %s, max |A - R| %.2g\n", w, nnz, ma, na, mb, nb, mr, nr, ha, hb, hr, same ? "yes" : "NO", dmax);
        fflush(stdout);
        RK(rocsparse_destroy_mat_info(info)); RK(rocsparse_destroy_mat_descr(descr));
        if (tune_b && B.set_compact) B.set_compact(0);
        CK(hipFree(d_off)); CK(hipFree(d_col)); CK(hipFree(d_val)); CK(hipFree(d_x)); for (auto p : d_y) CK(hipFree(p)); CK(hipFree(ga)); CK(hipFree(gb));
    }
    RK(rocsparse_destroy_handle(handle));
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <libA.so> <libB.so> [--fp32] [--loops=9] [--calls=2000] [--tune-b=<n>] [sizes ...]\n", argv[0]); return 1; }
    bool f32 = false; int loops = 9, calls = 2000, tune_b = 0, ipt_b = 0; std::vector<int> sizes; const char *env_a = nullptr, *env_b = nullptr;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--fp32")) f32 = true;
        else if (!strncmp(argv[i], "--loops=", 8)) loops = atoi(argv[i] + 8);
        else if (!strncmp(argv[i], "--calls=", 8)) calls = atoi(argv[i] + 8);
        else if (!strncmp(argv[i], "--tune-b=", 9)) tune_b = atoi(argv[i] + 9);
        else if (!strncmp(argv[i], "--ipt-b=", 8)) ipt_b = atoi(argv[i] + 8);
        else if (!strncmp(argv[i], "--env-a=", 8)) env_a = argv[i] + 8;
        else if (!strncmp(argv[i], "--env-b=", 8)) env_b = argv[i] + 8;
        else sizes.push_back(atoi(argv[i]));
    }
    if (sizes.empty()) sizes = {30, 100, 300, 500, 600, 700, 800, 900, 1000, 1200, 2000};
    if (f32) run<float>(argv[1], argv[2], true, loops, calls, tune_b, ipt_b, sizes, env_a, env_b); else run<double>(argv[1], argv[2], false, loops, calls, tune_b, ipt_b, sizes, env_a, env_b);
    return 0;
}
