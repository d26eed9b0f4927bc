// cpu_bench.hpp -- how the product's OpenMP merge-path CsrMV is TIMED (bench.py's cpu_baseline, cpu_spmv --pin):
// the reference's CPU protocol (cpu_spmv.cpp:372-405) on private copies of the arrays that every thread
// first-touches for its own merge-path swath, threads optionally bound to distinct physical cores of socket 0
// (BASELINE config 1 says "single socket"; the reference gets there with numactl / KMP_AFFINITY outside the program).
#pragma once
#include <omp.h>
#include <sched.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

#include "merge_csrmv.hpp"

namespace mspmv_host {

// Physical cores of package 0 that this process may run on (one hardware thread each, in cpu-id order), from sysfs.
std::vector<int> Socket0Cores(int &packages)
{
    cpu_set_t allowed; CPU_ZERO(&allowed);
    std::vector<int> out; std::set<int> seen_pkg; std::set<long long> seen_core;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) { packages = 0; return out; }
    for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
        if (!CPU_ISSET(cpu, &allowed)) continue;
        char path[128]; int pkg = -1, core = -1;
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", cpu);
        if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &pkg) != 1) pkg = -1; fclose(f); }
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/core_id", cpu);
        if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &core) != 1) core = -1; fclose(f); }
        if (pkg < 0 || core < 0) continue;
        seen_pkg.insert(pkg);
        if (pkg == 0 && seen_core.insert(core).second) out.push_back(cpu);
    }
    packages = (int) seen_pkg.size();
    return out;
}

template <typename V> V *AllocUntouched(size_t n)
{
    void *p = mmap(nullptr, std::max<size_t>(n, 1) * sizeof(V) + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    return p == MAP_FAILED ? nullptr : static_cast<V *>(p);
}
template <typename V> void FreeUntouched(V *p, size_t n) { if (p) munmap(p, std::max<size_t>(n, 1) * sizeof(V) + 4096); }

// Time MergeCsrmv with the reference's CPU protocol (cpu_spmv.cpp:372-405: 1 run, 3 cache-warming runs, then
// the timed loop) on PRIVATE copies of the arrays that every thread first-touches for its own merge-path
// swath (so pages land on the NUMA node of the thread that streams them; x, gathered by everyone, is
// touched in equal slices).  pin != 0: threads are bound to distinct physical cores of socket 0.
template <typename V>
int BenchMerge(int threads, int pin, int rows, int cols, int nnz, const int *row_offsets, const int *col_idx, const V *vals,
               const V *x, double budget_s, int max_iters, double *avg_ms, int *iters_done, int *pinned, int *packages, V *y_out)
{
    if (threads < 1) return 1;
    int pk = 0; std::vector<int> cores = Socket0Cores(pk);
    if (packages) *packages = pk;
    const bool do_pin = pin && (int) cores.size() >= threads;
    if (do_pin) {
        // spread the team evenly over the socket's cores (core ids are grouped by CCD / L3 slice: 16 threads on a
        // 64-core EPYC take every 4th core = 2 per CCD, i.e. all of the socket's L3 and memory channels)
        std::vector<int> spread((size_t) threads);
        for (int t = 0; t < threads; ++t) spread[(size_t) t] = cores[(size_t) ((long long) t * (long long) cores.size() / threads)];
        cores.swap(spread);
    }
    if (pinned) *pinned = do_pin ? 1 : 0;
    int *ro = AllocUntouched<int>((size_t) rows + 1), *ci = AllocUntouched<int>((size_t) nnz);
    V *va = AllocUntouched<V>((size_t) nnz), *xx = AllocUntouched<V>((size_t) cols), *yy = AllocUntouched<V>((size_t) rows);
    if (!ro || !ci || !va || !xx || !yy) return 2;
    std::vector<cpu_set_t> old((size_t) threads);
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num();
        sched_getaffinity(0, sizeof(cpu_set_t), &old[t]);
        if (do_pin) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cores[t], &one); sched_setaffinity(0, sizeof(one), &one); }
        PathPoint b, e;
        SegmentBounds(t, threads, rows, nnz, row_offsets + 1, b, e);
        const int r0 = b.row, r1 = t == threads - 1 ? rows : e.row;
        for (int r = r0; r < r1; ++r) { ro[r + 1] = row_offsets[r + 1]; yy[r] = 0; }
        if (t == 0) ro[0] = row_offsets[0];
        for (int k = b.nz; k < e.nz; ++k) { ci[k] = col_idx[k]; va[k] = vals[k]; }
        const long long c0 = (long long) cols * t / threads, c1 = (long long) cols * (t + 1) / threads;
        for (long long c = c0; c < c1; ++c) xx[c] = x[c];
    }
    std::vector<int> cr; std::vector<V> cv;
    for (int w = 0; w < 4; ++w) MergeCsrmv<V>(threads, rows, nnz, ro + 1, ci, va, xx, yy, cr, cv);
    if (y_out) memcpy(y_out, yy, sizeof(V) * (size_t) rows);
    const double t0 = omp_get_wtime();
    int it = 0;
    while (it < max_iters && (it == 0 || omp_get_wtime() - t0 < budget_s)) { MergeCsrmv<V>(threads, rows, nnz, ro + 1, ci, va, xx, yy, cr, cv); ++it; }
    const double dt = omp_get_wtime() - t0;
    if (avg_ms) *avg_ms = dt * 1e3 / std::max(it, 1);
    if (iters_done) *iters_done = it;
#pragma omp parallel num_threads(threads)
    { sched_setaffinity(0, sizeof(cpu_set_t), &old[omp_get_thread_num()]); }
    FreeUntouched(ro, (size_t) rows + 1); FreeUntouched(ci, (size_t) nnz); FreeUntouched(va, (size_t) nnz);
    FreeUntouched(xx, (size_t) cols); FreeUntouched(yy, (size_t) rows);
    return 0;
}

}  // namespace mspmv_host
