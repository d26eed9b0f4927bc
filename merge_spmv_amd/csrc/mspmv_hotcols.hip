// mspmv_hotcols.hip -- the opt-in HOT-COLUMN PLAN of include/mspmv.h (mspmv_csrmv_hotcols_*): the columns renumbered,
// once, in order of how often the matrix references them, for matrices whose x is far larger than the caches.
//
// Why.  BASELINE config 5 (R-MAT scale 26: x = 512 MB, 2 * 10^9 references) runs at the DRAM random-line rate: 57 G
// gathers/s, 0.09 of the HBM roofline, whatever the kernel does (profiles/r02_hw_ceilings.txt: 54.6 G/s over tables
// >= 128 MiB).  But a scale-free matrix references its columns very unevenly -- at scale 26 the 524 288 most referenced
// columns (4 MB of x) take 62 % of the references and the 4 M most referenced (34 MB) 90 % -- and in the matrix as given
// those hot columns are scattered over all of x, each sharing its 128-byte line with cold ones (R-MAT: the 314 000
// columns with at most six set bits occupy 110 000 lines = 14 MB; packed they are 2.5 MB).  Renumbered by reference
// count, the hot columns are contiguous: the hottest few MB stay in every XCD's L2, the next few hundred in the
// Infinity Cache, and config 5 runs in 20.6 ms instead of 34.2 (tools/hot_columns.py, profiles/r03_hot_columns.txt).
//
// What.  A column permutation only changes WHERE x is read: the CSR arrays keep their order, so every row sums the same
// products in the same order and y is bit for bit the stateless call's.  The plan holds
//   new_cols[nnz]   the renumbered column indices (the caller's values and row offsets are used as they are: no copy),
//   order[cols]     order[k] = the original column that became column k,
//   xp[cols]        x in the new numbering, refreshed by one gather pass per SpMV (0.7 ms for config 5's 67 M columns:
//                   what a caller pays for keeping x in the original order),
// and the temp storage of the inner stateless call.  Hotness is by CLASS -- floor(log2(count + 1)), 32 classes,
// hottest first; inside a class the original order is kept block-wise -- which needs no sort: a histogram of the
// columns (one atomic per nonzero), a 32-entry scan, one ranking pass.  The reference's driver does the same kind of
// thing for its HYB column: conversion timed once as set-up, SpMV timed separately (gpu_spmv.cu:106-257).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/mspmv.h"
#include "mspmv_internal.hpp"

namespace {

using namespace mspmv;

constexpr int CLASSES = 33;                 // floor(log2(count + 1)) of a count < 2^32: 0 .. 32
constexpr int HC_BLOCK = 256;

struct HotLayout {
    uint64_t cols_off, order_off, xp_off, counts_off, class_off, temp_off, total;
    uint64_t temp_bytes;
};

bool make_layout(int rows, int cols, int nnz, int value_bytes, HotLayout &L)
{
    memset(&L, 0, sizeof(L));
    if ((long long) rows + nnz > MAX_ITEMS) return false;
    uint64_t off = 0;
    L.cols_off = off; off = align256(off + uint64_t(std::max(nnz, 1)) * 4);
    L.order_off = off; off = align256(off + uint64_t(std::max(cols, 1)) * 4);
    L.xp_off = off; off = align256(off + uint64_t(std::max(cols, 1)) * value_bytes);
    L.counts_off = off; off = align256(off + uint64_t(std::max(cols, 1)) * 4);          // reference counts, then the ranks
    L.class_off = off; off = align256(off + uint64_t(3 * CLASSES) * 4);                 // class sizes, bases, cursors
    L.temp_bytes = csrmv_temp_bytes(rows, nnz, value_bytes);
    L.temp_off = off; off = align256(off + L.temp_bytes);
    L.total = off;
    return true;
}

__device__ __forceinline__ int hot_class(unsigned count) { return count == 0xffffffffu ? 32 : 32 - __builtin_clz(count + 1u) - 0; }   // floor(log2(count + 1)) + 0

// counts[c] += 1 per reference.  Hot columns are hit by many lanes at once: a wave first folds equal neighbours
// (sorted rows put equal or near columns side by side only rarely, so this is plain atomics for most lanes).
__global__ __launch_bounds__(HC_BLOCK) void hot_count_kernel(const int *__restrict__ cols, long long nnz, unsigned *__restrict__ counts, int num_cols)
{
    const long long stride = (long long) gridDim.x * HC_BLOCK;
    for (long long j = (long long) blockIdx.x * HC_BLOCK + threadIdx.x; j < nnz; j += stride) {
        const unsigned c = (unsigned) cols[j];
        if (c < (unsigned) num_cols) atomicAdd(&counts[c], 1u);          // (an out-of-range index is left alone: the SpMV would fault on it either way)
    }
}

// class sizes: sizes[k] = number of columns of class k
__global__ __launch_bounds__(HC_BLOCK) void hot_class_sizes_kernel(const unsigned *__restrict__ counts, int num_cols, unsigned *__restrict__ sizes)
{
    __shared__ unsigned s_n[CLASSES];
    if (threadIdx.x < CLASSES) s_n[threadIdx.x] = 0u;
    __syncthreads();
    const int c = blockIdx.x * HC_BLOCK + threadIdx.x;
    if (c < num_cols) atomicAdd(&s_n[hot_class(counts[c])], 1u);
    __syncthreads();
    if (threadIdx.x < CLASSES && s_n[threadIdx.x]) atomicAdd(&sizes[threadIdx.x], s_n[threadIdx.x]);
}

// bases: hottest class first.  sizes / bases / cursors are three consecutive arrays of CLASSES words.
__global__ void hot_class_bases_kernel(unsigned *__restrict__ cls)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned run = 0;
    for (int k = CLASSES - 1; k >= 0; --k) { cls[CLASSES + k] = run; cls[2 * CLASSES + k] = 0u; run += cls[k]; }
}

// rank[c] = base[class] + position inside the class.  A block reserves, per class, one range for all its columns of
// that class (one global atomic per class and block) and numbers them inside it in thread order, so neighbouring
// original columns of one class stay neighbours.  counts[] is overwritten by the ranks; order[rank] = c.
__global__ __launch_bounds__(HC_BLOCK) void hot_rank_kernel(unsigned *__restrict__ counts_then_rank, int num_cols, unsigned *__restrict__ cls,
                                                            int *__restrict__ order)
{
    __shared__ unsigned s_n[CLASSES], s_base[CLASSES];
    __shared__ unsigned char s_cls[HC_BLOCK];
    if (threadIdx.x < CLASSES) s_n[threadIdx.x] = 0u;
    __syncthreads();
    const int c = blockIdx.x * HC_BLOCK + threadIdx.x;
    const int k = c < num_cols ? hot_class(counts_then_rank[c]) : -1;
    s_cls[threadIdx.x] = (unsigned char) (k < 0 ? 255 : k);
    if (k >= 0) atomicAdd(&s_n[k], 1u);
    __syncthreads();
    if (threadIdx.x < CLASSES && s_n[threadIdx.x])
        s_base[threadIdx.x] = cls[CLASSES + threadIdx.x] + atomicAdd(&cls[2 * CLASSES + threadIdx.x], s_n[threadIdx.x]);
    __syncthreads();
    if (k >= 0) {
        unsigned before = 0;                                            // columns of my class earlier in this block
        for (int t = 0; t < (int) threadIdx.x; ++t) before += s_cls[t] == (unsigned char) k ? 1u : 0u;
        const unsigned r = s_base[k] + before;
        counts_then_rank[c] = r;
        order[r] = c;
    }
}

__global__ __launch_bounds__(HC_BLOCK) void hot_relabel_kernel(const int *__restrict__ cols, long long nnz, const unsigned *__restrict__ rank,
                                                               int *__restrict__ new_cols, int num_cols)
{
    const long long stride = (long long) gridDim.x * HC_BLOCK;
    for (long long j = (long long) blockIdx.x * HC_BLOCK + threadIdx.x; j < nnz; j += stride) {
        const unsigned c = (unsigned) cols[j];
        new_cols[j] = c < (unsigned) num_cols ? (int) rank[c] : (int) c;
    }
}

// xp[k] = x[order[k]]: the one extra pass per SpMV (a gather in the ORIGINAL numbering; the writes are coalesced)
template <typename V>
__global__ __launch_bounds__(HC_BLOCK) void hot_permute_x_kernel(const V *__restrict__ x, const int *__restrict__ order, V *__restrict__ xp, int num_cols)
{
    const int k = blockIdx.x * HC_BLOCK + threadIdx.x;
    if (k < num_cols) xp[k] = x[order[k]];
}

#define HC_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int) e_; } while (0)

int launched(hipStream_t stream, int debug_sync, const char *name, unsigned grid)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int) e;
    if (debug_sync) { printf("mspmv: %s<<<%u, %d>>>\n", name, grid, HC_BLOCK); fflush(stdout); e = hipStreamSynchronize(stream); }
    return (int) e;
}

int hot_build(void *d_plan, size_t plan_bytes, const int32_t *d_row_offsets, const int32_t *d_cols, int32_t rows, int32_t cols, int32_t nnz,
              int32_t value_bytes, hipStream_t stream, int debug_sync)
{
    if (!d_plan || rows < 0 || cols < 0 || nnz < 0 || (value_bytes != 4 && value_bytes != 8) || !d_row_offsets || (nnz > 0 && !d_cols)) return hipErrorInvalidValue;
    HotLayout L;
    if (!make_layout(rows, cols, nnz, value_bytes, L) || plan_bytes < L.total || (reinterpret_cast<uintptr_t>(d_plan) & 15)) return hipErrorInvalidValue;
    char *base = static_cast<char *>(d_plan);
    unsigned *counts = reinterpret_cast<unsigned *>(base + L.counts_off);
    unsigned *cls = reinterpret_cast<unsigned *>(base + L.class_off);
    int *order = reinterpret_cast<int *>(base + L.order_off);
    int *new_cols = reinterpret_cast<int *>(base + L.cols_off);
    HC_HIP(hipMemsetAsync(counts, 0, (size_t) std::max(cols, 1) * 4, stream));
    HC_HIP(hipMemsetAsync(cls, 0, (size_t) 3 * CLASSES * 4, stream));
    if (cols == 0) return hipSuccess;
    const unsigned cgrid = (unsigned) ((cols + HC_BLOCK - 1) / HC_BLOCK);
    const unsigned ngrid = (unsigned) std::min<long long>(((long long) nnz + HC_BLOCK - 1) / HC_BLOCK, 1 << 16);
    if (nnz > 0) {
        hipLaunchKernelGGL(hot_count_kernel, dim3(ngrid), dim3(HC_BLOCK), 0, stream, d_cols, (long long) nnz, counts, cols);
        if (int e = launched(stream, debug_sync, "hot_count_kernel", ngrid)) return e;
    }
    hipLaunchKernelGGL(hot_class_sizes_kernel, dim3(cgrid), dim3(HC_BLOCK), 0, stream, counts, cols, cls);
    if (int e = launched(stream, debug_sync, "hot_class_sizes_kernel", cgrid)) return e;
    hipLaunchKernelGGL(hot_class_bases_kernel, dim3(1), dim3(64), 0, stream, cls);
    if (int e = launched(stream, debug_sync, "hot_class_bases_kernel", 1)) return e;
    hipLaunchKernelGGL(hot_rank_kernel, dim3(cgrid), dim3(HC_BLOCK), 0, stream, counts, cols, cls, order);
    if (int e = launched(stream, debug_sync, "hot_rank_kernel", cgrid)) return e;
    if (nnz > 0) {
        hipLaunchKernelGGL(hot_relabel_kernel, dim3(ngrid), dim3(HC_BLOCK), 0, stream, d_cols, (long long) nnz, counts, new_cols, cols);
        if (int e = launched(stream, debug_sync, "hot_relabel_kernel", ngrid)) return e;
    }
    // the hints of the inner call's tiles (they depend on the row offsets alone)
    CallExtra ex; ex.phase = PHASE_COORDS_ONLY; ex.no_bands = true;
    size_t tb = (size_t) L.temp_bytes;
    if (value_bytes == 4)
        return csrmv_call<float>(base + L.temp_off, &tb, nullptr, d_row_offsets, nullptr, nullptr, nullptr, rows, 0, nnz, 1.f, 0.f, false, stream, debug_sync, ex);
    return csrmv_call<double>(base + L.temp_off, &tb, nullptr, d_row_offsets, nullptr, nullptr, nullptr, rows, 0, nnz, 1.0, 0.0, false, stream, debug_sync, ex);
}

// x -> the plan's numbering: out[k] = x[order[k]] (what hot_apply does per SpMV into the plan's own buffer; exported for callers that keep
// x in the plan's numbering themselves: mspmv_csrmv_hotcols_permute_* / _apply_permuted_*)
template <typename V>
int hot_permute(const void *d_plan, size_t plan_bytes, const V *d_x, V *d_out, int32_t rows, int32_t cols, int32_t nnz, hipStream_t stream, int debug_sync)
{
    if (!d_plan || rows < 0 || cols < 0 || nnz < 0) return hipErrorInvalidValue;
    HotLayout L;
    if (!make_layout(rows, cols, nnz, (int) sizeof(V), L) || plan_bytes < L.total || (reinterpret_cast<uintptr_t>(d_plan) & 15)) return hipErrorInvalidValue;
    if (cols == 0) return hipSuccess;
    if (!d_x || !d_out || d_x == d_out) return hipErrorInvalidValue;
    const unsigned grid = (unsigned) ((cols + HC_BLOCK - 1) / HC_BLOCK);
    hipLaunchKernelGGL((hot_permute_x_kernel<V>), dim3(grid), dim3(HC_BLOCK), 0, stream, d_x, reinterpret_cast<const int *>(static_cast<const char *>(d_plan) + L.order_off), d_out, cols);
    return launched(stream, debug_sync, "hot_permute_x_kernel", grid);
}

template <typename V>
int hot_apply(void *d_plan, size_t plan_bytes, const V *d_values, const int32_t *d_row_offsets, const V *d_x, V *d_y, int32_t rows, int32_t cols,
              int32_t nnz, V alpha, V beta, hipStream_t stream, int debug_sync, bool x_is_permuted = false)
{
    if (!d_plan || rows < 0 || cols < 0 || nnz < 0) return hipErrorInvalidValue;
    HotLayout L;
    if (!make_layout(rows, cols, nnz, (int) sizeof(V), L) || plan_bytes < L.total || (reinterpret_cast<uintptr_t>(d_plan) & 15)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!d_y || !d_row_offsets || (nnz > 0 && (!d_x || !d_values))) return hipErrorInvalidValue;
    char *base = static_cast<char *>(d_plan);
    const V *xp = x_is_permuted ? d_x : reinterpret_cast<const V *>(base + L.xp_off);
    if (cols > 0 && nnz > 0 && !x_is_permuted) {
        const unsigned grid = (unsigned) ((cols + HC_BLOCK - 1) / HC_BLOCK);
        hipLaunchKernelGGL((hot_permute_x_kernel<V>), dim3(grid), dim3(HC_BLOCK), 0, stream, d_x, reinterpret_cast<const int *>(base + L.order_off), reinterpret_cast<V *>(base + L.xp_off), cols);
        if (int e = launched(stream, debug_sync, "hot_permute_x_kernel", grid)) return e;
    }
    CallExtra ex; ex.phase = PHASE_ALL;            // (hints from the build; verified by the tiles as always)
    ex.no_bands = true;                            // the hot columns are contiguous now: column bands of x have nothing to add, and their verdict is
                                                   // sampled from the column indices -- which the renumbering changed -- so taking them would make the
                                                   // plan's rounding differ from the one-sweep stateless call's
    size_t tb = (size_t) L.temp_bytes;
    return csrmv_call<V>(base + L.temp_off, &tb, d_values, d_row_offsets, reinterpret_cast<const int *>(base + L.cols_off), xp, d_y, rows, cols, nnz,
                         alpha, beta, !(alpha == (V) 1 && beta == (V) 0), stream, debug_sync, ex);
}

// ---- would the hot-column plan pay?  A cheap on-device look at the column indices: SKEW_WINDOWS windows of SKEW_WINDOW consecutive
// nonzeros spread over the matrix (~1 M references), every referenced 128-byte line of x marked in ONE bitmap over all lines of x;
// the number of DISTINCT lines the sample touches, against what the same number of uniformly drawn references would touch
// (L (1 - exp(-n / L)) of L lines).  Uniformly spread columns: ~1.0 (nothing to concentrate); a scale-free matrix: 0.4-0.7 -- part of
// the references keeps coming back to the same lines, the hot ones the plan packs together (BASELINE config 5, R-MAT scale 26: 62 % of
// the references go to 4 MB of x); stencils and bands: a few percent (their gathers hit the caches as they are).  Also: how many
// windows span most of x (a band does not).  (A first version counted distinct lines per WINDOW of 2048 nonzeros: short-range reuse,
// which a hot set of 4 MB does not show -- config 5 looked uniform to it.)
constexpr int SKEW_WINDOWS = 512, SKEW_WINDOW = 2048;
__global__ __launch_bounds__(HC_BLOCK) void hot_skew_kernel(const int *__restrict__ cols, int nnz, int num_cols, int line_shift, unsigned *__restrict__ bitmap,
                                                           unsigned *__restrict__ out /* [0] distinct lines, [1] wide windows, [2] samples */)
{
    __shared__ int s_red[3][HC_BLOCK / 64];
    const int tid = threadIdx.x, w = blockIdx.x;
    const long long start = nnz > SKEW_WINDOW ? (long long) w * (nnz - SKEW_WINDOW) / (SKEW_WINDOWS - 1) : 0;
    const int len = nnz < SKEW_WINDOW ? nnz : SKEW_WINDOW;
    int fresh = 0, lo = 0x7fffffff, hi = -1;
    for (int j = tid; j < len; j += HC_BLOCK) {
        const int c = cols[start + j];
        lo = c < lo ? c : lo; hi = c > hi ? c : hi;
        const unsigned line = (unsigned) c >> line_shift;
        const unsigned bit = 1u << (line & 31u);
        fresh += (atomicOr(&bitmap[line >> 5], bit) & bit) ? 0 : 1;
    }
    for (int d = 32; d > 0; d >>= 1) {
        fresh += __shfl_xor(fresh, d, 64);
        const int l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((tid & 63) == 0) { s_red[0][tid / 64] = fresh; s_red[1][tid / 64] = lo; s_red[2][tid / 64] = hi; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < HC_BLOCK / 64; ++k) { fresh += s_red[0][k]; lo = s_red[1][k] < lo ? s_red[1][k] : lo; hi = s_red[2][k] > hi ? s_red[2][k] : hi; }
        atomicAdd(&out[0], (unsigned) fresh);
        if (4LL * ((long long) hi - lo) >= 3LL * num_cols) atomicAdd(&out[1], 1u);
        atomicAdd(&out[2], (unsigned) len);
    }
}

}  // namespace

extern "C" {

int mspmv_csrmv_hotcols_size(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, size_t *plan_bytes)
{
    if (!plan_bytes || rows < 0 || cols < 0 || nnz < 0 || (value_bytes != 4 && value_bytes != 8)) return hipErrorInvalidValue;
    HotLayout L;
    if (!make_layout(rows, cols, nnz, value_bytes, L)) return hipErrorInvalidValue;
    *plan_bytes = (size_t) L.total;
    return hipSuccess;
}

int mspmv_csrmv_hotcols_build(void *d_plan, size_t plan_bytes, const int32_t *d_row_offsets, const int32_t *d_column_indices, int32_t rows,
                              int32_t cols, int32_t nnz, int32_t value_bytes, mspmv_stream_t stream, int debug_sync)
{
    return hot_build(d_plan, plan_bytes, d_row_offsets, d_column_indices, rows, cols, nnz, value_bytes, reinterpret_cast<hipStream_t>(stream), debug_sync);
}

int mspmv_csrmv_hotcols_apply_f32(void *d_plan, size_t plan_bytes, const float *d_values, const int32_t *d_row_offsets, const float *d_x, float *d_y,
                                  int32_t rows, int32_t cols, int32_t nnz, float alpha, float beta, mspmv_stream_t stream, int debug_sync)
{
    return hot_apply<float>(d_plan, plan_bytes, d_values, d_row_offsets, d_x, d_y, rows, cols, nnz, alpha, beta, reinterpret_cast<hipStream_t>(stream), debug_sync);
}

int mspmv_csrmv_hotcols_apply_f64(void *d_plan, size_t plan_bytes, const double *d_values, const int32_t *d_row_offsets, const double *d_x, double *d_y,
                                  int32_t rows, int32_t cols, int32_t nnz, double alpha, double beta, mspmv_stream_t stream, int debug_sync)
{
    return hot_apply<double>(d_plan, plan_bytes, d_values, d_row_offsets, d_x, d_y, rows, cols, nnz, alpha, beta, reinterpret_cast<hipStream_t>(stream), debug_sync);
}

/* x kept in the plan's numbering by the caller: permute once (d_x_permuted[k] = d_x[order[k]]), then SpMVs without the per-call pass */
int mspmv_csrmv_hotcols_permute_f32(const void *d_plan, size_t plan_bytes, const float *d_x, float *d_x_permuted, int32_t rows, int32_t cols, int32_t nnz,
                                    mspmv_stream_t stream, int debug_sync)
{
    return hot_permute<float>(d_plan, plan_bytes, d_x, d_x_permuted, rows, cols, nnz, reinterpret_cast<hipStream_t>(stream), debug_sync);
}
int mspmv_csrmv_hotcols_permute_f64(const void *d_plan, size_t plan_bytes, const double *d_x, double *d_x_permuted, int32_t rows, int32_t cols, int32_t nnz,
                                    mspmv_stream_t stream, int debug_sync)
{
    return hot_permute<double>(d_plan, plan_bytes, d_x, d_x_permuted, rows, cols, nnz, reinterpret_cast<hipStream_t>(stream), debug_sync);
}
int mspmv_csrmv_hotcols_apply_permuted_f32(void *d_plan, size_t plan_bytes, const float *d_values, const int32_t *d_row_offsets, const float *d_x_permuted,
                                           float *d_y, int32_t rows, int32_t cols, int32_t nnz, float alpha, float beta, mspmv_stream_t stream, int debug_sync)
{
    return hot_apply<float>(d_plan, plan_bytes, d_values, d_row_offsets, d_x_permuted, d_y, rows, cols, nnz, alpha, beta, reinterpret_cast<hipStream_t>(stream), debug_sync, true);
}
int mspmv_csrmv_hotcols_apply_permuted_f64(void *d_plan, size_t plan_bytes, const double *d_values, const int32_t *d_row_offsets, const double *d_x_permuted,
                                           double *d_y, int32_t rows, int32_t cols, int32_t nnz, double alpha, double beta, mspmv_stream_t stream, int debug_sync)
{
    return hot_apply<double>(d_plan, plan_bytes, d_values, d_row_offsets, d_x_permuted, d_y, rows, cols, nnz, alpha, beta, reinterpret_cast<hipStream_t>(stream), debug_sync, true);
}

/* the plan's pieces, for callers that keep x in the plan's numbering themselves (an iterative method on a symmetric permutation)
 * and for the tests: order[k] = the original column that became column k (cols int32), the renumbered column indices (nnz int32) */
const int32_t *mspmv_csrmv_hotcols_order(const void *d_plan, int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes)
{
    HotLayout L;
    if (!d_plan || !make_layout(rows, cols, nnz, value_bytes, L)) return nullptr;
    return reinterpret_cast<const int32_t *>(static_cast<const char *>(d_plan) + L.order_off);
}
const int32_t *mspmv_csrmv_hotcols_columns(const void *d_plan, int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes)
{
    HotLayout L;
    if (!d_plan || !make_layout(rows, cols, nnz, value_bytes, L)) return nullptr;
    return reinterpret_cast<const int32_t *>(static_cast<const char *>(d_plan) + L.cols_off);
}


int mspmv_csrmv_hotcols_skew(const int32_t *d_column_indices, int32_t cols, int32_t nnz, int32_t value_bytes, mspmv_stream_t stream_,
                             int32_t *distinct_permille_of_uniform, int32_t *wide_windows)
{
    if (!d_column_indices || cols < 0 || nnz < 0 || (value_bytes != 4 && value_bytes != 8) || !distinct_permille_of_uniform) return hipErrorInvalidValue;
    *distinct_permille_of_uniform = -1;
    if (wide_windows) *wide_windows = 0;
    if (nnz < SKEW_WINDOW || cols < 1) return hipSuccess;            // too small to say (and far too small to need the plan)
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int line_shift = value_bytes == 4 ? 5 : 4;
    const long long lines = (((long long) cols - 1) >> line_shift) + 1;
    const size_t words = (size_t) ((lines + 31) / 32) + 4;            // bitmap, then the three result words
    unsigned *d_map = nullptr;
    hipError_t e = hipMalloc(&d_map, words * sizeof(unsigned));
    if (e != hipSuccess) return (int) e;
    e = hipMemsetAsync(d_map, 0, words * sizeof(unsigned), stream);
    unsigned h[3] = {0, 0, 0};
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hot_skew_kernel, dim3(SKEW_WINDOWS), dim3(HC_BLOCK), 0, stream, d_column_indices, nnz, cols, line_shift, d_map, d_map + words - 4);
        e = hipMemcpyAsync(h, d_map + words - 4, sizeof(h), hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void) hipFree(d_map);
    if (e != hipSuccess) return (int) e;
    // what h[2] uniformly drawn references would touch of `lines` lines (windows may overlap on a small matrix: an upper bound then)
    const double expect = (double) lines * (1.0 - exp(-(double) h[2] / (double) lines));
    *distinct_permille_of_uniform = expect > 0 ? (int32_t) (1000.0 * (double) h[0] / expect + 0.5) : -1;
    if (wide_windows) *wide_windows = (int32_t) h[1];
    return hipSuccess;
}

}  // extern "C"
