// mspmv_plan.hip -- the opt-in PREPARED PLAN of include/mspmv.h (mspmv_csrmv_plan_*): a band-major copy
// of the matrix, made once, that turns a gather-bound CsrMV into a cache-resident one.
//
// Why.  When x is larger than an XCD's 4 MiB L2 (BASELINE config 2: 12.5 MB), every XCD gathers from all
// of x, ~70 % of the 4-byte gathers miss and each miss moves a 128-byte line over the fabric: 11x the
// algorithmic bytes, and that -- not HBM -- bounds the stateless call.  The reference's
// interface has no analysis phase, so mspmv_csrmv_* can do nothing about it; a caller who multiplies by
// the same matrix many times can.  The reference does the same thing for its HYB comparison: conversion
// timed once as "setup", SpMV timed separately (gpu_spmv.cu:106-257).
//
// What.  The columns are cut into B equal bands (the fewest -- 2, 4, then multiples of the 8 XCDs -- whose
// slices of x stay <= 3.25 MiB); the matrix is re-laid out as the STACKED matrix A' = [A_0; A_1; ...; A_{B-1}]
// (A_b = the entries of A whose column lies in band b; B*rows rows, the same nnz, absolute column indices),
// which is an ordinary CSR matrix.  y' = A' x is computed by the UNCHANGED merge-path CsrMV (csrmv_call)
// with ONE difference in the launch: the block -> tile mapping gives every XCD one contiguous range of
// tiles, i.e. a range of bands, so at any time an XCD gathers from one band's slice of x, which stays in its
// L2; the CSR stream is read exactly once.  A last kernel folds the bands: y[r] = alpha * sum_b y'[b*rows + r]
// + beta * y[r], in band order (deterministic).  Merge-path tiles keep the load balanced whatever the
// distribution of the nonzeros over bands and rows.
//
// Cost.  Storage = a second copy of the matrix + (B*rows + 1) offsets + B*rows partial sums (caller-owned,
// sized by mspmv_csrmv_plan_size); build = three passes on the device (count, scan, scatter).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "../../include/mspmv.h"
#include "mspmv_internal.hpp"

namespace {

using namespace mspmv;

constexpr int CHUNK = 1024;                 // nonzeros per block of the count / scatter passes
constexpr int SCAN_BLOCK = 256, SCAN_IPT = 16, SCAN_CHUNK = SCAN_BLOCK * SCAN_IPT;
constexpr uint32_t PLAN_MAGIC = 0x4d53504c; // "MSPL"

struct PlanHeader {                         // first 256 bytes of the caller's storage (device memory)
    uint32_t magic; int32_t value_bytes, rows, cols, nnz, bands, band_width, unsorted;
};

struct PlanLayout {
    int bands, band_width;
    long long srows;                        // rows of the stacked matrix
    uint64_t header_off, offsets_off, cols_off, values_off, ypart_off, temp_off, bsum_off, total;
    uint64_t temp_bytes;
};

// bands: 0 = automatic.  A band's slice of x must sit in an XCD's 4 MiB L2 beside the (non-temporal) streams
// that pass through it; every extra band costs `rows` stacked rows and partial sums.  Measured on C2
// (profiles/r02_plan.txt): fp32, 12.5 MB of x: 8 bands (1.6 MB slices) 0.527 ms, 16: 0.574, 32: 0.729;
// 4 bands (3.1 MB slices, two XCDs per band) 0.511; fp64, 25 MB: 8 bands (3.1 MB slices) 0.667 ms, 16 (1.6 MB): 0.734,
// 4 (6.25 MB): 0.850.  So: the fewest bands -- 2, 4, then multiples of the 8 XCDs -- whose slices stay <= 3.25 MiB;
// 1 when x fits anyway.  (With one contiguous tile range per XCD any band count lines up with the XCDs.)
int pick_bands(long long cols, int value_bytes, int bands)
{
    if (bands > 0) return bands;
    const long long xbytes = cols * value_bytes;
    if (xbytes <= (3ll << 20)) return 1;
    const long long limit = 13ll << 18;                  // 3.25 MiB
    for (int b : {2, 4}) if ((xbytes + b - 1) / b <= limit) return b;
    int b = 8;
    while (b < 64 && (xbytes + b - 1) / b > limit) b += 8;
    return b;
}

bool make_layout(int rows, int cols, int nnz, int value_bytes, int bands, PlanLayout &L)
{
    memset(&L, 0, sizeof(L));
    L.bands = pick_bands(cols, value_bytes, bands);
    if (L.bands < 1 || L.bands > 64) return false;
    L.band_width = (int) (((long long) cols + L.bands - 1) / L.bands);
    if (L.band_width < 1) L.band_width = 1;
    L.srows = (long long) L.bands * rows;
    if (L.srows + nnz > MAX_ITEMS) return false;
    uint64_t off = 0;
    L.header_off = off; off = align256(off + sizeof(PlanHeader));
    L.offsets_off = off; off = align256(off + uint64_t(L.srows + 1) * 4);
    L.cols_off = off; off = align256(off + uint64_t(std::max(nnz, 1)) * 4);
    L.values_off = off; off = align256(off + uint64_t(std::max(nnz, 1)) * value_bytes);
    L.ypart_off = off; off = align256(off + uint64_t(std::max<long long>(L.srows, 1)) * std::max(value_bytes, 4));   // also the build's cursors
    L.temp_bytes = csrmv_temp_bytes((int32_t) L.srows, nnz, value_bytes);
    L.temp_off = off; off = align256(off + L.temp_bytes);
    const uint64_t scan_blocks = (uint64_t) (L.srows + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
    L.bsum_off = off; off = align256(off + scan_blocks * 4);
    L.total = off;
    return true;
}

// the row holding nonzero j: the largest r in [lo, hi] with row_offsets[r] <= j
__device__ __forceinline__ int row_of(const int *__restrict__ off, int lo, int hi, int j)
{
    while (lo < hi) {
        const int mid = (int) (((long long) lo + hi + 1) >> 1);
        if (off[mid] <= j) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// block-wide: the rows of the block's first and last nonzero (two waves search concurrently)
__device__ __forceinline__ void block_row_range(const int *__restrict__ off, int rows, int j0, int j1, int *s_range)
{
    if (threadIdx.x == 0) s_range[0] = row_of(off, 0, rows - 1, j0);
    if (threadIdx.x == 64) s_range[1] = row_of(off, 0, rows - 1, j1);
    __syncthreads();
}

// pass 1: counts[band * rows + row] += 1 for every nonzero; notes whether some row's columns are not sorted
__global__ __launch_bounds__(256) void plan_count_kernel(const int *__restrict__ off, const int *__restrict__ cols, int rows, int nnz,
                                                         int band_width, int *__restrict__ counts, PlanHeader *hdr)
{
    __shared__ int s_range[2];
    const int j0 = blockIdx.x * CHUNK;
    const int j1 = min(j0 + CHUNK, nnz) - 1;
    block_row_range(off, rows, j0, j1, s_range);
    const int r_lo = s_range[0], r_hi = s_range[1];
    bool unsorted = false;
    for (int j = j0 + (int) threadIdx.x; j <= j1; j += 256) {
        const int r = row_of(off, r_lo, r_hi, j);
        const int c = cols[j];
        if (j > off[r] && cols[j - 1] > c) unsorted = true;
        atomicAdd(&counts[(long long) (c / band_width) * rows + r], 1);
    }
    if (unsorted) hdr->unsorted = 1;
}

// ---- exclusive scan of n ints: out[0] = 0, out[i + 1] = in[0] + ... + in[i] ------------------------------------
__device__ __forceinline__ int block_inclusive_scan(int v, int *s_tmp)      // 256 threads
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(v, d, 64); if (lane >= d) v += u; }
    if (lane == 63) s_tmp[wave] = v;
    __syncthreads();
    int add = 0;
    for (int w = 0; w < wave; ++w) add += s_tmp[w];
    __syncthreads();
    return v + add;
}
__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(const int *__restrict__ in, long long n, int *__restrict__ bsum)
{
    __shared__ int s_tmp[4];
    const long long base = (long long) blockIdx.x * SCAN_CHUNK + (long long) threadIdx.x * SCAN_IPT;
    int t = 0;
    for (int k = 0; k < SCAN_IPT; ++k) if (base + k < n) t += in[base + k];
    const int incl = block_inclusive_scan(t, s_tmp);
    if (threadIdx.x == SCAN_BLOCK - 1) bsum[blockIdx.x] = incl;
}
__global__ __launch_bounds__(SCAN_BLOCK) void scan_blocksums_kernel(int *__restrict__ bsum, int nblocks)
{
    __shared__ int s_tmp[4];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += SCAN_BLOCK) {
        const int i = base + (int) threadIdx.x;
        const int v = i < nblocks ? bsum[i] : 0;
        const int incl = block_inclusive_scan(v, s_tmp);
        const int carry = s_carry;
        if (i < nblocks) bsum[i] = carry + incl - v;          // exclusive
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) s_carry = carry + incl;
        __syncthreads();
    }
}
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const int *__restrict__ in, long long n, const int *__restrict__ bsum,
                                                                int *__restrict__ out)
{
    __shared__ int s_tmp[4];
    const long long base = (long long) blockIdx.x * SCAN_CHUNK + (long long) threadIdx.x * SCAN_IPT;
    int v[SCAN_IPT]; int t = 0;
    for (int k = 0; k < SCAN_IPT; ++k) { v[k] = base + k < n ? in[base + k] : 0; t += v[k]; }
    int run = block_inclusive_scan(t, s_tmp) - t + bsum[blockIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
    for (int k = 0; k < SCAN_IPT; ++k) { run += v[k]; if (base + k < n) out[base + k + 1] = run; }
}

// pass 3: every nonzero goes to its (band, row) group of the stacked matrix.  Sorted rows (the reference's
// CSR, sparse_matrix.h:666-728): the position inside the group is the distance to the group's first entry,
// found by a search in the row -- order preserving and deterministic.  Unsorted rows: a cursor per group.
template <typename V>
__global__ __launch_bounds__(256) void plan_scatter_kernel(const int *__restrict__ off, const int *__restrict__ cols,
                                                           const V *__restrict__ vals, int rows, int nnz, int band_width,
                                                           const int *__restrict__ soff, int *__restrict__ cursors,
                                                           const PlanHeader *hdr, int *__restrict__ cols_out, V *__restrict__ vals_out)
{
    __shared__ int s_range[2];
    const int j0 = blockIdx.x * CHUNK;
    const int j1 = min(j0 + CHUNK, nnz) - 1;
    block_row_range(off, rows, j0, j1, s_range);
    const int r_lo = s_range[0], r_hi = s_range[1];
    const bool unsorted = hdr->unsorted != 0;
    for (int j = j0 + (int) threadIdx.x; j <= j1; j += 256) {
        const int r = row_of(off, r_lo, r_hi, j);
        const int c = cols[j];
        const int b = c / band_width;
        const long long group = (long long) b * rows + r;
        int pos;
        if (unsorted) pos = soff[group] + atomicAdd(&cursors[group], 1);
        else {
            const int key = b * band_width;              // first index in [row start, j] whose column is >= key
            int lo = off[r], hi = j;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cols[mid] < key) lo = mid + 1; else hi = mid; }
            pos = soff[group] + (j - lo);
        }
        cols_out[pos] = c;
        vals_out[pos] = vals[j];
    }
}

// y[r] = alpha * (y'[r] + y'[rows + r] + ...) + beta * y[r], bands added in order.  VEC: a thread folds one 16-byte unit
// of rows per band (rows a multiple of the unit and y 16-byte aligned, so every band's row block is aligned too).
template <typename V, bool VEC>
__global__ __launch_bounds__(256) void plan_combine_kernel(const V *__restrict__ ypart, V *__restrict__ y, int rows, int bands, V alpha,
                                                           V beta)
{
    constexpr int W = VEC ? 16 / (int) sizeof(V) : 1;
    typedef V vec_t __attribute__((ext_vector_type(W)));
    const long long r = ((long long) blockIdx.x * 256 + threadIdx.x) * W;
    if (r >= rows) return;
    if constexpr (VEC) {
        vec_t s = *reinterpret_cast<const vec_t *>(ypart + r);
        for (int b = 1; b < bands; ++b) s += __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(ypart + (size_t) b * rows + r));
        vec_t *out = reinterpret_cast<vec_t *>(y + r);
        *out = beta == (V) 0 ? alpha * s : alpha * s + beta * *out;
    } else {
        V s = ypart[r];
        for (int b = 1; b < bands; ++b) s += __builtin_nontemporal_load(ypart + (size_t) b * rows + r);
        y[r] = beta == (V) 0 ? alpha * s : alpha * s + beta * y[r];
    }
}

#define PL_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int) e_; } while (0)

static int launched(hipStream_t stream, int debug_sync, const char *name, unsigned grid)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int) e;
    if (debug_sync) { printf("mspmv: %s<<<%u, 256>>>\n", name, grid); fflush(stdout); e = hipStreamSynchronize(stream); }
    return (int) e;
}

template <typename V>
int plan_build(void *d_plan, size_t plan_bytes, const V *d_values, const int32_t *d_row_offsets, const int32_t *d_cols, int32_t rows,
               int32_t cols, int32_t nnz, int32_t bands, hipStream_t stream, int debug_sync)
{
    if (!d_plan || rows < 0 || cols < 0 || nnz < 0 || !d_row_offsets || (nnz > 0 && (!d_values || !d_cols))) return hipErrorInvalidValue;
    PlanLayout L;
    if (!make_layout(rows, cols, nnz, (int) sizeof(V), bands, L) || plan_bytes < L.total || (reinterpret_cast<uintptr_t>(d_plan) & 15)) return hipErrorInvalidValue;
    char *base = static_cast<char *>(d_plan);
    PlanHeader *hdr = reinterpret_cast<PlanHeader *>(base + L.header_off);
    int *soff = reinterpret_cast<int *>(base + L.offsets_off);
    int *cols_out = reinterpret_cast<int *>(base + L.cols_off);
    V *vals_out = reinterpret_cast<V *>(base + L.values_off);
    int *counts = reinterpret_cast<int *>(base + L.ypart_off);
    int *bsum = reinterpret_cast<int *>(base + L.bsum_off);
    PlanHeader h; h.magic = PLAN_MAGIC; h.value_bytes = (int) sizeof(V); h.rows = rows; h.cols = cols; h.nnz = nnz;
    h.bands = L.bands; h.band_width = L.band_width; h.unsorted = 0;
    PL_HIP(hipMemcpyAsync(hdr, &h, sizeof(h), hipMemcpyHostToDevice, stream));
    PL_HIP(hipMemsetAsync(counts, 0, (size_t) std::max<long long>(L.srows, 1) * 4, stream));
    PL_HIP(hipMemsetAsync(soff, 0, 4, stream));
    if (rows == 0) return hipSuccess;
    const unsigned nz_grid = (unsigned) ((nnz + CHUNK - 1) / CHUNK);
    if (nnz > 0) {
        hipLaunchKernelGGL(plan_count_kernel, dim3(nz_grid), dim3(256), 0, stream, d_row_offsets, d_cols, rows, nnz, L.band_width, counts, hdr);
        if (int e = launched(stream, debug_sync, "plan_count_kernel", nz_grid)) return e;
    }
    const unsigned sblocks = (unsigned) ((L.srows + SCAN_CHUNK - 1) / SCAN_CHUNK);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(sblocks), dim3(SCAN_BLOCK), 0, stream, counts, L.srows, bsum);
    if (int e = launched(stream, debug_sync, "scan_reduce_kernel", sblocks)) return e;
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, stream, bsum, (int) sblocks);
    if (int e = launched(stream, debug_sync, "scan_blocksums_kernel", 1)) return e;
    hipLaunchKernelGGL(scan_apply_kernel, dim3(sblocks), dim3(SCAN_BLOCK), 0, stream, counts, L.srows, bsum, soff);
    if (int e = launched(stream, debug_sync, "scan_apply_kernel", sblocks)) return e;
    if (nnz > 0) {
        PL_HIP(hipMemsetAsync(counts, 0, (size_t) L.srows * 4, stream));       // now the cursors of the unsorted path
        hipLaunchKernelGGL((plan_scatter_kernel<V>), dim3(nz_grid), dim3(256), 0, stream, d_row_offsets, d_cols, d_values, rows, nnz,
                           L.band_width, soff, counts, hdr, cols_out, vals_out);
        if (int e = launched(stream, debug_sync, "plan_scatter_kernel", nz_grid)) return e;
    }
    // tile coordinates of the stacked matrix, once
    CallExtra ex; ex.phase = PHASE_COORDS_ONLY;
    size_t tb = (size_t) L.temp_bytes;
    return csrmv_call<V>(base + L.temp_off, &tb, nullptr, soff, nullptr, nullptr, nullptr, (int32_t) L.srows, 0, nnz, (V) 1, (V) 0, false,
                         stream, debug_sync, ex);
}

template <typename V>
int plan_apply(void *d_plan, size_t plan_bytes, const V *d_x, V *d_y, int32_t rows, int32_t cols, int32_t nnz, int32_t bands, V alpha,
               V beta, hipStream_t stream, int debug_sync)
{
    if (!d_plan || rows < 0 || cols < 0 || nnz < 0) return hipErrorInvalidValue;
    PlanLayout L;
    if (!make_layout(rows, cols, nnz, (int) sizeof(V), bands, L) || plan_bytes < L.total || (reinterpret_cast<uintptr_t>(d_plan) & 15)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    if (!d_y || (nnz > 0 && !d_x)) return hipErrorInvalidValue;
    char *base = static_cast<char *>(d_plan);
    V *ypart = reinterpret_cast<V *>(base + L.ypart_off);
    CallExtra ex; ex.phase = PHASE_SKIP_COORDS; ex.tile_map = L.bands > 1 ? TILE_MAP_CONTIGUOUS_CODE : 0;
    size_t tb = (size_t) L.temp_bytes;
    // with one band the stacked matrix IS the matrix: write y directly
    if (L.bands == 1)
        return csrmv_call<V>(base + L.temp_off, &tb, reinterpret_cast<const V *>(base + L.values_off),
                             reinterpret_cast<const int *>(base + L.offsets_off), reinterpret_cast<const int *>(base + L.cols_off), d_x,
                             d_y, rows, cols, nnz, alpha, beta, !(alpha == (V) 1 && beta == (V) 0), stream, debug_sync, ex);
    const int st = csrmv_call<V>(base + L.temp_off, &tb, reinterpret_cast<const V *>(base + L.values_off),
                                 reinterpret_cast<const int *>(base + L.offsets_off), reinterpret_cast<const int *>(base + L.cols_off),
                                 d_x, ypart, (int32_t) L.srows, cols, nnz, (V) 1, (V) 0, false, stream, debug_sync, ex);
    if (st != 0) return st;
    constexpr int W = 16 / (int) sizeof(V);
    const bool vec = rows % W == 0 && (reinterpret_cast<uintptr_t>(d_y) & 15) == 0;
    const long long units = vec ? rows / W : rows;
    const unsigned grid = (unsigned) ((units + 255) / 256);
    if (vec) hipLaunchKernelGGL((plan_combine_kernel<V, true>), dim3(grid), dim3(256), 0, stream, ypart, d_y, rows, L.bands, alpha, beta);
    else hipLaunchKernelGGL((plan_combine_kernel<V, false>), dim3(grid), dim3(256), 0, stream, ypart, d_y, rows, L.bands, alpha, beta);
    return launched(stream, debug_sync, "plan_combine_kernel", grid);
}

}  // namespace

extern "C" {

int mspmv_csrmv_plan_size(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, int32_t bands, size_t *plan_bytes,
                          int32_t *bands_used)
{
    if (!plan_bytes || rows < 0 || cols < 0 || nnz < 0 || (value_bytes != 4 && value_bytes != 8) || bands < 0) return hipErrorInvalidValue;
    PlanLayout L;
    if (!make_layout(rows, cols, nnz, value_bytes, bands, L)) return hipErrorInvalidValue;
    *plan_bytes = (size_t) L.total;
    if (bands_used) *bands_used = L.bands;
    return hipSuccess;
}

int mspmv_csrmv_plan_build_f32(void *d_plan, size_t plan_bytes, const float *d_values, const int32_t *d_row_offsets,
                               const int32_t *d_column_indices, int32_t rows, int32_t cols, int32_t nnz, int32_t bands,
                               mspmv_stream_t stream, int debug_sync)
{
    return plan_build<float>(d_plan, plan_bytes, d_values, d_row_offsets, d_column_indices, rows, cols, nnz, bands,
                             reinterpret_cast<hipStream_t>(stream), debug_sync);
}
int mspmv_csrmv_plan_build_f64(void *d_plan, size_t plan_bytes, const double *d_values, const int32_t *d_row_offsets,
                               const int32_t *d_column_indices, int32_t rows, int32_t cols, int32_t nnz, int32_t bands,
                               mspmv_stream_t stream, int debug_sync)
{
    return plan_build<double>(d_plan, plan_bytes, d_values, d_row_offsets, d_column_indices, rows, cols, nnz, bands,
                              reinterpret_cast<hipStream_t>(stream), debug_sync);
}
int mspmv_csrmv_plan_apply_f32(void *d_plan, size_t plan_bytes, const float *d_x, float *d_y, int32_t rows, int32_t cols, int32_t nnz,
                               int32_t bands, float alpha, float beta, mspmv_stream_t stream, int debug_sync)
{
    return plan_apply<float>(d_plan, plan_bytes, d_x, d_y, rows, cols, nnz, bands, alpha, beta, reinterpret_cast<hipStream_t>(stream),
                             debug_sync);
}
int mspmv_csrmv_plan_apply_f64(void *d_plan, size_t plan_bytes, const double *d_x, double *d_y, int32_t rows, int32_t cols, int32_t nnz,
                               int32_t bands, double alpha, double beta, mspmv_stream_t stream, int debug_sync)
{
    return plan_apply<double>(d_plan, plan_bytes, d_x, d_y, rows, cols, nnz, bands, alpha, beta, reinterpret_cast<hipStream_t>(stream),
                              debug_sync);
}

}  // extern "C"
