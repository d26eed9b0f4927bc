/*
 * merge_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity oracle).
 *
 * A plain-C restatement of the reference's CPU merge-path CsrMV and of the
 * pieces of its GPU decomposition that can be stated sequentially.  Nothing in
 * the product (merge_spmv_amd/, include/) links, loads or calls this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - oracle_merge_path_search is checked against the reference's own
 *     cub::MergePathSearch (cub/thread/thread_search.cuh:53-84) compiled from
 *     the reference tree by oracle/Makefile -> oracle/_ref/ref_search, for
 *     every diagonal of every fixture (tests/golden/*.json, made by
 *     oracle/make_golden.py).
 *   - y is checked against the reference's doc-comment known answer
 *     (cub/device/device_spmv.cuh:90-123) and closed forms of its generators.
 *   - cpu_spmv.cpp itself cannot be compiled here (unconditional
 *     #include <mkl.h>, cpu_spmv.cpp:61; no MKL in the image and we do not
 *     write a stand-in), so OmpMergeCsrmv is restated from the source text
 *     and cross-checked against the sequential definition only.
 *
 * Every function cites the reference lines it follows.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * MergePathSearch -- cpu_spmv.cpp:223-245 / cub/thread/thread_search.cuh:53-84
 * List A = row END offsets (a_len = rows), list B = counting 0..b_len-1.
 * "a[p] <= b[d-p-1]" with b the identity is "row_end[p] <= d-p-1".
 * ---------------------------------------------------------------------- */
void oracle_merge_path_search(int diagonal, const int *row_end, int a_len,
                              int b_len, int *out_x, int *out_y)
{
    int x_min = diagonal - b_len; if (x_min < 0) x_min = 0;
    int x_max = diagonal < a_len ? diagonal : a_len;
    while (x_min < x_max) {
        int pivot = (x_min + x_max) >> 1;
        if (row_end[pivot] <= diagonal - pivot - 1)
            x_min = pivot + 1;
        else
            x_max = pivot;
    }
    *out_x = x_min < a_len ? x_min : a_len;
    *out_y = diagonal - x_min;
}

/* 64-bit twin used to pin the multi-GPU partitioner (global
 * rows+nnz needs int64; same recurrence as above). */
void oracle_merge_path_search_i64(int64_t diagonal, const int64_t *row_end,
                                  int64_t a_len, int64_t b_len,
                                  int64_t *out_x, int64_t *out_y)
{
    int64_t x_min = diagonal - b_len; if (x_min < 0) x_min = 0;
    int64_t x_max = diagonal < a_len ? diagonal : a_len;
    while (x_min < x_max) {
        int64_t pivot = (x_min + x_max) >> 1;
        if (row_end[pivot] <= diagonal - pivot - 1)
            x_min = pivot + 1;
        else
            x_max = pivot;
    }
    *out_x = x_min < a_len ? x_min : a_len;
    *out_y = diagonal - x_min;
}

/* All coordinates for diagonals 0, step, 2*step, ... clamped to rows+nnz
 * (DeviceSpmvSearchKernel, dispatch_spmv_orig.cuh:104-143 with
 * step = TILE_ITEMS).  coords has 2*(count) ints: x0,y0,x1,y1,... */
void oracle_tile_coords(int rows, int nnz, const int *row_end, int step,
                        int count, int *coords)
{
    int total = rows + nnz;
    for (int t = 0; t < count; ++t) {
        long long d = (long long) t * step;
        int diag = d < total ? (int) d : total;
        oracle_merge_path_search(diag, row_end, rows, nnz,
                                 &coords[2 * t], &coords[2 * t + 1]);
    }
}

/* ------------------------------------------------------------------------
 * SpmvGold -- cpu_spmv.cpp:257-277, gpu_spmv.cu:72-92.
 * Sequential, accumulates in ValueT, y = beta*y_in + sum alpha*val*x[col].
 * ---------------------------------------------------------------------- */
#define DEF_GOLD(SUF, T)                                                      \
void oracle_spmv_gold_##SUF(int rows, const int *row_offsets, const int *col, \
                            const T *val, const T *x, const T *y_in, T *y_out,\
                            T alpha, T beta)                                  \
{                                                                             \
    for (int r = 0; r < rows; ++r) {                                          \
        T partial = beta * y_in[r];                                           \
        for (int k = row_offsets[r]; k < row_offsets[r + 1]; ++k)             \
            partial += alpha * val[k] * x[col[k]];                            \
        y_out[r] = partial;                                                   \
    }                                                                         \
}
DEF_GOLD(f32, float)
DEF_GOLD(f64, double)

/* Strict gold: fp64-accumulated sum g[r] and s[r] = sum|val*x|,
 * used for the stated tolerance |y-g| <= c*eps*s. */
#define DEF_GOLD64(SUF, T)                                                    \
void oracle_spmv_gold_acc64_##SUF(int rows, const int *row_offsets,           \
                                  const int *col, const T *val, const T *x,   \
                                  double *g, double *s)                       \
{                                                                             \
    _Pragma("omp parallel for schedule(dynamic, 4096)")                       \
    for (int r = 0; r < rows; ++r) {                                          \
        double acc = 0.0, mag = 0.0;                                          \
        for (int k = row_offsets[r]; k < row_offsets[r + 1]; ++k) {           \
            double p = (double) val[k] * (double) x[col[k]];                  \
            acc += p; mag += fabs(p);                                         \
        }                                                                     \
        g[r] = acc; s[r] = mag;                                               \
    }                                                                         \
}
DEF_GOLD64(f32, float)
DEF_GOLD64(f64, double)

/* ------------------------------------------------------------------------
 * OmpMergeCsrmv -- cpu_spmv.cpp:292-353.
 * Per "thread" tid: diagonals [min(ipt*tid,N), min(+ipt,N)) with
 * ipt = ceil((rows+nnz)/T) (:311-318); two searches (:320-321); whole rows
 * (:324-333); partial tail (:336-340); carry-out (:343-344); serial fix-up
 * guarded by row < rows (:348-352).
 * The reference's fixed row_carry_out[256]/value_carry_out[256] stack arrays
 * (:302-303) are heap arrays sized by num_threads here (it would smash its
 * stack for T > 256); results for T <= 256 are unaffected.
 * The association order depends only on num_threads, not on how many OS
 * threads execute the loop, so y is reproducible for a given num_threads.
 * Returns 0, or -1 on allocation failure.
 * ---------------------------------------------------------------------- */
#define DEF_OMP_MERGE(SUF, T)                                                 \
int oracle_omp_merge_csrmv_##SUF(int num_threads, int rows, int nnz,          \
                                 const int *row_end, const int *col,          \
                                 const T *val, const T *x, T *y)              \
{                                                                             \
    if (num_threads < 1) return -1;                                           \
    int *row_carry = (int *) malloc(sizeof(int) * (size_t) num_threads);      \
    T *val_carry = (T *) malloc(sizeof(T) * (size_t) num_threads);            \
    if (!row_carry || !val_carry) { free(row_carry); free(val_carry); return -1; } \
    _Pragma("omp parallel for schedule(static)")                              \
    for (int tid = 0; tid < num_threads; ++tid) {                             \
        int total = rows + nnz;                                               \
        int ipt = (int) (((long long) total + num_threads - 1) / num_threads);\
        long long sd = (long long) ipt * tid;                                 \
        int start_d = sd < total ? (int) sd : total;                          \
        long long ed = (long long) start_d + ipt;                             \
        int end_d = ed < total ? (int) ed : total;                            \
        int cx, cy, ex, ey;                                                   \
        oracle_merge_path_search(start_d, row_end, rows, nnz, &cx, &cy);      \
        oracle_merge_path_search(end_d, row_end, rows, nnz, &ex, &ey);        \
        for (; cx < ex; ++cx) {                                               \
            T running = 0;                                                    \
            for (; cy < row_end[cx]; ++cy)                                    \
                running += val[cy] * x[col[cy]];                              \
            y[cx] = running;                                                  \
        }                                                                     \
        T running = 0;                                                        \
        for (; cy < ey; ++cy)                                                 \
            running += val[cy] * x[col[cy]];                                  \
        row_carry[tid] = ex;                                                  \
        val_carry[tid] = running;                                             \
    }                                                                         \
    for (int tid = 0; tid < num_threads - 1; ++tid)                           \
        if (row_carry[tid] < rows)                                            \
            y[row_carry[tid]] += val_carry[tid];                              \
    free(row_carry); free(val_carry);                                         \
    return 0;                                                                 \
}
DEF_OMP_MERGE(f32, float)
DEF_OMP_MERGE(f64, double)

/* ------------------------------------------------------------------------
 * Tile-decomposed CsrMV: a sequential emulation of the reference's GPU
 * decomposition (Appendix B of SURVEY.md; DeviceSpmvKernel +
 * DeviceSegmentFixupKernel, dispatch_spmv_orig.cuh:157-224):
 *   every tile of `tile_items` diagonals stores y[i] for the rows that END in
 *   it (agent_spmv_orig.cuh:604-634) and emits one carry (row open at the tile
 *   end, partial) (:906-913); afterwards carries with key < rows are added
 *   (agent_segment_fixup.cuh:226-260, with the guard the reference lacks).
 * Within a tile, `threads` segments of ceil(tile_items/threads) items are
 * walked exactly as agent_spmv_orig.cuh:557-578 and combined left-to-right
 * (the order ReduceByKeyOp, thread_operators.cuh:291-301, yields for an
 * in-order scan).  carries_key/carries_val (num_tiles entries) are returned
 * for inspection; pass NULL to skip.  Used to pin the HIP kernels' tile
 * coordinates and carry pairs, not their rounding.
 * ---------------------------------------------------------------------- */
#define DEF_TILED(SUF, T)                                                     \
int oracle_tiled_csrmv_##SUF(int rows, int nnz, const int *row_end,           \
                             const int *col, const T *val, const T *x, T *y,  \
                             int tile_items, int *carries_key, T *carries_val)\
{                                                                             \
    long long total = (long long) rows + nnz;                                 \
    int num_tiles = (int) ((total + tile_items - 1) / tile_items);            \
    int *ck = (int *) malloc(sizeof(int) * (size_t) (num_tiles + 1));         \
    T *cv = (T *) malloc(sizeof(T) * (size_t) (num_tiles + 1));               \
    if (!ck || !cv) { free(ck); free(cv); return -1; }                        \
    for (int t = 0; t < num_tiles; ++t) {                                     \
        long long d0 = (long long) t * tile_items;                            \
        long long d1 = d0 + tile_items; if (d1 > total) d1 = total;           \
        int cx, cy, ex, ey;                                                   \
        oracle_merge_path_search((int) d0, row_end, rows, nnz, &cx, &cy);     \
        oracle_merge_path_search((int) d1, row_end, rows, nnz, &ex, &ey);     \
        T running = 0;                                                        \
        for (long long d = d0; d < d1; ++d) {                                 \
            if (cx < rows && cy >= row_end[cx]) {                             \
                y[cx] = running; running = 0; ++cx;                           \
            } else {                                                          \
                running += val[cy] * x[col[cy]]; ++cy;                        \
            }                                                                 \
        }                                                                     \
        if (cx != ex || cy != ey) { free(ck); free(cv); return -2; }          \
        ck[t] = ex; cv[t] = running;                                          \
    }                                                                         \
    for (int t = 0; t < num_tiles; ++t)                                       \
        if (ck[t] < rows) y[ck[t]] += cv[t];                                  \
    if (carries_key) memcpy(carries_key, ck, sizeof(int) * (size_t) num_tiles);\
    if (carries_val) memcpy(carries_val, cv, sizeof(T) * (size_t) num_tiles); \
    free(ck); free(cv);                                                       \
    return num_tiles;                                                         \
}
DEF_TILED(f32, float)
DEF_TILED(f64, double)

/* ------------------------------------------------------------------------
 * CompareResults -- utils.h:692-742 (float and double overloads are the same
 * computation: cast to float, integer difference of bit patterns, FAIL iff
 * sqrt(int_diff) > len).  Returns 1 on mismatch, 0 on "equal"; *first_bad
 * receives the index (or -1).  Kept to reproduce the reference's PASS/FAIL
 * line next to our strict check.
 * ---------------------------------------------------------------------- */
static int weak_cmp(float a, float b, long long len)
{
    int ia, ib;
    memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
    /* std::abs(int) of the wrapped difference, as the reference computes it */
    int diff = (int) ((unsigned) ia - (unsigned) ib);
    int int_diff = diff < 0 ? (int) (0u - (unsigned) diff) : diff;
    float sqrt_diff = sqrtf((float) int_diff);
    return sqrt_diff > (float) len;
}
int oracle_compare_results_f32(const float *computed, const float *reference,
                               int len, int *first_bad)
{
    if (first_bad) *first_bad = -1;
    for (int i = 0; i < len; ++i)
        if (weak_cmp(computed[i], reference[i], len)) {
            if (first_bad) *first_bad = i;
            return 1;
        }
    return 0;
}
int oracle_compare_results_f64(const double *computed, const double *reference,
                               int len, int *first_bad)
{
    if (first_bad) *first_bad = -1;
    for (int i = 0; i < len; ++i)
        if (weak_cmp((float) computed[i], (float) reference[i], len)) {
            if (first_bad) *first_bad = i;
            return 1;
        }
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
