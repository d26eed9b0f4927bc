/*
 * mspmv.h -- C ABI of the MI355X-native merge-based CsrMV (libmspmv.so).
 *
 * This is the drop-in boundary for the reference's device API
 *     cub::DeviceSpmv::CsrMV(d_temp_storage, temp_storage_bytes, d_values,
 *         d_row_offsets, d_column_indices, d_vector_x, d_vector_y,
 *         num_rows, num_cols, num_nonzeros, stream, debug_synchronous)
 * (reference cub/device/device_spmv.cuh:129-164, instantiated for float and
 * double with int offsets at gpu_spmv.cu:730,734).  Plain pointers and sizes
 * only; no C++ or torch types.  A header-only C++ shim with the reference's
 * spelling lives in merge_spmv_amd/host/device_spmv.hpp.
 *
 * Conventions kept from the reference:
 *   - two-phase temp storage: d_temp == NULL -> *temp_bytes receives the size
 *     needed, no work is done, returns 0 (dispatch_spmv_orig.cuh:651-655);
 *     otherwise *temp_bytes < needed -> hipErrorInvalidValue
 *     (util_device.cuh:90-93); d_temp (and a plan's storage) must be 16-byte
 *     aligned, as every device allocation is -- it holds 64-bit records that
 *     are updated atomically -- else hipErrorInvalidValue;
 *   - the caller owns every buffer including temp; the callee allocates
 *     nothing and keeps no state between calls (what a call leaves in temp
 *     storage is scratch, except for mspmv_csrmv_prepare); the library reads
 *     nothing from the environment and has no setters -- the forcing and
 *     re-tuning aids of the tests live in libmspmv_dev.so (include/mspmv_dev.h)
 *     only; the one piece of process state is the opt-in event profiler
 *     (mspmv_profile_begin/_end), which never changes what a call launches;
 *   - all array pointers are DEVICE pointers; d_row_offsets has rows+1 entries
 *     ([0]=0, [rows]=nnz, non-decreasing), 0-based int32 column indices,
 *     duplicates allowed (sparse_matrix.h:645-650,666-728); d_x holds cols
 *     readable entries (a tiny x -- cols * sizeof(value) <= 4 KB -- is copied
 *     to LDS whole, whether or not every column is referenced);
 *   - y is fully overwritten: y = A*x (alpha=1, beta=0 as
 *     device_spmv.cuh:155-156 forces); rows without entries get exactly 0;
 *   - asynchronous on `stream` unless debug_sync != 0 (then every kernel is
 *     followed by a stream sync and a launch-config line on stdout, like
 *     dispatch_spmv_orig.cuh:685-739);
 *   - return value: 0 (hipSuccess) or the first hipError_t as int.
 * Deliberate differences (the reference's out-of-bounds habits, SURVEY.md
 * Appendix B, are NOT inherited): nothing outside [0,rows] of d_row_offsets,
 * [0,nnz) of values/columns, [0,cols) of x or [0,rows) of y is touched.
 * Requires rows >= 0, cols >= 0, nnz >= 0 and rows + nnz <= 2^31 - 65537 (int32 path
 * arithmetic with one tile of slack).
 */
#ifndef MSPMV_H_
#define MSPMV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t passed as an opaque pointer so that C/ctypes/cgo callers need
 * no HIP headers (NULL = the default stream). */
typedef void *mspmv_stream_t;

#define MSPMV_VERSION 102 /* 0.1.2: + mspmv_get_clocked_bands (the clock-scheduled column bands serve the column-band candidates); 0.1.1: mspmv_launch_info_t grew (records_offset, layout_offset), the setters moved to mspmv_dev.h */
int mspmv_version(void);

/* hipGetErrorString for codes returned by this library. */
const char *mspmv_error_string(int status);

/* ---- the drop-in pair: replaces cub::DeviceSpmv::CsrMV<float|double> ---- */
int mspmv_csrmv_f32(void *d_temp, size_t *temp_bytes, const float *d_values,
                    const int32_t *d_row_offsets, const int32_t *d_column_indices,
                    const float *d_x, float *d_y, int32_t rows, int32_t cols,
                    int32_t nnz, mspmv_stream_t stream, int debug_sync);

int mspmv_csrmv_f64(void *d_temp, size_t *temp_bytes, const double *d_values,
                    const int32_t *d_row_offsets, const int32_t *d_column_indices,
                    const double *d_x, double *d_y, int32_t rows, int32_t cols,
                    int32_t nnz, mspmv_stream_t stream, int debug_sync);

/* ---- extension: y = alpha*A*x + beta*y.  The reference
 * parses --alpha/--beta (gpu_spmv.cu:721-722) and carries them in SpmvParams
 * (agent_spmv_orig.cuh:109-110) but its CsrMV forces 1/0.  With beta == 0 the
 * old contents of y are ignored (never read), as in BLAS. ---- */
int mspmv_csrmv_axpby_f32(void *d_temp, size_t *temp_bytes, const float *d_values,
                          const int32_t *d_row_offsets, const int32_t *d_column_indices,
                          const float *d_x, float *d_y, int32_t rows, int32_t cols,
                          int32_t nnz, float alpha, float beta,
                          mspmv_stream_t stream, int debug_sync);

int mspmv_csrmv_axpby_f64(void *d_temp, size_t *temp_bytes, const double *d_values,
                          const int32_t *d_row_offsets, const int32_t *d_column_indices,
                          const double *d_x, double *d_y, int32_t rows, int32_t cols,
                          int32_t nnz, double alpha, double beta,
                          mspmv_stream_t stream, int debug_sync);

/* ---- extension: SpMM, Y = alpha*A*X + beta*Y for k right-hand sides.
 * X is cols x k and Y is rows x k, both ROW-major with leading dimensions ldx, ldy >= k (elements):
 * the k entries X[col, :] that one column index needs are contiguous, so one gather serves a pack
 * of right-hand sides -- up to 16 bytes of them on the CsrMV-sized tiles, and 32 or 64 bytes (a whole
 * row of X) on smaller tiles when X is larger than 1 MiB, i.e. when its gathers miss the caches;
 * wider blocks run as several groups of the widest pack inside one pass over the matrix.  Matrices of 8 M path items and more
 * (X below 4 GB) take groups of 8 / 16 right-hand sides through the slot form instead (mspmv_spmm.hpp: spmm_lane_kernel -- every
 * nonzero share of a tile walked by four or eight lanes holding 16 bytes of right-hand sides each, rows written from registers);
 * ONE right-hand side stored as a plain vector (k = ldx = ldy = 1) is the CsrMV call.  Same two-phase temp storage, ownership, stream and error
 * conventions as mspmv_csrmv_*; with beta == 0 the old Y is never read.  Results per column are
 * within the same tolerance as CsrMV and bitwise reproducible.  No reference counterpart (the
 * reference ships CsrMV only). ---- */
int mspmv_csrmm_f32(void *d_temp, size_t *temp_bytes, const float *d_values,
                    const int32_t *d_row_offsets, const int32_t *d_column_indices,
                    const float *d_x, int32_t ldx, float *d_y, int32_t ldy,
                    int32_t rows, int32_t cols, int32_t nnz, int32_t k,
                    float alpha, float beta, mspmv_stream_t stream, int debug_sync);
int mspmv_csrmm_f64(void *d_temp, size_t *temp_bytes, const double *d_values,
                    const int32_t *d_row_offsets, const int32_t *d_column_indices,
                    const double *d_x, int32_t ldx, double *d_y, int32_t ldy,
                    int32_t rows, int32_t cols, int32_t nnz, int32_t k,
                    double alpha, double beta, mspmv_stream_t stream, int debug_sync);

/* ---- extension for iterated SpMV (solvers): the tile coordinates -- the output of the
 * reference's DeviceSpmvSearchKernel, dispatch_spmv_orig.cuh:104-143 -- depend on d_row_offsets
 * alone, yet the reference's stateless CsrMV recomputes them on every call (8-22 us here, 5-15 % of
 * a mid-size SpMV).  mspmv_csrmv_prepare runs that pass once into the caller's temp storage
 * (same two-phase size query; the size equals mspmv_csrmv_*'s for the same rows/nnz/value_bytes).  The default
 * one-launch kernel treats the coordinates it finds in temp storage as hints and verifies them, so a stateless
 * call that follows another on the same temp storage and matrix already runs without a search; preparing makes
 * the FIRST call as fast as the later ones, and it is what the classic pipeline (column-band candidates, arrays that
 * are not 16-byte aligned) skips its coordinate launch on;
 * mspmv_csrmv_prepared_* then compute y = alpha*A*x + beta*y with the coordinates found there.
 * The caller guarantees that d_temp was prepared for this d_row_offsets / rows / nnz / value_bytes
 * and has since been used only by mspmv calls for the same matrix: they leave the prepared coordinates
 * intact (the one family of calls that picks another tile shape by its column count -- mspmv_get_launch_info_cols --
 * keeps that shape's hints in a region of its own behind the default layout, and only in the one-launch form, whose
 * hints are verified).  mspmv_csrmv_prepared_* pick their shape by the same rule as the stateless calls, so results
 * are bitwise those of mspmv_csrmv_* / mspmv_csrmv_axpby_* given the same temp_bytes. ---- */
int mspmv_csrmv_prepare(void *d_temp, size_t *temp_bytes, const int32_t *d_row_offsets,
                        int32_t rows, int32_t nnz, int32_t value_bytes,
                        mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_prepared_f32(void *d_temp, size_t *temp_bytes, const float *d_values,
                             const int32_t *d_row_offsets, const int32_t *d_column_indices,
                             const float *d_x, float *d_y, int32_t rows, int32_t cols,
                             int32_t nnz, float alpha, float beta,
                             mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_prepared_f64(void *d_temp, size_t *temp_bytes, const double *d_values,
                             const int32_t *d_row_offsets, const int32_t *d_column_indices,
                             const double *d_x, double *d_y, int32_t rows, int32_t cols,
                             int32_t nnz, double alpha, double beta,
                             mspmv_stream_t stream, int debug_sync);

/* ---- extension: PREPARED PLAN for a gather-bound matrix multiplied many times (opt-in; the stateless
 * drop-in calls above never use it).  When x is larger than an XCD's 4 MiB L2, ~70 % of the x gathers miss
 * and every miss moves a 128-byte line: that, not HBM, bounds mspmv_csrmv_* on such a matrix (BASELINE
 * config 2).  mspmv_csrmv_plan_build_* makes, ONCE, a band-major copy of the matrix in the caller's
 * storage: the columns are cut into `bands` equal bands (0 = pick: the fewest of 2, 4, 8, 16, ... with <= 3.25 MiB
 * of x per band; 1 when x fits anyway) and the entries of band b form the b-th block of rows of a stacked CSR
 * matrix; mspmv_csrmv_plan_apply_* then runs the ordinary merge-path CsrMV over the stacked matrix with
 * one contiguous tile range per XCD -- so each XCD's L2 only ever holds one band's slice of x and the CSR
 * stream is read exactly once -- and folds the bands' partial sums in band order:
 *     y = alpha * A * x + beta * y        (beta == 0: y is never read).
 * Conventions: caller-owned storage of mspmv_csrmv_plan_size bytes (device memory; the same rows / cols /
 * nnz / bands must be passed to _size, _build and _apply); nothing is allocated; asynchronous on `stream`;
 * the original arrays are not needed after _build.  Results are within the CsrMV tolerance and bitwise
 * reproducible for a given plan.  Needs bands * rows + nnz <= 2^31 - 65537 (hipErrorInvalidValue otherwise:
 * use the stateless call).  This is the counterpart of what the reference's driver does for its HYB
 * comparison -- conversion timed once as set-up, SpMV timed separately (gpu_spmv.cu:106-257). ---- */
int mspmv_csrmv_plan_size(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, int32_t bands,
                          size_t *plan_bytes, int32_t *bands_used);
int mspmv_csrmv_plan_build_f32(void *d_plan, size_t plan_bytes, const float *d_values,
                               const int32_t *d_row_offsets, const int32_t *d_column_indices,
                               int32_t rows, int32_t cols, int32_t nnz, int32_t bands,
                               mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_plan_build_f64(void *d_plan, size_t plan_bytes, const double *d_values,
                               const int32_t *d_row_offsets, const int32_t *d_column_indices,
                               int32_t rows, int32_t cols, int32_t nnz, int32_t bands,
                               mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_plan_apply_f32(void *d_plan, size_t plan_bytes, const float *d_x, float *d_y,
                               int32_t rows, int32_t cols, int32_t nnz, int32_t bands,
                               float alpha, float beta, mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_plan_apply_f64(void *d_plan, size_t plan_bytes, const double *d_x, double *d_y,
                               int32_t rows, int32_t cols, int32_t nnz, int32_t bands,
                               double alpha, double beta, mspmv_stream_t stream, int debug_sync);

/* ---- extension: HOT-COLUMN PLAN for a matrix whose x is far larger than the caches and whose columns are referenced
 * very unevenly (scale-free graphs: BASELINE config 5; opt-in, the stateless calls never use it).  Such a CsrMV runs at
 * the DRAM random-line rate -- 57 G gathers/s on MI355X, 0.09 of the HBM roofline -- because the few hot columns that
 * take most of the references are scattered over all of x, each sharing its cache line with cold ones.
 * mspmv_csrmv_hotcols_build renumbers the columns ONCE in order of reference count (by class: floor(log2(count + 1)),
 * hottest first -- a histogram and one ranking pass, no sort) into the caller's storage: the renumbered column indices
 * (4 * nnz bytes; values and row offsets are used from the caller's arrays, not copied), the permutation, x in the new
 * numbering and the inner call's temp storage.  mspmv_csrmv_hotcols_apply_* then permutes x (one pass over cols entries)
 * and runs the ordinary stateless CsrMV on the renumbered indices:
 *     y = alpha * A * x + beta * y        (beta == 0: y is never read).
 * A column permutation only changes where x is read -- every row still sums the same products in the same order -- so y
 * is BIT FOR BIT the result of mspmv_csrmv_* / _axpby_* in its ONE-LAUNCH form.  The plan's inner call is never a CANDIDATE for
 * the column-band passes; a stateless call whose sizes make it one (mspmv_get_band_passes > 1: x of 5.5-40 MiB, >= 8 nonzeros
 * per row, a CSR stream >= 160 MiB) runs the classic three launches -- one carry per tile and a fix-up, another association of
 * the sums of rows that cross tiles -- whether or not the device-side windows then let the passes run, so compare with
 * the one-launch form there (dev library: mspmv_set_band_passes(vb, -1)).  Config 5 on one GPU: 34.2 -> 20.6 ms.  Same conventions as
 * the band-major plan below (caller-owned storage of mspmv_csrmv_hotcols_size bytes, the same rows / cols / nnz /
 * value_bytes to every call, asynchronous on `stream`); d_values / d_row_offsets passed to _apply must be the arrays
 * the plan was built for.  No reference counterpart (its HYB column is the precedent for set-up timed apart,
 * gpu_spmv.cu:106-257). ---- */
/* Would the plan pay?  A cheap look at the column indices (synchronous: a bitmap of one bit per 128-byte line of x, one small kernel,
 * a 12-byte copy): 512 windows of 2048 consecutive nonzeros spread over the matrix (~1 M references); *distinct_permille_of_uniform =
 * 1000 x the number of DISTINCT lines of x the sample touches / what as many uniformly drawn references would touch (-1: fewer than
 * 2048 nonzeros), *wide_windows = how many of the 512 windows span >= 3/4 of the columns.  Uniformly spread columns give ~1000
 * (nothing to concentrate), stencils and bands < 100 (their gathers hit the caches as they are), scale-free matrices 400-700: lines
 * that keep coming back -- the case the plan is for.  mspmv_mg_plan_* builds the plan by itself when x is beyond the Infinity Cache
 * and 150 <= the figure < 800 with >= 256 wide windows (mspmv_mg_plan_hot_columns).  Measured (profiles/r05_skew_probe.txt):
 * uniform columns 1000 / 512 wide, a band 48 / 0, a 5-point grid 206 / 0, R-MAT scale 22-26 289-374 / 370-450 (config 5: 351 / 372),
 * a circuit-shaped matrix 634 / 492, a small R-MAT (scale 18: every line of its 2 MB x referenced) 842 / 456. */
int mspmv_csrmv_hotcols_skew(const int32_t *d_column_indices, int32_t cols, int32_t nnz, int32_t value_bytes, mspmv_stream_t stream,
                             int32_t *distinct_permille_of_uniform, int32_t *wide_windows);
int mspmv_csrmv_hotcols_size(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, size_t *plan_bytes);
int mspmv_csrmv_hotcols_build(void *d_plan, size_t plan_bytes, const int32_t *d_row_offsets,
                              const int32_t *d_column_indices, int32_t rows, int32_t cols, int32_t nnz,
                              int32_t value_bytes, mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_hotcols_apply_f32(void *d_plan, size_t plan_bytes, const float *d_values,
                                  const int32_t *d_row_offsets, const float *d_x, float *d_y,
                                  int32_t rows, int32_t cols, int32_t nnz, float alpha, float beta,
                                  mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_hotcols_apply_f64(void *d_plan, size_t plan_bytes, const double *d_values,
                                  const int32_t *d_row_offsets, const double *d_x, double *d_y,
                                  int32_t rows, int32_t cols, int32_t nnz, double alpha, double beta,
                                  mspmv_stream_t stream, int debug_sync);
/* A caller that keeps x in the plan's numbering -- a fixed right-hand side, or a method whose vector updates are element-wise and can
 * live in that numbering -- pays the permutation once instead of per SpMV (config 5 on one GPU: 0.7 of 21 ms):
 * mspmv_csrmv_hotcols_permute_* writes d_x_permuted[k] = d_x[order[k]] (cols entries; not in place), and
 * mspmv_csrmv_hotcols_apply_permuted_* is mspmv_csrmv_hotcols_apply_* on such a vector, without the pass.  y comes out in the ORIGINAL
 * row order, bit for bit what _apply_* returns for the unpermuted x. */
int mspmv_csrmv_hotcols_permute_f32(const void *d_plan, size_t plan_bytes, const float *d_x, float *d_x_permuted,
                                    int32_t rows, int32_t cols, int32_t nnz, mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_hotcols_permute_f64(const void *d_plan, size_t plan_bytes, const double *d_x, double *d_x_permuted,
                                    int32_t rows, int32_t cols, int32_t nnz, mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_hotcols_apply_permuted_f32(void *d_plan, size_t plan_bytes, const float *d_values,
                                           const int32_t *d_row_offsets, const float *d_x_permuted, float *d_y,
                                           int32_t rows, int32_t cols, int32_t nnz, float alpha, float beta,
                                           mspmv_stream_t stream, int debug_sync);
int mspmv_csrmv_hotcols_apply_permuted_f64(void *d_plan, size_t plan_bytes, const double *d_values,
                                           const int32_t *d_row_offsets, const double *d_x_permuted, double *d_y,
                                           int32_t rows, int32_t cols, int32_t nnz, double alpha, double beta,
                                           mspmv_stream_t stream, int debug_sync);
/* the plan's pieces (device pointers into d_plan): order[k] = the original column that became column k (cols entries),
 * and the renumbered column indices (nnz entries) */
const int32_t *mspmv_csrmv_hotcols_order(const void *d_plan, int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes);
const int32_t *mspmv_csrmv_hotcols_columns(const void *d_plan, int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes);

/* ---- introspection (the counterpart of the reference's debug_synchronous
 * launch log, dispatch_spmv_orig.cuh:685-739, as data) ---- */
typedef struct mspmv_launch_info {
    int32_t block_threads;     /* threads per merge tile                      */
    int32_t items_per_thread;  /* merge items per thread                      */
    int32_t tile_items;        /* block_threads * items_per_thread            */
    int32_t num_tiles;         /* ceil((rows+nnz) / tile_items)               */
    int32_t fixup_chunk;       /* carry pairs per fix-up block                */
    int32_t fixup_levels;      /* fix-up launches (0 when num_tiles <= 1)     */
    int32_t flags;             /* option bits in effect: 0 in this library (mspmv_dev.h: MSPMV_TUNE_*) */
    int32_t snap_head_max;     /* > 0: calls of these sizes run ONE launch of row-snapped tiles (tile_kernel_snap;
                                  16-byte aligned arrays assumed, decided again per call):
                                  a tile boundary that falls <= this many nonzeros into a row is moved to the row's
                                  first nonzero, so the per-tile carry that mspmv_debug_read_tiles returns is 0
                                  there; 0: classic tiles                    */
    uint64_t temp_bytes;       /* what the size query returns                 */
    uint64_t coords_offset;    /* byte offsets of regions inside temp         */
    uint64_t carries_offset;
    uint64_t diag_offset;      /* two int32: [0] = tag of the last one-launch call in which a tile gave up waiting for another
                                  workgroup's record and computed the sum itself (debug_sync reports it), [1] = how many such
                                  episodes this temp storage has seen; neither is ever needed for a result */
    uint64_t records_offset;   /* the tagged records of the one-launch kernel: 16 bytes per tile + per group of 64 tiles; every slot a
                                  launch TOUCHED is (0, 0) again once it has ended (slots it never used keep what the buffer held:
                                  tests/test_forward_progress.py zeroes the buffer first; tests/test_record_protocol_model.py) */
} mspmv_launch_info_t;

/* value_bytes = 4 (float) or 8 (double).  The layout of a call of these sizes under the default choice of tile shape; temp_bytes is
 * large enough for any column count (one family of calls picks its shape by the column count too: see mspmv_get_launch_info_cols). */
int mspmv_get_launch_info(int32_t rows, int32_t nnz, int32_t value_bytes,
                          mspmv_launch_info_t *info);
/* The same with the column count, i.e. exactly what mspmv_csrmv_f32 / _f64 (_axpby_*, _prepared_*) run for these sizes when given at
 * least info->temp_bytes of temp storage and 16-byte aligned arrays: a large fp64 matrix (more than 8 M rows + nonzeros, at most 256 MB
 * of CSR arrays) of at most 8 nonzeros per row on average over an x of at most 4 KB -- the reference's --dense=<cols> inputs,
 * cpu_spmv.cpp:581-587 -- takes the small tile shape behind the compact front end (csrc/mspmv_api.hip: skinny_rule); its layout sits
 * BEHIND the default one in temp storage (the offsets reported here are absolute), so temp_bytes is the sum of the two.  Rows of
 * closed lean tiles -- every row of a --dense input -- are summed left to right whatever the shape, so their y does not change by a
 * bit; a long row among them is associated as that shape's tiles cut it (the stated bound holds either way). */
int mspmv_get_launch_info_cols(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes,
                               mspmv_launch_info_t *info);

/* Copy the tile coordinates ((num_tiles+1) x {row, nonzero}) and the per-tile
 * carry pairs (num_tiles keys + values of value_bytes each) that the last
 * csrmv call left in d_temp back to HOST arrays (synchronises `stream`).
 * They correspond to d_tile_coordinates / d_tile_carry_pairs of
 * dispatch_spmv_orig.cuh:643-646 and are what the parity tests pin against
 * the oracle.  Any output pointer may be NULL.  (Laid out as mspmv_get_launch_info says: not for the calls that pick their
 * tile shape by the column count, mspmv_get_launch_info_cols.) */
int mspmv_debug_read_tiles(const void *d_temp, int32_t rows, int32_t nnz,
                           int32_t value_bytes, int32_t *h_coords,
                           int32_t *h_carry_keys, void *h_carry_values,
                           mspmv_stream_t stream);

/* What the automatic column-band policy is derived from on the current device: one XCD's L2 in bytes and the number of XCDs
 * (queried from the runtime once per device), and the
 * CU count.  Without a device: the MI355X figures (4 MiB, 8, 256).  Any pointer may be NULL. */
int mspmv_get_device_caches(int64_t *l2_bytes_per_xcd, int32_t *xcds, int32_t *cus);
/* Measuring aid: ONE launch of a bare read stream over d_buf (16-byte aligned; bytes / 16 sixteen-byte loads, 256 x 11 per block
 * like a tile's nonzero stream; nontemporal != 0: non-temporal loads), asynchronous on `stream`.  Timed over a buffer that stays
 * in the Infinity Cache it gives the rate a cache-resident SpMV's algorithmic bytes are to be read against -- the HBM peak is not
 * the bound of such a call (bench.py: `roofline.bound` = "infinity_cache"). */
int mspmv_probe_read_stream(const void *d_buf, size_t bytes, int32_t nontemporal, mspmv_stream_t stream);
/* Column bands (extension).  A large matrix whose columns are spread uniformly over an x of 1.375-10 x one XCD's L2 (fp32;
 * 1.75-9 x in fp64: 5.5-40 / 7-36 MiB on MI355X) is gather-bound at the Infinity-Cache rate when x is gathered as it comes.  Such a call
 * (a CANDIDATE by its sizes: *passes > 1 below) stays stateless, asynchronous and three launches: 64 blocks added to the coordinate
 * launch sample 64 windows of 2048 consecutive column indices, and the tile kernel reads their verdicts and runs either its ordinary
 * body or, when the columns are spread, the banded form:
 *   CLOCK-SCHEDULED BANDS (round 6; csrc/mspmv_tdm.hpp; what the library runs): one pass.  A block sorts its tile's nonzeros by column
 *     band (1 MiB of x) in LDS and gathers band by band, the band "on air" being read off the chip-wide 100 MHz clock -- blocks never
 *     talk to each other, yet at any moment every XCD gathers from a band or two of x, which its L2 keeps.  y is BIT FOR BIT the
 *     classic three-launch result without bands.  C2: 1.22 -> 0.64 ms (fp32), 1.57 -> 1.00 ms (fp64).
 *   COLUMN-BAND PASSES (rounds 2-5; development library only since: mspmv_set_tdm(vb, -1)): the CSR stream read 2-4 times, each pass
 *     multiplying the nonzeros of one band; a re-association of the sums (0.84 / 1.30 ms on C2).
 * Either way results stay within the strict bound and are bitwise reproducible.
 * *passes = how many passes a call of these sizes would be offered (0: none, not a candidate; the aligned, vectorised path is
 * assumed; csrc/mspmv_api.hip: band_passes_for); the device-side verdicts have the last word. */
int mspmv_get_band_passes(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, int32_t *passes);
/* The bands of the clock-scheduled form for a call of these sizes: *bands = how many (0: not a candidate), *band_cols = columns per
 * band (a power of two: 1 MiB of x, widened until 32 bands cover it). */
int mspmv_get_clocked_bands(int32_t rows, int32_t cols, int32_t nnz, int32_t value_bytes, int32_t *bands, int32_t *band_cols);
/* The 64 window verdicts (1 = columns look uniformly spread) the last automatic call left in d_temp -> HOST
 * array of 64 int32 (synchronises `stream`); at least 56 ones select the band passes. */
int mspmv_debug_band_windows(const void *d_temp, int32_t rows, int32_t nnz, int32_t value_bytes,
                             int32_t *h_verdicts, mspmv_stream_t stream);

/* Opt-in per-kernel timing with hipEvents recorded on the caller's stream
 * around each of the three passes (the counterpart of the reference's
 * GpuTimer, utils.h:624-658, at kernel granularity).  While active, every
 * csrmv call (up to max_calls) records 4 events; mspmv_profile_end
 * synchronises them, returns the number of profiled calls and the AVERAGE
 * milliseconds per call of the search, tile and fix-up passes, and turns
 * profiling off.  Process-global measuring aid for ONE host thread (a mutex
 * keeps concurrent callers from corrupting it, nothing more); calls that only
 * run the coordinate pass (mspmv_csrmv_prepare) and SpMM calls are not recorded. */
int mspmv_profile_begin(int32_t max_calls);
int mspmv_profile_end(int32_t *calls, float *search_ms, float *tile_ms, float *fixup_ms);

/* ---- multi-GPU merge partitioning (new design, the reference
 * has no multi-device code: its claim is README.md:5).  Host-side, 64-bit. --
 *
 * The global merge path (rows + nnz items) is cut at `parts` equally spaced
 * diagonals; part g owns rows-ending [row_split[g], row_split[g+1]) and
 * nonzeros [nz_split[g], nz_split[g+1]).  Seen locally, part g is an ordinary
 * CSR matrix with  local_rows = row_split[g+1]-row_split[g] + 1  rows: its
 * first row may be the tail of a row begun on an earlier part, and its extra
 * LAST row is the open row cut by the part's right boundary, so the local
 * y[local_rows-1] IS the carry for global row row_split[g+1].
 * h_row_offsets: HOST int64 [rows+1].  row_split/nz_split: HOST int64
 * [parts+1]. */
int mspmv_mg_partition(const int64_t *h_row_offsets, int64_t rows, int64_t nnz,
                       int32_t parts, int64_t *row_split, int64_t *nz_split);

/* Fill the local int32 row_offsets (local_rows+1 entries, rebased by
 * -nz_split[g]) of part g. Returns hipErrorInvalidValue if the part does not
 * fit int32. */
int mspmv_mg_local_offsets(const int64_t *h_row_offsets, int64_t rows,
                           int64_t row_begin, int64_t row_end_,
                           int64_t nz_begin, int64_t nz_end_,
                           int32_t *h_local_offsets);

#define MSPMV_MG_MAX_PARTS 64   /* parts a carry exchange can address (one mask bit each) */

/* After the one exchange (all-gather of every part's carry value into
 * d_carries[parts]): add to d_y_local[0] every carry of parts j < part whose
 * key row_split[j+1] equals row_split[part] ... i.e. the rows this part owns
 * that were begun earlier.  keys are given by the HOST array row_split.
 * Deterministic (rank order).  value_bytes = 4 or 8.  parts <= MSPMV_MG_MAX_PARTS
 * (mspmv_mg_partition itself has no such limit). */
int mspmv_mg_apply_carries(void *d_y_local, const void *d_carries,
                           const int64_t *row_split, int32_t parts,
                           int32_t part, int32_t value_bytes,
                           mspmv_stream_t stream);

/* ---- the multi-GPU operator (mspmv_mg_plan_create / _csrmv / _destroy).  New design; the
 * reference has a single --device (utils.h:465-474).
 *
 * A plan drives the parts ONE PROCESS holds of a matrix cut by mspmv_mg_partition:
 *   - all of them (local_parts == parts: single-process form; device_ids may repeat, so a 1-GPU box can run
 *     G parts on one device), or
 *   - some, typically one (one process per GPU; every process passes the same 128-byte id128 that ONE of
 *     them obtained from mspmv_mg_unique_id and shipped by whatever means the launcher has).
 * The caller owns the part's CSR arrays (values, column indices and the local int32 row offsets of
 * mspmv_mg_local_offsets, all on the part's device) and attaches them with mspmv_mg_plan_set_part, which
 * also finds the part's tile coordinates once.  The plan owns one stream per part, the replicated x (one
 * replica per distinct device, mspmv_mg_plan_x -- the caller fills it before the first SpMV), the row-sharded
 * result (mspmv_mg_plan_y: the part's owned rows, row_split[id+1]-row_split[id] entries, followed by one
 * scratch entry) and the exchange buffers.
 *
 * mspmv_mg_csrmv: y = A*x on every local part + the ONE exchange of the boundary-row carries + the owners'
 * adds, all asynchronous on the parts' streams (mspmv_mg_synchronize waits).  Exchange backends:
 *   MSPMV_MG_EXCHANGE_PEER  (single-process form only) events between the parts' streams and one small kernel
 *                           that reads the carries straight out of the peers' memory over xGMI;
 *   MSPMV_MG_EXCHANGE_RCCL  one ncclAllGather of a scalar per part (needs distinct devices; librccl is loaded
 *                           on first use, the library has no link-time dependency on it);
 *   MSPMV_MG_EXCHANGE_IPC   any split of the parts over processes (typically one process per GPU), no collective library:
 *                           after creating its plan every process calls mspmv_mg_plan_ipc_export, the launcher gathers the
 *                           blobs (MPI / torch.distributed / a file -- anything), and every process passes ALL of them to
 *                           mspmv_mg_plan_ipc_import, which opens the peers' x replicas and mailbox blocks through hipIpc.
 *                           A step is then the SpMV + one tiny kernel that writes the carry, tagged with the step number,
 *                           straight into its owner's mailbox (a peer write) and, on the owner, waits for
 *                           the tags of its sources, adds them in part order and acknowledges; the row all-gather is the
 *                           PEER backend's pushes fenced by step-tagged flags.  Nothing on the host, no rendezvous; a
 *                           producer runs at most two steps ahead of its consumer.  Waits are bounded (seconds):
 *                           mspmv_mg_synchronize returns hipErrorLaunchFailure if one ran out.  (HSA_ENABLE_IPC_MODE_LEGACY=0
 *                           where the host driver only supports dmabuf IPC.)  EXPERIMENTAL: exercised with several processes
 *                           sharing ONE device only (no multi-GPU node has run it); a process's local parts must sit on one
 *                           device; the mailboxes need uncached or fine-grained device memory (hipErrorNotSupported otherwise).
 *   MSPMV_MG_EXCHANGE_AUTO  PEER when the process holds every part, else RCCL.
 * mspmv_mg_allgather_rows (square matrices): x <- y on every replica -- PEER: each part
 * pushes its owned rows into every replica (direct peer writes, unpadded); RCCL: grouped ncclBroadcast.
 * Results are deterministic (carries are added in part order) and within the CsrMV tolerance.
 * Returns 0 or a hipError_t; hipErrorNotSupported = librccl could not be loaded, hipErrorUnknown = an
 * RCCL call failed (message on stderr). ---- */
typedef struct mspmv_mg_plan mspmv_mg_plan_t;
#define MSPMV_MG_EXCHANGE_AUTO 0
#define MSPMV_MG_EXCHANGE_RCCL 1
#define MSPMV_MG_EXCHANGE_PEER 2
#define MSPMV_MG_EXCHANGE_IPC  3

typedef struct mspmv_mg_info {
    int32_t parts, local_parts, exchange /* backend in effect */, value_bytes, replicas, hot_parts /* local parts running the hot-column plan */;
    int64_t rows, cols;
    uint64_t carry_bytes_per_step;       /* payload of the carry exchange: parts * value_bytes        */
    uint64_t allgather_bytes_per_step;   /* payload of y -> x: rows * value_bytes to every OTHER GPU  */
    uint64_t steps;                      /* mspmv_mg_csrmv calls so far                               */
} mspmv_mg_info_t;

int mspmv_mg_unique_id(void *id128);
int mspmv_mg_plan_create(mspmv_mg_plan_t **plan, int32_t parts, int32_t local_parts, const int32_t *part_ids,
                         const int32_t *device_ids, const int64_t *row_split, const int64_t *nz_split,
                         int64_t cols, int32_t value_bytes, int32_t exchange, const void *id128);
/* IPC backend: blob == NULL -> *blob_bytes = the size of this process's blob; else the blob is written.  _import takes `count`
 * blobs laid out `blob_stride` bytes apart (every process's, its own included, in any order), once. */
int mspmv_mg_plan_ipc_export(mspmv_mg_plan_t *plan, void *blob, size_t *blob_bytes);
int mspmv_mg_plan_ipc_import(mspmv_mg_plan_t *plan, const void *blobs, int32_t count, size_t blob_stride);
int mspmv_mg_plan_set_part(mspmv_mg_plan_t *plan, int32_t local_index, const void *d_values,
                           const int32_t *d_local_row_offsets, const int32_t *d_column_indices);
void *mspmv_mg_plan_x(mspmv_mg_plan_t *plan, int32_t local_index);
void *mspmv_mg_plan_y(mspmv_mg_plan_t *plan, int32_t local_index);
mspmv_stream_t mspmv_mg_plan_stream(mspmv_mg_plan_t *plan, int32_t local_index);
int mspmv_mg_plan_info(mspmv_mg_plan_t *plan, mspmv_mg_info_t *info);
/* The hot-column plan of the parts (above: columns renumbered by reference count, built once per part in plan-owned storage of
 * 4 * local_nnz + 16 * cols bytes; mspmv_mg_csrmv permutes x into the part's numbering before its SpMV):
 *   enable < 0   AUTOMATIC, the default of every plan: decided per part when its matrix is attached (mspmv_mg_plan_set_part) -- a
 *                part gets the plan when the x replica is beyond the 256 MB Infinity Cache (cols * value_bytes) AND the on-device
 *                sample of its column indices says the columns come back (mspmv_csrmv_hotcols_skew: ~1 M sampled references touch 15-80 % of
 *                the distinct lines of x a uniform draw would, >= 256 of 512 windows spanning most of x): config 5, every part 33 -> 21 ms-equivalent.  Uniformly
 *                spread columns, stencils / bands, an x that fits the cache, or a part that cannot afford the storage: no plan.
 *   enable > 0   always (every local part; hipErrorOutOfMemory if one cannot)
 *   enable == 0  never; releases the storage
 * y is bit for bit the same either way (parts that would take the column-band passes excepted, as above); mspmv_mg_plan_info reports
 * how many local parts run it (hot_parts).  May be called before the parts' matrices are attached: the mode is then applied by
 * mspmv_mg_plan_set_part, which -- in the automatic and "always" modes -- runs a SYNCHRONOUS probe / build on the part's stream and
 * may allocate the storage above; under "always" a part that cannot afford it makes set_part return hipErrorOutOfMemory with the
 * matrix attached and no plan (the part runs the ordinary call). */
int mspmv_mg_plan_hot_columns(mspmv_mg_plan_t *plan, int32_t enable);
/* Milliseconds the EXCHANGE of the last mspmv_mg_csrmv took on local part `local_index` -- hipEvents on the part's stream right after
 * its SpMV and right after its share of the exchange (the all-gather / the peers' events, the owner's add): what a step costs beyond
 * its kernels, measured rather than inferred; includes the time this part waited for slower parts.  Synchronises with the step. */
int mspmv_mg_plan_exchange_ms(mspmv_mg_plan_t *plan, int32_t local_index, float *ms);
int mspmv_mg_csrmv(mspmv_mg_plan_t *plan);
int mspmv_mg_allgather_rows(mspmv_mg_plan_t *plan);
int mspmv_mg_synchronize(mspmv_mg_plan_t *plan);
int mspmv_mg_plan_destroy(mspmv_mg_plan_t *plan);

#ifdef __cplusplus
}
#endif
#endif /* MSPMV_H_ */
