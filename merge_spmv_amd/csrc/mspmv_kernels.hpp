// mspmv_kernels.hpp -- hand-written CDNA4 (gfx950, wave64) kernels of the
// merge-based CsrMV.  Replaces the reference's cub/agent/agent_spmv_orig.cuh,
// agent_segment_fixup.cuh and the kernels of
// cub/device/dispatch/dispatch_spmv_orig.cuh:68-224; none of that code (nor
// any CUB/hipCUB/rocPRIM primitive) is used.  The algorithm contract is
// SURVEY.md Appendix B.
//
// Three device passes per SpMV, all on the caller's stream:
//   1. search_kernel : one WAVE per tile boundary does a 64-ary merge-path
//                      search (4 dependent loads for 2^24 rows instead of 24)
//                      -> coords[tile]           (ref: DeviceSpmvSearchKernel)
//   2. tile_kernel   : one 256-thread block per merge tile: coalesced stream
//                      of (col,val), gather of x, products and row-end
//                      offsets staged in LDS, per-thread merge-path walk,
//                      wave64 shuffle segmented scan for the partial-row
//                      carries, y stored for rows ending in the tile, one
//                      (row, partial) carry per tile     (ref: DeviceSpmvKernel)
//   3. fixup_kernel  : deterministic reduce-by-key over the per-tile carries,
//                      y[row] += sum.  Chunked two-level instead of the
//                      reference's decoupled look-back (no spin-waits, no
//                      forward-progress assumption) and instead of its fp32
//                      atomics (run-to-run reproducible). (ref: DeviceSegmentFixupKernel)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mspmv {

constexpr int WAVE = 64;

struct Coord {
    int x;  // row-ends consumed   (list A index)
    int y;  // nonzeros consumed   (list B index)
};

template <typename V> struct Carry;
template <> struct alignas(8) Carry<float> { int key; float value; };
template <> struct alignas(16) Carry<double> { int key; int pad; double value; };

template <typename V>
struct Params {
    const V *__restrict__ values;
    const int *__restrict__ row_end;  // d_row_offsets + 1 (device_spmv.cuh:148)
    const int *__restrict__ cols;
    const V *__restrict__ x;
    V *__restrict__ y;
    int rows;
    int nnz;
    V alpha;
    V beta;
};

// ---------------------------------------------------------------------------
// Merge-path diagonal search (reference
// cub/thread/thread_search.cuh:53-84), wave-cooperative.
//   smallest p in [lo, hi] with p == hi or row_end[p] + p + 1 > diagonal.
// row_end[p] + p is strictly increasing, so the predicate is monotone; each
// round the wave's 64 lanes probe the last element of 64 equal chunks of the
// candidate range and a ballot picks the chunk: range / 64 per dependent load.
// All lanes return the same coordinate.
// ---------------------------------------------------------------------------
__device__ __forceinline__ Coord wave_merge_path_search(int diagonal, const int *__restrict__ row_end,
                                                        int rows, int nnz)
{
    const int lane = threadIdx.x & (WAVE - 1);
    int lo = diagonal - nnz; lo = lo < 0 ? 0 : lo;
    int hi = diagonal < rows ? diagonal : rows;
    while (lo < hi) {
        const int n = hi - lo;
        const int step = (n + WAVE - 1) / WAVE;
        const int chunk_lo = lo + lane * step;          // may exceed hi for trailing lanes
        int q = chunk_lo + step; q = (q < hi ? q : hi) - 1;
        bool pred = false;                               // lanes whose chunk is empty abstain
        if (chunk_lo < hi) pred = (row_end[q] + q + 1 > diagonal);
        const unsigned long long mask = __ballot(pred);
        if (mask == 0ull) { lo = hi; break; }            // no row-end beyond the diagonal in range
        const int f = __ffsll((long long) mask) - 1;     // first chunk whose last element is past
        const int new_lo = lo + f * step;
        int new_hi = new_lo + step; new_hi = (new_hi < hi ? new_hi : hi) - 1;
        lo = new_lo; hi = new_hi;                        // answer in [new_lo, q_f]
    }
    Coord c; c.x = lo < rows ? lo : rows; c.y = diagonal - lo;
    return c;
}

// ref: DeviceSpmvSearchKernel, dispatch_spmv_orig.cuh:104-143 (there: one
// thread per boundary, binary search).  coords has num_tiles+1 entries.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void search_kernel(const int *__restrict__ row_end, int rows, int nnz,
                                                       int tile_items, int num_tiles, Coord *__restrict__ coords)
{
    const int wave_in_block = threadIdx.x / WAVE;
    const int boundary = blockIdx.x * (BLOCK / WAVE) + wave_in_block;
    if (boundary > num_tiles) return;
    const long long total = (long long) rows + nnz;
    long long d = (long long) boundary * tile_items;
    const int diagonal = (int) (d < total ? d : total);
    const Coord c = wave_merge_path_search(diagonal, row_end, rows, nnz);
    if ((threadIdx.x & (WAVE - 1)) == 0) coords[boundary] = c;
}

// ---------------------------------------------------------------------------
// wave64 inclusive segmented (reduce-by-key) scan over one (key, value) pair
// per lane.  Keys are non-decreasing across lanes (rows along the merge
// path), so "same key" == "same segment"; the combine is the reference's
// ReduceByKeyOp<Sum> (thread_operators.cuh:291-301).
// ---------------------------------------------------------------------------
template <typename V>
__device__ __forceinline__ V wave_segmented_inclusive_sum(int key, V val)
{
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const int k2 = __shfl_up(key, d, WAVE);
        const V v2 = __shfl_up(val, d, WAVE);
        if (lane >= d && k2 == key) val += v2;
    }
    return val;
}

// Block-wide exclusive reduce-by-key scan of one pair per thread.
//   in : (key, val) of this thread, keys non-decreasing with threadIdx.x
//   out: carry_in  = sum of val over preceding threads whose key equals the
//                    key of the thread just before this one, restricted to
//                    that trailing equal-key run (0 for thread 0);
//        returns the block aggregate (key, sum) in agg_* for the LAST thread.
// s_wave_key/s_wave_val: LDS scratch, one slot per wave.
template <typename V, int BLOCK>
__device__ __forceinline__ void block_exclusive_rbk(int key, V val, int *s_wave_key, V *s_wave_val,
                                                    int &prev_key, V &carry_in, int &agg_key, V &agg_val)
{
    constexpr int NW = BLOCK / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const V incl = wave_segmented_inclusive_sum<V>(key, val);
    if (lane == WAVE - 1) { s_wave_key[wave] = key; s_wave_val[wave] = incl; }
    __syncthreads();
    // fold the aggregates of the preceding waves, in order
    int pk = -1; V pv = 0; bool have = false;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
        if (w < wave) {
            const int k = s_wave_key[w]; const V v = s_wave_val[w];
            pv = (have && pk == k) ? pv + v : v; pk = k; have = true;
        }
    }
    // exclusive within the wave
    int ek = __shfl_up(key, 1, WAVE);
    V ev = __shfl_up(incl, 1, WAVE);
    if (lane == 0) { prev_key = pk; carry_in = have ? pv : (V) 0; }
    else { prev_key = ek; carry_in = (have && pk == ek) ? pv + ev : ev; }
    // aggregate as seen by this thread (meaningful for the last thread)
    agg_key = key; agg_val = (have && pk == key) ? pv + incl : incl;
}

// ---------------------------------------------------------------------------
// The tile kernel.  ref: DeviceSpmvKernel / AgentSpmv::ConsumeTile,
// dispatch_spmv_orig.cuh:157-186, agent_spmv_orig.cuh:413-639,856-914.
// Everything inside a tile is tile-relative (row 0 == coords[tile].x,
// nonzero 0 == coords[tile].y), so all indices are small.
//   LDS: s_end[r]  = tile-relative nonzero index where tile row r ends, for
//                    the tile_rows rows that end in the tile, then a +inf
//                    sentinel for the row left open (the reference instead
//                    loads row_end[tile_rows] -- one past the array for the
//                    last tile);
//        s_prod[j] = values[j] * x[cols[j]] for the tile's nonzeros.
// ---------------------------------------------------------------------------
template <typename V, int BLOCK, int IPT, bool AXPBY, bool XCD_REMAP>
__global__ __launch_bounds__(BLOCK) void tile_kernel(Params<V> p, const Coord *__restrict__ coords,
                                                     Carry<V> *__restrict__ carries, int num_tiles)
{
    constexpr int TILE = BLOCK * IPT;
    constexpr int NW = BLOCK / WAVE;
    __shared__ int s_end[TILE + 1];
    __shared__ V s_prod[TILE];
    __shared__ int s_wave_key[NW];
    __shared__ V s_wave_val[NW];

    int tile = blockIdx.x;
    if (XCD_REMAP) {
        // Blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8);
        // give each XCD (private 4 MiB L2) a contiguous range of tiles so that
        // neighbouring tiles' x / row-offset lines share one L2.  Bijective
        // for any num_tiles.  Placement only affects speed.
        const int q = num_tiles / 8, r = num_tiles % 8;
        const int xcd = tile % 8, idx = tile / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tid = threadIdx.x;

    const Coord c0 = coords[tile];
    const Coord c1 = coords[tile + 1];
    const int tile_rows = c1.x - c0.x;
    const int tile_nnz = c1.y - c0.y;
    const int tile_items = tile_rows + tile_nnz;

    // ---- stream the tile's nonzeros (coalesced), gather x, stage products
    const int *__restrict__ cols = p.cols + c0.y;
    const V *__restrict__ vals = p.values + c0.y;
    int col_r[IPT];
    V val_r[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * BLOCK;
        if (j < tile_nnz) { col_r[k] = cols[j]; val_r[k] = vals[j]; }
    }
    // ---- row-end offsets of the rows ending in this tile
    const int *__restrict__ row_end = p.row_end + c0.x;
    for (int r = tid; r < tile_rows; r += BLOCK) s_end[r] = row_end[r] - c0.y;
    if (tid == 0) s_end[tile_rows] = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * BLOCK;
        if (j < tile_nnz) s_prod[j] = val_r[k] * p.x[col_r[k]];
    }
    __syncthreads();

    // ---- per-thread merge-path search inside the tile (LDS), diagonal tid*IPT
    int diag = tid * IPT; diag = diag < tile_items ? diag : tile_items;
    int lo = diag - tile_nnz; lo = lo < 0 ? 0 : lo;
    int hi = diag < tile_rows ? diag : tile_rows;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_end[mid] <= diag - mid - 1) lo = mid + 1; else hi = mid;
    }
    int row = lo;            // tile-relative row this thread starts in
    int nz = diag - lo;      // tile-relative nonzero it starts at
    int n_items = tile_items - diag; n_items = n_items < IPT ? n_items : IPT;

    // ---- walk IPT path items
    V *__restrict__ y = p.y + c0.x;
    int cur_end = s_end[row];
    V total = 0;
    int first_row = -1;      // first row this thread completes (needs the carry-in)
    V first_total = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (k < n_items) {
            if (nz < cur_end) {
                total += s_prod[nz];
                ++nz;
            } else {
                if (first_row < 0) { first_row = row; first_total = total; }
                else if (AXPBY) y[row] = p.alpha * total + (p.beta == (V) 0 ? (V) 0 : p.beta * y[row]);
                else y[row] = total;
                total = 0;
                ++row;
                cur_end = s_end[row];
            }
        }
    }

    // ---- carry between threads: block-wide exclusive reduce-by-key scan
    int prev_key, agg_key; V carry_in, agg_val;
    block_exclusive_rbk<V, BLOCK>(row, total, s_wave_key, s_wave_val, prev_key, carry_in, agg_key, agg_val);
    if (first_row >= 0) {
        // prev_key == first_row whenever tid > 0 (the previous thread ended in
        // the row this thread started in); thread 0 has no in-tile carry.
        const V sum = first_total + ((tid > 0 && prev_key == first_row) ? carry_in : (V) 0);
        if (AXPBY) y[first_row] = p.alpha * sum + (p.beta == (V) 0 ? (V) 0 : p.beta * y[first_row]);
        else y[first_row] = sum;
    }
    if (tid == BLOCK - 1) {
        // the row left open at the tile end (ref: agent_spmv_orig.cuh:906-913)
        Carry<V> c; c.key = c0.x + agg_key; c.value = agg_val;
        carries[tile] = c;
    }
}

// ---------------------------------------------------------------------------
// Fix-up: reduce-by-key over n carry pairs whose keys are non-decreasing;
// y[key] += alpha * sum for key < rows (the guard the reference lacks,
// agent_segment_fixup.cuh:257-259).  ref: DeviceSegmentFixupKernel,
// dispatch_spmv_orig.cuh:198-224.
// Each block owns CHUNK = BLOCK*IPT consecutive pairs.  A segment (run of
// equal keys) lying strictly inside the chunk has a unique owner and is
// applied with a plain read-modify-write.  With more than one block, the
// chunk's first and last segments may continue in the neighbours: they are
// written to out[2*chunk], out[2*chunk+1] (keys stay non-decreasing) and the
// kernel is run again on `out`; the final level has a single block.
// Fixed association order => bitwise reproducible results.
// ---------------------------------------------------------------------------
template <typename V, int BLOCK, int IPT>
__global__ __launch_bounds__(BLOCK) void fixup_kernel(const Carry<V> *__restrict__ in, int n,
                                                      Carry<V> *__restrict__ out, V *__restrict__ y, int rows,
                                                      V alpha)
{
    constexpr int CHUNK = BLOCK * IPT;
    constexpr int NW = BLOCK / WAVE;
    __shared__ int s_wave_key[NW];
    __shared__ V s_wave_val[NW];
    const bool multi = gridDim.x > 1;
    const int tid = threadIdx.x;
    const int base = blockIdx.x * CHUNK;
    const int key0 = in[base].key;           // base < n by construction of the grid

    int keys[IPT]; V vals[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int i = base + tid * IPT + k;
        if (i < n) { const Carry<V> c = in[i]; keys[k] = c.key; vals[k] = c.value; }
        else { keys[k] = 0x7fffffff; vals[k] = 0; }
    }
    // Thread-local fold.  A thread conceptually starts inside the segment its
    // predecessor ended in (start_key = key of the item just before its range),
    // so that a segment ending exactly at a thread boundary is closed -- with
    // the carry-in of the preceding threads -- by the thread that follows.
    const int first_i = base + tid * IPT;
    int cur = (tid > 0 && first_i - 1 < n) ? in[first_i - 1].key : keys[0];
    if (tid > 0 && first_i - 1 >= n) cur = 0x7fffffff;
    V total = 0;
    int first_key = -1; V first_total = 0; bool have_first = false;
    auto emit = [&](int key, V sum) {
        if (multi && key == key0) { Carry<V> c; c.key = key; c.value = sum; out[2 * blockIdx.x] = c; }
        else if (key < rows) y[key] += alpha * sum;
    };
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (keys[k] != cur) {
            if (!have_first) { have_first = true; first_key = cur; first_total = total; }
            else emit(cur, total);
            cur = keys[k]; total = vals[k];
        } else total += vals[k];
    }
    int prev_key, agg_key; V carry_in, agg_val;
    block_exclusive_rbk<V, BLOCK>(cur, total, s_wave_key, s_wave_val, prev_key, carry_in, agg_key, agg_val);
    if (have_first) emit(first_key, first_total + ((tid > 0 && prev_key == first_key) ? carry_in : (V) 0));
    // threads whose whole range is one key contributed through the scan only.
    if (tid == BLOCK - 1) {
        if (multi) {
            Carry<V> c; c.key = agg_key; c.value = agg_val; out[2 * blockIdx.x + 1] = c;
            if (agg_key == key0) { Carry<V> z; z.key = key0; z.value = 0; out[2 * blockIdx.x] = z; }
        } else if (agg_key < rows) y[agg_key] += alpha * agg_val;
    }
}

// Single-launch alternative (MSPMV_TUNE_ATOMIC_FIX): one atomicAdd per carry,
// like the reference's fp32 path (agent_segment_fixup.cuh:226-260); order of
// the additions, hence the rounding, varies from run to run.
template <typename V, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fixup_atomic_kernel(const Carry<V> *__restrict__ in, int n,
                                                             V *__restrict__ y, int rows, V alpha)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Carry<V> c = in[i];
    if (c.key < rows && c.value != (V) 0) atomicAdd(&y[c.key], alpha * c.value);
}

// y_local[0] += sum of the carries of earlier parts that belong to it
// (multi-GPU, mspmv.h: mspmv_mg_apply_carries).  take_mask bit j set => add
// carries[j]; rank order => deterministic.
template <typename V>
__global__ void mg_apply_kernel(V *__restrict__ y_local, const V *__restrict__ carries, unsigned long long take_mask,
                                int parts)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    V acc = y_local[0];
    for (int j = 0; j < parts; ++j)
        if ((take_mask >> j) & 1ull) acc += carries[j];
    y_local[0] = acc;
}

}  // namespace mspmv
