// mspmv_kernels.hpp -- hand-written CDNA4 (gfx950, wave64) kernels of the
// merge-based CsrMV.  Replaces the reference's cub/agent/agent_spmv_orig.cuh,
// agent_segment_fixup.cuh and the kernels of
// cub/device/dispatch/dispatch_spmv_orig.cuh:68-224; none of that code (nor
// any CUB/hipCUB/rocPRIM primitive) is used.  The algorithm contract is
// SURVEY.md Appendix B.
//
// What a SpMV runs, all on the caller's stream (mspmv_api.hip picks):
//   ONE launch, the default: tile_kernel_snap -- one 256-thread block per merge tile (XCD-chunked tile order).  The tile's
//                      two boundaries are HINTS read from the caller's temp storage and verified against four row offsets
//                      (wrong: two waves search them -- wave_merge_path_guess -- and store the repaired hints); boundaries
//                      that fall a short way into a row are snapped to the row's first nonzero, so there are no carries
//                      and no fix-up for short rows; rows longer than that hand their pieces over as tagged records
//                      (LookBack).  Inside the tile: 16-byte streaming of (col, val) and row offsets, gather of x (from a
//                      per-block LDS copy when x is <= 4 KB), products and 16-bit tile-relative row ends staged in LDS, one
//                      bit per row start; segmented running sums over 12 consecutive products per thread + one block-wide
//                      DPP segmented scan (consume_tile_flags), y stored per row from registers  (ref: DeviceSpmvKernel);
//                      closed tiles of short rows: a thread per row straight from LDS instead (consume_tile_rows);
//                      fp64 values read non-temporally are fetched line by line over the wave (ld_stream4_linewise)
//   the classic three launches (column-band candidates, the band-major plan's tile order, unaligned arrays, tuning options):
//   1. tile boundaries of the merge path -> coords[tile]            (ref: DeviceSpmvSearchKernel)
//        coords_scatter_kernel : ONE coalesced pass over row_offsets (row end r sits at path position
//                      r + row_end[r]); the default below 10 M rows;
//        coords_interp_kernel  : one THREAD per boundary -- bracket in an LDS table of 1025 samples, secant steps,
//                      exact binary search of what is left; latency-bound, the default from 10 M rows up;
//        search_kernel : one WAVE per boundary, 64-ary search (option)
//      (the same kernels fill in the hints of tile_kernel_snap ahead of time: mspmv_csrmv_prepare)
//   2. tile_kernel_vec : the same tile body on stored coordinates, one (row, partial) carry per tile
//      tile_kernel_vec<.., BAND> : the same kernel with the column-band passes compiled in (run_band_passes): when 64
//                      sampled windows of column indices (band_detect_block, riding on the coordinate launch) say the
//                      columns are spread uniformly over an x several times L2, the first 4-5 blocks per CU stream the
//                      matrix once per band of x instead (C2 fp32: 1.22 -> 0.83 ms); otherwise the ordinary body runs
//      tile_kernel     : dword-per-lane fallback for unaligned arrays, with the reference's
//                      per-thread search + path walk (consume_tile_lds)
//   3. fixup_onepass_kernel : deterministic reduce-by-key over the per-tile carries in one
//                      launch (every run of equal keys has one owner block), y[row] += sum;
//                      no decoupled look-back, no spin-waits, no atomics: bitwise reproducible
//                      (ref: DeviceSegmentFixupKernel).  fixup_kernel / fixup_atomic_kernel:
//                      multi-level and atomic options.
// Development variants of tile_kernel_vec (persistent grid, ablations, cycle stamps) compile only with -DMSPMV_DEV.
// mspmv_spmm.hpp builds the SpMM kernels on the same pieces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <tuple>
#include <utility>

namespace mspmv {

constexpr int WAVE = 64;

typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

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
    int x_lds;                        // > 0: x has this many entries and the tile kernels gather it from LDS
    // column-band passes (BAND kernels only): this launch multiplies the nonzeros whose column lies in
    // [band_lo, band_lo + band_len) and treats the others as zeros; band_pass > 0 adds its carries to the stored ones
    int band_lo, band_len, band_pass;
};

// ---------------------------------------------------------------------------
// Tagged records: how blocks of ONE launch hand each other 64 bits (tile_kernel_snap).  A record is two
// 64-bit words, each carrying half of a per-call tag beside 32 bits of payload, written and read with relaxed agent-scope
// atomics (visible across the XCDs' L2s without a cache flush).  A word is valid when its tag half matches, so no ordering
// between the two is needed, and stale or uninitialised memory is told apart by the 63 tag bits.
// EVERY RECORD SLOT IS CLEAN (0, 0) WHEN THE LAUNCH ENDS -- whoever acts on a slot LAST leaves it clean -- so no record outlives
// its call and a captured call replays with the same tags, even after the matrix changed.  A slot goes through
//     (anything) --publisher ANNOUNCES--> PENDING --publisher stores the record--> RECORD --consumer takes it--> (0, 0)
//   * a publisher ANNOUNCES itself as soon as it knows it will publish -- an atomic exchange of a marker into word 0, issued before
//     the tile's reduction, its answer looked at only when the sum is there: nothing waits for it -- and later stores the record
//     with two plain atomic stores;
//   * the one consumer every record has takes it (both tags match) and clears it;
//   * NOTHING DEPENDS ON ANOTHER WORKGROUP BEING DISPATCHED -- only on workgroups that HAVE STARTED running to their stores (a block that
//     has announced itself and is then preempted for good or faults hangs its consumer, as it would hang any kernel).  Polling for a
//     publisher that has not started is bounded; a consumer whose poll runs out -- the awaited block
//     has not been dispatched: possible only when far fewer blocks are resident than mspmv_api.hip assumed (a CU-masked stream, a
//     long kernel beside this one, several processes on the device) AND the resident ones are all waiting -- CANCELS the slot by an
//     atomic exchange.  What comes back decides: the RECORD (it arrived between the last look and now) -> taken like any other;
//     PENDING -> the publisher is RUNNING (it announced itself); a plain publisher waits for nothing, a group LEADER (below) only --
//     bounded, or for plain publishers that have started -- for its 63 predecessors: a chain of depth two, no cycle, so the record
//     comes in finite time whatever else the device does: the consumer goes on polling for it WITHOUT a bound; anything else -> the publisher has not
//     started: the cancellation stays, the block computes the missing sum itself from the matrix (recompute_row_head), and the
//     publisher, whose announcement brings the cancellation back, wipes the slot instead of publishing.
// (Until round 5 a cancelled slot's late record stayed where it was and an EPOCH word, bumped by the recomputing consumer and mixed
//  into the tags, was to keep a replayed launch from mistaking it for its own.  That left two holes: a publisher dispatched AFTER the
//  bump tagged its record with the NEW epoch -- the one the next replay reads at its start --, and blocks of one launch that read the
//  epoch before and after a bump disagreed about every tag between them, each such pair costing a full poll budget.  A publisher
//  that EXCHANGES its record in and cleans up when it finds a cancellation was measured first: correct, but the answer of the
//  exchange sits on the critical path of a tile that does nothing after publishing -- BASELINE config 4, 23 800 such tiles: +7.5 %.)
// The reference's fp64 fix-up spins without bound on the same residency assumption (single_pass_scan_operators.cuh:620-639);
// here a broken assumption costs time, never the result.  Every interleaving of these operations under relaxed ordering -- one
// publisher and one consumer, and the publisher -> leader -> consumer chain, every poll budget, clean and stale slots -- is
// enumerated by tests/test_record_protocol_model.py (payload or recompute; slot (0, 0) at the end; no wait that nothing ends;
// the unbounded wait only for a publisher that has started).  What the invariant excludes: a slot that already holds THIS call's
// marker when the launch starts (a launch killed between announce and store, then replayed with the same tags).
// ---------------------------------------------------------------------------
constexpr int REC_MAX_POLLS = 1 << 17;      // x ~1 us: ~0.1 s before a consumer gives up on a record and computes the sum itself
// (markers carry the call's tag with its low bit CLEARED: never a valid record tag, tag_a | 1)
__device__ __forceinline__ unsigned long long rec_cancel_word(unsigned tag_a) { return ((unsigned long long) (tag_a & ~1u) << 32) | 0x0CA9CE11ull; }
__device__ __forceinline__ unsigned long long rec_pending_word(unsigned tag_a) { return ((unsigned long long) (tag_a & ~1u) << 32) | 0x9E0D1D61ull; }
// the publisher's announcement; the returned word goes to rec_store when the sum is there
__device__ __forceinline__ unsigned long long rec_announce(unsigned long long *rec, unsigned tag_a)
{
    return __hip_atomic_exchange(rec, rec_pending_word(tag_a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rec_store(unsigned long long *rec, unsigned tag_a, unsigned tag_b, unsigned p0, unsigned p1, unsigned long long announced)
{
    if (announced == rec_cancel_word(tag_a)) {      // the one consumer of this slot gave up before this block started: nobody will take the record
        __hip_atomic_store(rec, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __hip_atomic_store(rec + 1, ((unsigned long long) tag_b << 32) | p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec, ((unsigned long long) tag_a << 32) | p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// waits (bounded) until both words carry this call's tag, clears the record; false = the slot is cancelled (payload undefined)
__device__ __forceinline__ bool rec_take(unsigned long long *rec, unsigned tag_a, unsigned tag_b, unsigned &p0, unsigned &p1, int max_polls)
{
    for (int polls = 0; polls < max_polls; ++polls) {
        const unsigned long long w0 = __hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w1 = __hip_atomic_load(rec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned) (w0 >> 32) == tag_a && (unsigned) (w1 >> 32) == tag_b) {
            __hip_atomic_store(rec, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rec + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p0 = (unsigned) w0; p1 = (unsigned) w1;
            return true;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    // give up on this slot: cancel it -- unless its publisher has at least started
    unsigned long long w0 = __hip_atomic_exchange(rec, rec_cancel_word(tag_a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned) (w0 >> 32) != tag_a && w0 != rec_pending_word(tag_a)) return false;       // not started: the publisher will find the cancellation
    // The record's word 0 (its word 1 was stored before it by the same thread and is on its way), or the announcement of a publisher
    // that is running and waits for nothing (its record overwrites the cancellation): a wait for stores of a block that HAS started
    // (the exchange REPLACED word 0: a record's word 0 is the one it brought back; after an announcement the publisher's store will
    //  overwrite the cancellation)
    const bool have_w0 = (unsigned) (w0 >> 32) == tag_a;
    unsigned long long w1;
    for (;;) {
        if (!have_w0) w0 = __hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w1 = __hip_atomic_load(rec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned) (w0 >> 32) == tag_a && (unsigned) (w1 >> 32) == tag_b) break;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(rec, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p0 = (unsigned) w0; p1 = (unsigned) w1;
    return max_polls > 0;                   // (max_polls == 0, the "never look" testing aid: the slot is cleaned, the record discarded -- every such tile recomputes)
}

// Where a coordinate pass puts what it finds about tile boundary t -- the merge-path point (x, y) of diagonal t * TILE:
//   coords[t] = (x, y)                       what the tile kernels read; mspmv_debug_read_tiles
//   rstart[t] = row_offsets[x]               the first nonzero of the row the boundary falls in (tile_kernel_snap; nullptr: not wanted)
struct BoundaryOut {
    Coord *coords;
    int *rstart;
};
__device__ __forceinline__ void emit_boundary(const BoundaryOut &o, int t, int x, int y, int row_start)
{
    Coord c; c.x = x; c.y = y;
    o.coords[t] = c;
    if (o.rstart) o.rstart[t] = row_start;
}

// A tiny x (the reference's --dense=<cols> inputs: 5 or 32 entries) is copied to LDS once per block and
// gathered there: the x gather then costs LDS reads instead of 12 vector-memory instructions per thread --
// as many as the whole CSR stream needs.  Bounded so that residency barely changes.
constexpr int X_LDS_MAX_BYTES = 4096;

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

// The same search as the self-searching tiles of small problems do it: round 1 probes 64 FIXED samples
// (k + 1) * rows / 64 -- the same addresses for every block, so they come from L2 --, round 2 probes 64 CONSECUTIVE
// rows around the point linear interpolation inside that bracket predicts (two cache lines); on a matrix whose row
// lengths vary smoothly that finishes it: two dependent loads instead of four, ~1.5 us less before a block's first
// nonzero is requested.  Whatever is left of the bracket goes through 64-ary rounds.  Exact; all lanes return the
// same coordinate.  M(p) = row_end[p] + p + 1 (strictly increasing), M(rows) = +inf; the point of diagonal d is
// (p, d - p) for the first p with M(p) > d.
__device__ __forceinline__ Coord wave_merge_path_search_interp(int diagonal, const int *__restrict__ row_end, int rows, int nnz)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int inf = rows + nnz + 1;                              // < 2^31
    auto m_of = [&](int p) { return p < rows ? row_end[p] + p + 1 : inf; };
    const int pk = (int) ((long long) (lane + 1) * rows / WAVE);  // lane 63: rows
    const int mk = m_of(pk);
    unsigned long long mask = __ballot(mk > diagonal);            // monotone; lane 63 is always set
    int f = __ffsll((long long) mask) - 1;
    int b = __shfl(pk, f, WAVE);
    const int m_hi = __shfl(mk, f, WAVE);
    const int p_lo = f == 0 ? -1 : __shfl(pk, f - 1, WAVE);
    const int m_lo = f == 0 ? 0 : __shfl(mk, f - 1, WAVE);
    int a = p_lo + 1;                                             // M(p) <= diagonal for p < a;  M(b) > diagonal
    if (a < b) {
        const long long g = p_lo + 1 + (long long) ((double) (diagonal - m_lo) * (double) (b - p_lo - 1) / (double) (m_hi - m_lo));
        int w0 = (int) g - WAVE / 2;
        const int w_max = b - (WAVE - 1) > a ? b - (WAVE - 1) : a;
        w0 = w0 < a ? a : w0 > w_max ? w_max : w0;               // window [w0, w0 + 63], inside [a, b] where possible
        const int p = w0 + lane;
        const bool pred = p >= b ? true : m_of(p) > diagonal;
        mask = __ballot(pred);
        f = __ffsll((long long) mask) - 1;
        if (mask == 0ull) a = w0 + WAVE;                          // the answer lies beyond the window
        else if (f == 0 && w0 > a) b = w0;                        // ... at or before its first row
        else a = b = w0 + f;                                      // found
        while (a < b) {                                           // 64-ary rounds over [a, b), M(b) > diagonal
            const int n = b - a;
            const int step = (n + WAVE - 1) / WAVE;
            const int chunk_lo = a + lane * step;
            int q = chunk_lo + step; q = (q < b ? q : b) - 1;
            bool pr = false;
            if (chunk_lo < b) pr = m_of(q) > diagonal;
            const unsigned long long mm = __ballot(pr);
            if (mm == 0ull) { a = b; break; }
            const int ff = __ffsll((long long) mm) - 1;
            const int na = a + ff * step;
            int nb = na + step; nb = (nb < b ? nb : b) - 1;
            a = na; b = nb;
        }
    }
    Coord c; c.x = b < rows ? b : rows; c.y = diagonal - b;
    return c;
}

// ref: DeviceSpmvSearchKernel, dispatch_spmv_orig.cuh:104-143 (there: one
// thread per boundary, binary search).  coords has num_tiles+1 entries.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void search_kernel(const int *__restrict__ row_end, int rows, int nnz,
                                                       int tile_items, int num_tiles, BoundaryOut out)
{
    const int wave_in_block = threadIdx.x / WAVE;
    const int boundary = blockIdx.x * (BLOCK / WAVE) + wave_in_block;
    if (boundary > num_tiles) return;
    const long long total = (long long) rows + nnz;
    long long d = (long long) boundary * tile_items;
    const int diagonal = (int) (d < total ? d : total);
    const Coord c = wave_merge_path_search(diagonal, row_end, rows, nnz);
    if ((threadIdx.x & (WAVE - 1)) == 0) emit_boundary(out, boundary, c.x, c.y, c.x > 0 ? row_end[c.x - 1] : 0);
}

// ---------------------------------------------------------------------------
// Column-band passes (no counterpart in the reference).  A matrix whose columns are spread
// uniformly over an x several times the 4 MiB of an XCD's L2 is gather-bound at the Infinity-Cache rate
// (C2: 100 M gathers in 1.2 ms, the CSR stream alone takes 0.14 ms).  Streaming the matrix B times, each
// pass multiplying only the nonzeros of one column band -- an x slice that stays in L2 -- is faster: C2
// fp32 0.83 ms with B = 3.  Whether the columns ARE spread like that is not known to the host, and a
// stateless, asynchronous call cannot wait for the answer: 64 extra blocks of the coordinate pass sample 64
// windows of 2048 consecutive nonzeros, and the tile kernel (its BAND variant) reads the 64 verdicts and runs
// either its ordinary body or the passes (run_band_passes).
// ---------------------------------------------------------------------------
constexpr int BAND_WINDOWS = WAVE;         // one verdict per lane of the wave that reads them
constexpr int BAND_WINDOW = 2048;          // consecutive nonzeros per window
constexpr int BAND_MAJORITY = 56;          // windows that must look uniform
constexpr int BAND_BITMAP_WORDS = 2048;    // 65 536 bits, one per 128-byte line of x modulo 8 MiB
// set bits expected from d distinct lines hashed uniformly: m (1 - exp(-d / m)); d = 0.93 * 2048 -> 1877
constexpr int BAND_MIN_BITS = 1877;
constexpr int BAND_COUNTER_STRIDE = 64;    // ints between the 8 claim counters of run_band_passes (they follow the verdicts)

__device__ __forceinline__ bool band_mode(const int *__restrict__ verdict)      // wave-uniform
{
    const int v = verdict[threadIdx.x & (WAVE - 1)];
    return __popcll(__ballot(v != 0)) >= BAND_MAJORITY;
}

struct BandDetectArgs {
    const int *cols; int nnz, num_cols;
    int line_shift;            // column index -> 128-byte line of x: 5 (fp32), 4 (fp64)
    int *verdict;              // BAND_WINDOWS verdicts, then the claim counters
};

// verdict[w] = 1 when window w touches (almost) as many distinct 128-byte lines of x as it has nonzeros -- no
// short-range reuse for a cache to exploit: R-MAT windows reach 0.6-0.88 of that, stencils and bands a few percent,
// uniform columns 0.975-0.99 -- AND spans at least 3/4 of the columns.  Block `w` of BAND_WINDOWS; BLOCK threads.
template <int BLOCK>
__device__ __forceinline__ void band_detect_block(const BandDetectArgs &a, int w)
{
    constexpr int NW = BLOCK / WAVE;
    __shared__ unsigned s_bits[BAND_BITMAP_WORDS];
    __shared__ int s_red[3][NW];
    const int tid = threadIdx.x;
    for (int i = tid; i < BAND_BITMAP_WORDS; i += BLOCK) s_bits[i] = 0u;
    __syncthreads();
    const long long start = a.nnz > BAND_WINDOW ? (long long) w * (a.nnz - BAND_WINDOW) / (BAND_WINDOWS - 1) : 0;
    const int len = a.nnz < BAND_WINDOW ? a.nnz : BAND_WINDOW;
    int fresh = 0, lo = 0x7fffffff, hi = -1;
    for (int j = tid; j < len; j += BLOCK) {
        const int c = a.cols[start + j];
        lo = c < lo ? c : lo; hi = c > hi ? c : hi;
        const unsigned line = ((unsigned) c >> a.line_shift) & (BAND_BITMAP_WORDS * 32u - 1u);
        const unsigned bit = 1u << (line & 31u);
        fresh += (atomicOr(&s_bits[line >> 5], bit) & bit) ? 0 : 1;
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        fresh += __shfl_xor(fresh, d, WAVE);
        const int l2 = __shfl_xor(lo, d, WAVE), h2 = __shfl_xor(hi, d, WAVE);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((tid & (WAVE - 1)) == 0) { s_red[0][tid / WAVE] = fresh; s_red[1][tid / WAVE] = lo; s_red[2][tid / WAVE] = hi; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < NW; ++k) {
            fresh += s_red[0][k];
            lo = s_red[1][k] < lo ? s_red[1][k] : lo; hi = s_red[2][k] > hi ? s_red[2][k] : hi;
        }
        const bool wide = 4LL * ((long long) hi - lo) >= 3LL * a.num_cols;
        a.verdict[w] = (len == BAND_WINDOW && fresh >= BAND_MIN_BITS && wide) ? 1 : 0;
        if (w < 8) a.verdict[BAND_WINDOWS + w * BAND_COUNTER_STRIDE] = 0;        // the claim counters of run_band_passes
    }
}

// stand-alone form (calls that do not run the scatter coordinate pass: prepared calls, >= 10 M rows)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void band_detect_kernel(BandDetectArgs a)
{
    band_detect_block<BLOCK>(a, (int) blockIdx.x);
}

// ---------------------------------------------------------------------------
// The same coordinates by ONE coalesced pass over the row offsets instead of
// num_tiles independent searches (each 4 dependent, scattered loads): the
// merge position of row-end r is m_r = r + row_end[r] (r row-ends and
// row_end[r] nonzeros precede it -- ties go to the row end, Appendix B.1), and
// the point of diagonal d has x(d) = #{r : m_r < d}.  So thread r owns exactly
// the tile boundaries d = t*tile_items with m_{r-1} < d <= m_r and writes
// (r, d - r) for them; thread `rows` owns the boundaries past the last row end.
// Usually a thread owns 0 or 1 boundary; a row longer than a tile owns many, and
// a wave then fills one such row's run of boundaries cooperatively (a giant row
// of 2^26 nonzeros owns ~37 000 of them).  Measured on MI355X: 3-15 us where
// the search kernel took 27-52 us.
// ---------------------------------------------------------------------------
template <int BLOCK, int TILE_ITEMS, bool VEC>
__device__ __forceinline__ void coords_scatter_block(const int *__restrict__ row_offsets, int rows, int nnz, int num_tiles,
                                                     const BoundaryOut &out, unsigned block)
{
    // With M(i) = (i - 1) + row_offsets[i] (i >= 1; the merge position of row-end i - 1) and
    // M(0) = -1, row r owns the boundaries t with M(r) < t*TILE_ITEMS <= M(r + 1); row index
    // `rows` owns the ones past M(rows).  A thread takes 4 consecutive r (one 16-byte load).
    // A boundary owned by row r lies INSIDE row r (r row-ends consumed, between row_offsets[r] and
    // row_offsets[r + 1] nonzeros), so the row's first nonzero is o[j] -- what tile_kernel_snap wants to know.
    const int lane = threadIdx.x & (WAVE - 1);
    const long long gid = (long long) block * BLOCK + threadIdx.x;
    const int total = rows + nnz;                              // < 2^31
    const long long base = gid * 4;                            // first r of this thread
    int o[5];                                                  // row_offsets[base .. base + 4]
    if (VEC && base + 4 <= rows) {
        const int4v v = *reinterpret_cast<const int4v *>(row_offsets + base);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; o[4] = row_offsets[base + 4];
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) o[j] = base + j <= rows ? row_offsets[base + j] : 0;
    }
    int t_lo[4], t_hi[4];
    bool any_long = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long r = base + j;
        t_lo[j] = 1; t_hi[j] = 0;
        if (r <= rows) {
            const int m_lo = r == 0 ? -1 : (int) r - 1 + o[j];
            t_lo[j] = m_lo < 0 ? 0 : m_lo / TILE_ITEMS + 1;
            t_hi[j] = r < rows ? ((int) r + o[j + 1]) / TILE_ITEMS : num_tiles;
            const int count = t_hi[j] - t_lo[j] + 1;
            if (count > 0 && count <= 2) {
                for (int t = t_lo[j]; t <= t_hi[j]; ++t) {
                    const long long d = (long long) t * TILE_ITEMS;
                    emit_boundary(out, t, (int) r, (int) (d < total ? d : total) - (int) r, o[j]);
                }
            }
            any_long |= count > 2;
        }
    }
    // rows longer than two tiles: the wave fills each such run of boundaries together
    if (__ballot(any_long) == 0ull) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned long long pending = __ballot(t_hi[j] - t_lo[j] + 1 > 2);
        while (pending) {
            const int src = __ffsll((long long) pending) - 1;
            pending &= pending - 1;
            const int lo = __shfl(t_lo[j], src, WAVE), hi = __shfl(t_hi[j], src, WAVE);
            const int rr = (int) __shfl((int) base, src, WAVE) + j;
            const int rs = __shfl(o[j], src, WAVE);
            for (int t = lo + lane; t <= hi; t += WAVE) {
                const long long d = (long long) t * TILE_ITEMS;
                emit_boundary(out, t, rr, (int) (d < total ? d : total) - rr, rs);
            }
        }
    }
}

template <int BLOCK, int TILE_ITEMS, bool VEC, bool DETECT = false>
__global__ __launch_bounds__(BLOCK) void coords_scatter_kernel(const int *__restrict__ row_offsets, int rows, int nnz,
                                                               int num_tiles, BoundaryOut out, BandDetectArgs da)
{
    // DETECT: the first BAND_WINDOWS blocks of the grid sample the column windows (column-band passes, above) -- first, so
    // that their two dependent memory round trips run under the coordinate work instead of after it
    if constexpr (DETECT) {
        if ((int) blockIdx.x < BAND_WINDOWS) { band_detect_block<BLOCK>(da, (int) blockIdx.x); return; }
    }
    coords_scatter_block<BLOCK, TILE_ITEMS, VEC>(row_offsets, rows, nnz, num_tiles, out, DETECT ? blockIdx.x - BAND_WINDOWS : blockIdx.x);
}

// ---------------------------------------------------------------------------
// The same coordinates WITHOUT reading every row offset: one THREAD per tile boundary, interpolation search.
// With M(p) = row_end[p] + p + 1 (the merge position just past row-end p; strictly increasing) and
// M(rows) = +inf, the point of diagonal d is (p, d - p) for the first p in [0, rows] with M(p) > d (for
// p > d or p < d - nnz the predicate holds / fails by itself, so the search bounds of Appendix B.2 are implied).
//   1. every block loads the same K + 1 samples M(k * rows / K) into LDS (L2 hits after the first block);
//   2. a thread finds its bracket among them (10 LDS steps) and runs up to four secant steps -- evaluate M at the
//      linearly interpolated point, shrink the bracket -- then a short walk; on a matrix whose row lengths vary
//      smoothly (grids, bands, dense blocks, uniform random rows) that ends within a row or two of the first guess;
//   3. otherwise (a giant row, a power-law neighbourhood) an exact binary search over what is left of the bracket.
// Always exact -- the result is bit for bit the scatter pass's.  The kernel is latency-bound: its time is that of
// its slowest thread, ~2-8 us on regular matrices and <= 17 us (a full binary search of a bracket) on any matrix,
// whatever the row count, where the scatter pass reads all of row_offsets: 8 us at 3 M rows, 12 at 8 M, 20-23 at
// 16.8 M.  The dispatcher therefore uses it from 10 M rows up (measured: band5 19.1 -> 7.5 us, C4 22.7 -> 11.5,
// grid2d-4096 20.3 -> 16.5; below that size the scatter pass is at least as fast on irregular matrices).
// ---------------------------------------------------------------------------
constexpr int INTERP_SAMPLES = 1024;
// s_m: SAMPLES + 1 ints of LDS (M values fit 32 bits: rows + nnz + 1 < 2^31); has a barrier inside
template <int BLOCK, int SAMPLES = INTERP_SAMPLES>
__device__ __forceinline__ void coords_interp_block(const int *__restrict__ row_end, int rows, int nnz, int tile_items, int num_tiles,
                                                    const BoundaryOut &out, unsigned block, int *s_m)
{
    static_assert(SAMPLES % BLOCK == 0, "whole rounds of sample loads");
    const int total = rows + nnz;
    auto sample_pos = [&](int k) { return (int) ((long long) k * rows / SAMPLES); };
    auto m_of = [&](int p) { return p < rows ? row_end[p] + p + 1 : total + 1; };
    for (int k = threadIdx.x; k < SAMPLES; k += BLOCK) s_m[k] = m_of(sample_pos(k));
    if (threadIdx.x == 0) s_m[SAMPLES] = total + 1;          // M(rows) = +inf
    __syncthreads();
    const long long boundary_ll = (long long) block * BLOCK + threadIdx.x;
    if (boundary_ll > num_tiles) return;
    const int boundary = (int) boundary_ll;
    long long dl = (long long) boundary * tile_items;
    const int d = (int) (dl < total ? dl : total);
    // bracket: first sample k with M(p_k) > d  (k = SAMPLES always qualifies)
    int klo = 0, khi = SAMPLES;
    while (klo < khi) { const int mid = (klo + khi) >> 1; if (s_m[mid] > d) khi = mid; else klo = mid + 1; }
    int b = sample_pos(klo);                                       // M(b) = m_hi > d
    int m_hi = s_m[klo];
    int p_lo = klo == 0 ? -1 : sample_pos(klo - 1);                // M(p_lo) = m_lo <= d   (p_lo = -1: nothing below)
    int m_lo = klo == 0 ? 0 : s_m[klo - 1];
    // secant steps: evaluate M at the interpolated point, keep the half that holds the answer
    for (int it = 0; it < 4 && b - p_lo > 1; ++it) {
        long long g = p_lo + 1 + (long long) ((double) (d - m_lo) * (double) (b - p_lo - 1) / (double) (m_hi - m_lo));
        const int p = (int) (g <= p_lo ? p_lo + 1 : g >= b ? b - 1 : g);
        const int mp = m_of(p);
        if (mp > d) { b = p; m_hi = mp; } else { p_lo = p; m_lo = mp; }
    }
    // short walk from both ends, then an exact binary search over what is left (nothing, on a smooth matrix)
    int a = p_lo + 1;                                              // M(p) <= d for p < a;  M(b) > d
    for (int step = 0; step < 2 && a < b; ++step) { if (m_of(a) > d) b = a; else ++a; }
    for (int step = 0; step < 2 && a < b; ++step) { if (m_of(b - 1) > d) --b; else a = b; }
    while (a < b) { const int mid = (int) (((long long) a + b) >> 1); if (m_of(mid) > d) b = mid; else a = mid + 1; }
    const int x = b < rows ? b : rows;
    emit_boundary(out, boundary, x, d - b, x > 0 ? row_end[x - 1] : 0);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void coords_interp_kernel(const int *__restrict__ row_end, int rows, int nnz, int tile_items,
                                                              int num_tiles, BoundaryOut out)
{
    __shared__ int s_m[INTERP_SAMPLES + 1];
    coords_interp_block<BLOCK>(row_end, rows, nnz, tile_items, num_tiles, out, blockIdx.x, s_m);
}

// ---------------------------------------------------------------------------
// wave64 inclusive segmented (reduce-by-key) scan over one (key, value) pair
// per lane.  Keys are non-decreasing across lanes (rows along the merge
// path), so "same key" == "same segment"; the combine is the reference's
// ReduceByKeyOp<Sum> (thread_operators.cuh:291-301).
// ---------------------------------------------------------------------------
// Data-parallel-primitive (DPP) moves: cross-lane operands delivered inside the VALU
// instead of through the LDS crossbar (__shfl_up lowers to ds_bpermute_b32, ~100 cycles
// of dependent latency per step; a DPP move costs a VALU slot).  Lanes that receive
// nothing (out of the row for row_shr, rows masked off for the broadcasts) keep `old`.
//   row_shr:n   = 0x110 + n   shift right by n inside each row of 16 lanes
//   row_bcast15 = 0x142       lane 15 of each row -> all lanes of the next row
//   row_bcast31 = 0x143       lane 31 -> lanes 32..63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float src)
{
    return __builtin_bit_cast(float, dpp_move<CTRL, ROW_MASK>(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double old, double src)
{
    const long long o = __builtin_bit_cast(long long, old), v = __builtin_bit_cast(long long, src);
    const int lo = dpp_move<CTRL, ROW_MASK>((int) o, (int) v);
    const int hi = dpp_move<CTRL, ROW_MASK>((int) (o >> 32), (int) (v >> 32));
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned) lo);
}

template <typename V, int CTRL, int ROW_MASK>
__device__ __forceinline__ void rbk_step(int key, V &val)
{
    const int k2 = dpp_move<CTRL, ROW_MASK>(-1, key);       // -1: "no source lane", never equals a key
    const V v2 = dpp_move<CTRL, ROW_MASK>((V) 0, val);
    val += k2 == key ? v2 : (V) 0;
}

template <typename V>
__device__ __forceinline__ V wave_segmented_inclusive_sum(int key, V val)
{
    // keys are >= 0 and non-decreasing across lanes: equal key == same segment
    rbk_step<V, 0x111, 0xf>(key, val);      // row_shr:1
    rbk_step<V, 0x112, 0xf>(key, val);      // row_shr:2
    rbk_step<V, 0x114, 0xf>(key, val);      // row_shr:4
    rbk_step<V, 0x118, 0xf>(key, val);      // row_shr:8   -> inclusive within each 16-lane row
    rbk_step<V, 0x142, 0xa>(key, val);      // row_bcast15 into rows 1 and 3
    rbk_step<V, 0x143, 0xc>(key, val);      // row_bcast31 into rows 2 and 3
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
// Streaming loads.  The nonzero arrays and the row offsets are read exactly
// once per SpMV: load them non-temporally so they do not evict x from the
// XCD's 4 MiB L2 (measured: -7 % time on the 12.5 MB-x gather), 16 bytes per
// lane (a dword-per-lane stream tops out at 4.0 TB/s on MI355X, 16 B/lane at
// 6.4 TB/s).
// ---------------------------------------------------------------------------
// NT = false: ordinary loads.  A matrix that fits the 256 MB Infinity Cache stays there
// between SpMVs of an iterative solver when it is read with ordinary loads (measured +9-10 %
// at 58 MB and 272 MB), while for larger matrices the non-temporal form wins (+3-6 %): the
// dispatcher picks per call from the matrix's byte size.
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <typename T> struct Vec4;
template <> struct Vec4<int> { int4v v; __device__ __forceinline__ int get(int i) const { return v[i]; } };
template <> struct Vec4<float> { float4v v; __device__ __forceinline__ float get(int i) const { return v[i]; } };
template <> struct Vec4<double> { double2v a, b; __device__ __forceinline__ double get(int i) const { return i < 2 ? a[i] : b[i - 2]; } };

template <bool NT>
__device__ __forceinline__ Vec4<int> ld_stream4(const int *p)
{
    Vec4<int> r; const int4v *q = reinterpret_cast<const int4v *>(p);
    r.v = NT ? __builtin_nontemporal_load(q) : *q; return r;
}
template <bool NT>
__device__ __forceinline__ Vec4<float> ld_stream4(const float *p)
{
    Vec4<float> r; const float4v *q = reinterpret_cast<const float4v *>(p);
    r.v = NT ? __builtin_nontemporal_load(q) : *q; return r;
}
template <bool NT>
__device__ __forceinline__ Vec4<double> ld_stream4(const double *p)
{
    Vec4<double> r; const double2v *q = reinterpret_cast<const double2v *>(p);
    r.a = NT ? __builtin_nontemporal_load(q) : q[0];
    r.b = NT ? __builtin_nontemporal_load(q + 1) : q[1];
    return r;
}
__device__ __forceinline__ void zero4(Vec4<float> &r) { r.v = float4v{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void zero4(Vec4<double> &r) { r.a = double2v{0.0, 0.0}; r.b = r.a; }
__device__ __forceinline__ void st_lds4(float *p, const float (&v)[4])
{ float4v w; w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3]; *reinterpret_cast<float4v *>(p) = w; }
__device__ __forceinline__ void st_lds4(double *p, const double (&v)[4])
{
    double2v a, b; a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
    *reinterpret_cast<double2v *>(p) = a; *(reinterpret_cast<double2v *>(p) + 1) = b;
}
__device__ __forceinline__ void st_lds4(int *p, const int (&v)[4])
{ int4v w; w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3]; *reinterpret_cast<int4v *>(p) = w; }
// Tile-relative row ends as the flag/scan reduction keeps them: 16 bits each (a tile has < 65536
// nonzeros; entries of rows outside the tile are never read, so truncating them is harmless).
// LDS is what limits the resident blocks per CU, and residency is worth more than anything else
// here: 6 KB per 256x11 tile.
typedef unsigned short end16_t;
__device__ __forceinline__ void st_lds4(end16_t *p, const int (&v)[4])
{
    int2v w;
    w.x = (v[0] & 0xffff) | (v[1] << 16);
    w.y = (v[2] & 0xffff) | (v[3] << 16);
    *reinterpret_cast<int2v *>(p) = w;
}
template <bool FL> struct EndType { typedef int type; };
template <> struct EndType<true> { typedef end16_t type; };

// LDS index swizzle of the staged products.  Thread t of the walk reads products around index
// t*IPT - (rows before it); when rows have a regular length the per-lane stride resonates with
// the 32 banks (dense 32-nnz rows, IPT = 11: lanes l, l+3, l+6 ... hit one bank, an 11-way
// conflict: SQ_LDS_BANK_CONFLICT was 56 % of all LDS cycles).  XOR-ing index bits 2..4 with
// bits 5..7 keeps every 4-element chunk contiguous and 16-byte aligned (the staging writes
// whole chunks) and moves indices 32 apart to different banks.
__device__ __forceinline__ int swz_prod(int i) { return i ^ (((i >> 5) & 7) << 2); }

// ---------------------------------------------------------------------------
// The reference's in-tile algorithm (agent_spmv_orig.cuh:539-634,906-913), tuned for CDNA4:
// per-thread merge-path search on diagonal tid*IPT, the IPT-step path walk, the block-wide
// carry scan, the y stores and the tile's carry-out.  Used by the dword-per-lane fallback
// kernel (tile_kernel) and selectable in tile_kernel_vec for comparison (tuning flag 0x70000);
// the production kernels use consume_tile_flags below.  Everything is tile-relative
// (row 0 == c0.x, nonzero 0 == c0.y).
//   s_end[r]  = tile-relative nonzero index where tile row r ends (r <
//               tile_rows), +inf for the row left open at the tile end (the
//               reference instead loads row_end[tile_rows] -- one past the
//               array for the last tile);
//   s_prod[j] = values[j] * x[cols[j]] for the tile's nonzeros.
// ---------------------------------------------------------------------------
template <typename V, int BLOCK, int IPT, bool AXPBY, bool SWZ = false>
__device__ __forceinline__ void consume_tile_lds(const Params<V> &p, const Coord c0, int tile_rows, int tile_nnz,
                                                 const int *s_end, const V *s_prod, V *s_y, int *s_wave_key,
                                                 V *s_wave_val, Carry<V> *__restrict__ carry_out, int pshift = 0)
{
    // s_y (tile_rows entries) may alias the memory behind s_prod.  SWZ: s_prod is the raw
    // (unshifted) array, product j lives at swz_prod(pshift + j).
    // Straight-line code throughout: divergent branches in the search and the walk made
    // every wave execute both sides of each step (measured 15-23 % of the kernel on
    // short-row matrices).  Requirements on the staging: s_end[r] = +inf for every
    // r >= tile_rows and s_prod[j] = 0 for every j >= tile_nnz, up to TILE + IPT + 3.
    const int tid = threadIdx.x;
    const int tile_items = tile_rows + tile_nnz;
    int diag = tid * IPT; diag = diag < tile_items ? diag : tile_items;

    // ---- in-tile merge-path search: x(diag) = #{r < tile_rows : s_end[r] + r < diag}
    // (the predicate is monotone in r, Appendix B.2; the bounds max(diag - nnz, 0) and
    // min(diag, rows) of the reference's search are implied by it).  Uniform trip count.
    int row = 0;
    for (int step = 1 << (31 - __builtin_clz(tile_rows | 1)); step > 0; step >>= 1) {
        const int q = row + step;                      // candidate count
        const int e = s_end[q - 1 < tile_rows ? q - 1 : tile_rows];   // clamped: slot tile_rows holds +inf
        row = (q <= tile_rows && e + (q - 1) < diag) ? q : row;
    }
    const int first_row = row;                         // the row this thread starts in
    const int nz = diag - row;

    // ---- walk IPT path items, latency-free form.
    // The reference's walk (agent_spmv_orig.cuh:557-578) reads LDS once per step with the
    // address depending on the previous step: IPT dependent LDS round trips.  Here the
    // <= IPT row ends this thread can reach are read in one batch; row-end j sits at path
    // item (s_end[first_row + j] - nz) + j of this thread, which gives a bit mask of the
    // items that are row ends; the k-th item is then nonzero nz + k - popcount(mask below
    // k), so all products are read in a second batch and the segmented sum runs in registers.
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int pos = s_end[row + j] - nz + j;       // +inf sentinel (2^30) stays out of range
        mask |= pos < IPT ? 1u << pos : 0u;
    }
    V prod[IPT];
    {
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            prod[k] = SWZ ? s_prod[swz_prod(pshift + nz + k - cnt)] : s_prod[nz + k - cnt];
            cnt += (mask >> k) & 1u;
        }
    }
    // segmented sum in registers: ended[k] = total of the row that ends at item k
    V ended[IPT];
    V total = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const bool end = (mask >> k) & 1u;
        ended[k] = total;
        total = end ? (V) 0 : total + prod[k];
    }
    const int done_all = __builtin_popcount(mask);
    // ---- carry between threads: block-wide exclusive reduce-by-key scan (has a barrier
    //      inside: after it every thread has finished reading s_prod, which s_y aliases)
    int prev_key, agg_key; V carry_in, agg_val;
    block_exclusive_rbk<V, BLOCK>(row + done_all, total, s_wave_key, s_wave_val, prev_key, carry_in, agg_key, agg_val);
    // ---- row totals -> LDS (scattered 4/8-byte LDS writes are cheap; scattered global
    //      stores were not: ~1.3 rows per L2 write request, as many requests as the
    //      whole read stream on a 5-nnz/row matrix), then one coalesced copy to y.
    {
        // prev_key == first_row whenever tid > 0 (the previous thread ended in the row this
        // thread started in); thread 0 has no in-tile carry.
        const V first_carry = (tid > 0 && prev_key == first_row) ? carry_in : (V) 0;
        int done = 0;
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            if ((mask >> k) & 1u) {
                s_y[row + done] = done == 0 ? ended[k] + first_carry : ended[k];
                ++done;
            }
        }
    }
    if (tid == BLOCK - 1) {
        // the row left open at the tile end (ref: agent_spmv_orig.cuh:906-913)
        Carry<V> c; c.key = c0.x + agg_key; c.value = agg_val;
        *carry_out = c;
    }
    __syncthreads();
    V *__restrict__ y = p.y + c0.x;
    for (int r = tid; r < tile_rows; r += BLOCK) {
        if (AXPBY) y[r] = p.alpha * s_y[r] + (p.beta == (V) 0 ? (V) 0 : p.beta * y[r]);
        else y[r] = s_y[r];
    }
}

// ---------------------------------------------------------------------------
// Carries without a second launch (tile_kernel_snap, for the rows its boundary snapping leaves open).  The per-tile carries
// are LOCAL partial sums -- no tile needs another tile's result to compute its own -- so the fix-up "y[row] += carries of
// the tiles before the one in which the row ends" can be done by that tile itself.  A tile that leaves a row open PUBLISHES
// its partial sum as soon as its scan is done.  The tile in which the row ENDS knows exactly which earlier tiles hold
// pieces of it: the row's first nonzero is path item r + row_offsets[r], i.e. it lies in tile (r + row_offsets[r]) / TILE,
// so the pieces are the carries of tiles [that tile, this tile).  It takes exactly those records (one lane each: up to 64 by
// wave 0 alone while the other waves run the row phase, longer lists by the whole block), adds them to its first row in a
// fixed order and clears them.  There is no chain of waits (a waiting tile waits for blocks that wait for nobody before
// publishing), and workgroups are dispatched in order: an awaited block is running or done (mspmv_api.hip:
// safe_chunk_log2 keeps that true under the XCD-chunked tile order).  The reference's fp64 fix-up relies on the same
// property (decoupled look-back, agent_segment_fixup.cuh:262-341 with single_pass_scan_operators.cuh); unlike it,
// nothing here spins on a chain -- and when the property does not hold (fewer resident blocks than assumed) the waiting
// tile computes the sum itself after a bounded poll (recompute_row_head).  Records: rec_store / rec_take above.
// ---------------------------------------------------------------------------
struct LookBack {
    unsigned long long *rec;      // 2 words per tile; nullptr = off (the fix-up launch adds the carries)
    unsigned tag_a, tag_b;        // tag_a is never 0 (a cleared word is never valid)
    int *error;                   // two words of the call's temp storage: [0] receives call_tag when a poll ran out and the consumer computed the
                                  // sum itself (a diagnostic: debug_sync reports it), [1] counts such episodes (was: an epoch mixed into the tags)
    unsigned call_tag;            // what the host compares word [0] with
    int max_polls;                // how often a consumer looks for a record before it computes the sum itself (REC_MAX_POLLS; tests: 1, or 0 = never looks)
    int group_base;               // record index of the first GROUP record (= the launch's number of tiles): see LB_GROUP
};
// GROUP RECORDS: a row spanning thousands of tiles (BASELINE config 4: one row of 67 M nonzeros = 23 800 tiles) left its last tile
// 23 800 records to take -- 24 rounds of memory latency by ONE block, 50 us at the end of a 130 us launch while the chip idles.
// So every LB_GROUP-th piece of a long row is a group LEADER: a tile without a row end, tile % LB_GROUP == LB_GROUP - 1, whose row
// began at or before the group's first tile.  The leader takes its LB_GROUP - 1 predecessors' records (one wave, one round), adds its
// own partial sum and publishes the total as group record tile / LB_GROUP -- instead of a record of its own.  The tile in which
// the row ends derives the same set of leaders from the same two numbers (the row's first piece, its own index) and takes the
// group records of the complete groups plus the single records before the first and after the last of them: at most
// 2 * (LB_GROUP - 1) + pieces / LB_GROUP records.  Every record still has exactly one taker, which clears it; the order of the
// additions is fixed by the tile indices alone; and nothing depends on a record arriving -- a leader whose poll runs out
// computes its group's sum from the matrix, like any consumer.
constexpr int LB_GROUP = 64;
template <typename V> struct LbBits;
template <> struct LbBits<float> {
    static __device__ __forceinline__ void split(float v, unsigned &p0, unsigned &p1) { p0 = __builtin_bit_cast(unsigned, v); p1 = 0u; }
    static __device__ __forceinline__ float join(unsigned p0, unsigned) { return __builtin_bit_cast(float, p0); }
};
template <> struct LbBits<double> {
    static __device__ __forceinline__ void split(double v, unsigned &p0, unsigned &p1)
    { const unsigned long long b = __builtin_bit_cast(unsigned long long, v); p0 = (unsigned) (b >> 32); p1 = (unsigned) b; }
    static __device__ __forceinline__ double join(unsigned p0, unsigned p1)
    { return __builtin_bit_cast(double, ((unsigned long long) p0 << 32) | p1); }
};
template <typename V>
__device__ __forceinline__ void lb_publish(const LookBack &lb, int slot, V value, unsigned long long announced)
{
    unsigned p0, p1; LbBits<V>::split(value, p0, p1);
    rec_store(lb.rec + 2 * (size_t) slot, lb.tag_a, lb.tag_b, p0, p1, announced);
}
__device__ __forceinline__ unsigned long long lb_announce(const LookBack &lb, int slot) { return rec_announce(lb.rec + 2 * (size_t) slot, lb.tag_a); }
// the carry tile s published: waits (bounded) until both words carry this call's tag, then clears the record
template <typename V>
__device__ __forceinline__ V lb_take(const LookBack &lb, int s, bool &ok)
{
    unsigned p0, p1;
    if (rec_take(lb.rec + 2 * (size_t) s, lb.tag_a, lb.tag_b, p0, p1, lb.max_polls)) return LbBits<V>::join(p0, p1);
    ok = false;                             // the caller computes the sum itself (recompute_row_head)
    return (V) 0;
}
// Sum of the carries of tiles [tile - count, tile) -- the pieces of this tile's first row held by earlier tiles --
// taken by the lanes of ONE wave (count <= 64), nearest tile on lane 0; fixed butterfly order; wave-uniform result.
template <typename V>
__device__ __forceinline__ V lb_take_wave(const LookBack &lb, int tile, int count, bool &ok)
{
    const int lane = threadIdx.x & (WAVE - 1);
    V part = lane < count ? lb_take<V>(lb, tile - 1 - lane, ok) : (V) 0;
    if (count > 1) {
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) part += __shfl_xor(part, d, WAVE);
    } else part = __shfl(part, 0, WAVE);
    return part;
}
// The same for a long list (a row spanning more than 64 tiles), by the whole block: thread j takes tiles tile-1-j,
// tile-1-j-BLOCK, ... in that order; wave butterflies; the wave partials in wave order.  Block-uniform call (it has a
// barrier); the result is valid on thread 0.
template <typename V, int BLOCK, int U, typename SlotOf>
__device__ __forceinline__ V lb_take_block(const LookBack &lb, int count, V *s_wave_val, bool &ok, SlotOf slot_of)
{   // (slot_of(i): the record index of the i-th of the `count` records to take)
    // U records per thread and round are requested before any is looked at (a round costs one memory latency whatever its
    // width: with U = 4 a row spanning 37 000 tiles is 37 rounds); one that is not there yet -- only ever among the nearest
    // tiles -- is then polled for.  The order of the additions is fixed by (thread, round, slot).  (U is what the register
    // budget of the calling kernel allows.)
    V part = 0;
    for (int i0 = threadIdx.x; i0 < count; i0 += U * BLOCK) {
        unsigned long long w0[U], w1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * BLOCK;
            w0[u] = w1[u] = 0ull;
            if (i < count) {
                unsigned long long *r = lb.rec + 2 * (size_t) slot_of(i);
                w0[u] = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w1[u] = __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * BLOCK;
            if (i < count) {
                if (lb.max_polls > 0 && (unsigned) (w0[u] >> 32) == lb.tag_a && (unsigned) (w1[u] >> 32) == lb.tag_b) {
                    unsigned long long *r = lb.rec + 2 * (size_t) slot_of(i);
                    __hip_atomic_store(r, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(r + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    part += LbBits<V>::join((unsigned) w0[u], (unsigned) w1[u]);
                } else part += lb_take<V>(lb, slot_of(i), ok);
            }
        }
    }
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) part += __shfl_xor(part, d, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_wave_val[threadIdx.x / WAVE] = part;
    __syncthreads();
    V total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / WAVE; ++w) total += s_wave_val[w];
    return total;
}

// What the records of tiles [first piece, this tile) add up to, computed from the matrix instead: the products of the
// nonzeros [j0, j1) of the row that ends in this tile (from the row's first nonzero to the tile's first) -- the path of a
// consumer whose poll ran out.  Whole block (barriers inside); fixed order; the result on every thread.
template <typename V, int BLOCK>
__device__ __forceinline__ V recompute_row_head(const Params<V> &p, int j0, int j1, V *s_wave_val)
{
    V part = 0;
    for (int j = j0 + (int) threadIdx.x; j < j1; j += BLOCK) part += p.values[j] * p.x[p.cols[j]];
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) part += __shfl_xor(part, d, WAVE);
    __syncthreads();                                   // (s_wave_val may still be read by a take that preceded)
    if ((threadIdx.x & (WAVE - 1)) == 0) s_wave_val[threadIdx.x / WAVE] = part;
    __syncthreads();
    V total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / WAVE; ++w) total += s_wave_val[w];
    return total;
}

// ---------------------------------------------------------------------------
// In-tile reduction of the production kernels: head flags + segmented scan.
//
// The reference walks the tile's merge path thread by thread (a binary search per
// thread, then ITEMS dependent LDS reads, agent_spmv_orig.cuh:539-578); consume_tile_lds
// above is that algorithm.  On CDNA4 it costs ~100 instructions per path item, which is what
// bounds a streaming CsrMV here (39 lane-operations per fp32 nonzero are available at the HBM
// peak).  The same tile result is obtained with no per-thread search and no dependent walk:
//   * staging sets one bit per row START in a bit array over the tile's nonzeros (row r ends
//     at tile-relative nonzero e => the next row starts at e: bit e, when e < tile_nnz; several
//     empty rows set the same bit), plus the bit of the tile's first nonzero;
//   * NONZERO PHASE: thread t owns the NPT consecutive staged products [t*NPT, (t+1)*NPT) (16-byte
//     LDS reads), runs the segmented inclusive sum over them in registers, one block-wide
//     segmented scan of (has-flag, running-sum) supplies the sum flowing in from the threads
//     before it, and S[j] = "sum of the row containing nonzero j, up to j" is written back in place;
//   * ROW PHASE: thread r of the tile's rows reads S at its row's last nonzero (0 for an empty
//     row) and stores y[r] straight from registers, coalesced.  The row left open at the tile
//     end is S at the tile's last nonzero: the carry.
// Path items still bound a tile (rows + nonzeros <= BLOCK*IPT), tiles/carries/fix-up are
// unchanged; only the association order inside a tile differs (sequential inside a thread's
// NPT nonzeros, scan tree across threads), as it already did from the reference's.
// Raw LDS positions outside [pshift, pshift + tile_nnz) may hold anything (the interior
// staging path does not clear them): the bit at pshift isolates what precedes the tile and
// nothing reads S past the tile's last nonzero.
// ---------------------------------------------------------------------------
// 16-byte units of the staged products are XOR-swizzled when a thread of the nonzero phase
// owns 2 or 4 of them (lanes 8 or 4 apart would otherwise start in the same banks).
template <typename V, int CPT>
__device__ __forceinline__ int prod_unit(int u)
{
    constexpr int UPT = CPT * 4 * (int) sizeof(V) / 16;
    // UPT == 6 (fp64, 12 products per thread): lanes 8 apart start 48 units apart = the same banks
    return UPT == 4 ? u ^ ((u >> 4) & 3) : UPT == 2 ? u ^ ((u >> 4) & 1) : UPT == 6 ? u ^ ((u / 48) & 1) : u;
}
template <typename V, int CPT>
__device__ __forceinline__ int prod_slot(int e)      // element index (>= 0) in raw order -> LDS element index
{
    constexpr int EPU = 16 / (int) sizeof(V);
    return prod_unit<V, CPT>(e / EPU) * EPU + (e % EPU);
}
// (plain: the lean row phase of short-row tiles, consume_tile_rows, reads the products of a row at consecutive addresses)
template <typename V, int CPT>
__device__ __forceinline__ int prod_unit(int u, bool plain) { return plain ? u : prod_unit<V, CPT>(u); }
template <typename V, int CPT>
__device__ __forceinline__ int prod_slot(int e, bool plain) { return plain ? e : prod_slot<V, CPT>(e); }
template <int CPT>
__device__ __forceinline__ void st_prod_chunk(float *s_prod_raw, int chunk, const float (&v)[4], bool plain = false)
{
    st_lds4(&s_prod_raw[4 * prod_unit<float, CPT>(chunk, plain)], v);
}
template <int CPT>
__device__ __forceinline__ void st_prod_chunk(double *s_prod_raw, int chunk, const double (&v)[4], bool plain = false)
{
    double2v a, b; a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
    *reinterpret_cast<double2v *>(&s_prod_raw[2 * prod_unit<double, CPT>(2 * chunk, plain)]) = a;
    *reinterpret_cast<double2v *>(&s_prod_raw[2 * prod_unit<double, CPT>(2 * chunk + 1, plain)]) = b;
}
__device__ __forceinline__ void ld_unit(const float *p, float *out)
{ const float4v w = *reinterpret_cast<const float4v *>(p); out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = w.w; }
__device__ __forceinline__ void ld_unit(const double *p, double *out)
{ const double2v w = *reinterpret_cast<const double2v *>(p); out[0] = w.x; out[1] = w.y; }
__device__ __forceinline__ void st_unit(float *p, const float *in)
{ float4v w; w.x = in[0]; w.y = in[1]; w.z = in[2]; w.w = in[3]; *reinterpret_cast<float4v *>(p) = w; }
__device__ __forceinline__ void st_unit(double *p, const double *in)
{ double2v w; w.x = in[0]; w.y = in[1]; *reinterpret_cast<double2v *>(p) = w; }

// one Kogge-Stone step of the wave's inclusive segmented sum over (flag, value) pairs
template <typename V, int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step(int &f, V &v)
{
    const int f2 = dpp_move<CTRL, ROW_MASK>(0, f);
    const V v2 = dpp_move<CTRL, ROW_MASK>((V) 0, v);
    v += f ? (V) 0 : v2;
    f |= f2;
}
// Block-wide EXCLUSIVE segmented sum: returns the sum of `val` over the threads before this
// one back to (and including) the nearest preceding thread with flag set.  One barrier inside.
template <typename V, int BLOCK>
__device__ __forceinline__ V block_exclusive_segsum(bool flag, V val, int *s_wave_flag, V *s_wave_val, int tid_in = -1)
{
    constexpr int NW = BLOCK / WAVE;
    const unsigned tidu = tid_in < 0 ? threadIdx.x : (unsigned) tid_in;       // (tid_in: see tile_kernel_band)
    const int lane = tidu & (WAVE - 1);
    const int wave = tidu / WAVE;
    int f = flag ? 1 : 0;
    V v = val;
    seg_step<V, 0x111, 0xf>(f, v);      // row_shr:1
    seg_step<V, 0x112, 0xf>(f, v);      // row_shr:2
    seg_step<V, 0x114, 0xf>(f, v);      // row_shr:4
    seg_step<V, 0x118, 0xf>(f, v);      // row_shr:8
    seg_step<V, 0x142, 0xa>(f, v);      // row_bcast15 into rows 1 and 3
    seg_step<V, 0x143, 0xc>(f, v);      // row_bcast31 into rows 2 and 3
    if (lane == WAVE - 1) { s_wave_flag[wave] = f; s_wave_val[wave] = v; }
    __syncthreads();
    V pv = 0;                           // segmented sum of the preceding waves, in order
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
        if (w < wave) { const V wv = s_wave_val[w]; pv = s_wave_flag[w] ? wv : pv + wv; }
    }
    const int ef = __shfl_up(f, 1, WAVE);
    const V ev = __shfl_up(v, 1, WAVE);
    return lane == 0 ? pv : (ef ? ev : pv + ev);
}

template <typename V, int BLOCK, int IPT, bool AXPBY, int LB_BATCH = 1>
__device__ __forceinline__ void consume_tile_flags(const Params<V> &p, const Coord c0, int tile_rows, int tile_nnz,
                                                   const end16_t *s_end, V *s_prod_raw, unsigned *s_flag,
                                                   int *s_wave_flag, V *s_wave_val, Carry<V> *__restrict__ carry_out,
                                                   int pshift, unsigned long long *tr = nullptr, const LookBack *lb = nullptr,
                                                   int tile = 0, bool publish = false, int first_row_tile = 0, int tid_in = -1,
                                                   int first_row_start = 0)
{
    constexpr int TAKE_BATCH = LB_BATCH;
    constexpr int CPT = IPT / 4 + 1;
    constexpr int NPT = CPT * 4;                  // staged products per thread
    constexpr int EPU = 16 / (int) sizeof(V);     // elements per 16-byte unit
    constexpr int UPT = NPT / EPU;
    constexpr int FLAG_WORDS = CPT * BLOCK * 4 / 32 + 1;
    static_assert(NPT <= 16 && FLAG_WORDS <= BLOCK, "flag word handling");
    const int tid = tid_in < 0 ? (int) threadIdx.x : tid_in;

    // one-launch path: a tile that will publish ANNOUNCES itself now -- the thread that will store the record exchanges a marker into
    // the slot; the answer is looked at when the sum is there (rec_store), by then it has long arrived ("EVERY RECORD SLOT IS CLEAN")
    // (a group leader -- "GROUP RECORDS"; first_row_tile < tile only when the tile begins inside a row that began in an earlier one --
    //  publishes the group's slot, by thread 0; a piece without a row end its own, by the thread that holds its last nonzero; a tile
    //  with row ends its own, by the last thread)
    const bool leader = lb != nullptr && publish && tile_rows == 0 && tile % LB_GROUP == LB_GROUP - 1 && first_row_tile <= tile - (LB_GROUP - 1);
    unsigned long long announced = 0ull;
    if (lb != nullptr && publish) {
        const int pub_thread = tile_rows > 0 ? BLOCK - 1 : leader ? 0 : tile_nnz > 0 ? (pshift + tile_nnz - 1) / NPT : 0;
        if (tid == pub_thread) announced = lb_announce(*lb, leader ? lb->group_base + tile / LB_GROUP : tile);
    }

    // ---- nonzero phase
    V s[NPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) ld_unit(&s_prod_raw[prod_unit<V, CPT>(tid * UPT + u) * EPU], &s[u * EPU]);
    const int base = tid * NPT;
    const unsigned lo = s_flag[base >> 5], hi = s_flag[(base >> 5) + 1];
    const unsigned m = (unsigned) ((((unsigned long long) hi << 32) | lo) >> (base & 31)) & ((1u << NPT) - 1u);
    V run = 0;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        run = ((m >> k) & 1u) ? s[k] : run + s[k];
        s[k] = run;
    }
    if (tr && tid == 0) tr[6] = clock64();
    const V carry_in = block_exclusive_segsum<V, BLOCK>(m != 0u, run, s_wave_flag, s_wave_val, tid_in);
    const unsigned lead = (m & (0u - m)) - 1u;    // bits below the first row start (all ones when none)
#pragma unroll
    for (int k = 0; k < NPT; ++k) s[k] += ((lead >> k) & 1u) ? carry_in : (V) 0;
    if (tile_rows == 0) {
        // block-uniform: no row ends in this tile (a slice of one long row): only the running sum
        // at the tile's last nonzero is needed -- the carry.  No write-back, no third barrier.
        if (tid < FLAG_WORDS) s_flag[tid] = 0u;
        const int last = pshift + tile_nnz - 1;
        if (tile_nnz > 0 ? tid == last / NPT : tid == 0) {
            V v = 0;
#pragma unroll
            for (int k = 0; k < NPT; ++k) v = (tile_nnz > 0 && k == last % NPT) ? s[k] : v;
            Carry<V> c; c.key = c0.x; c.value = v;
            if (p.band_pass > 0) c.value += carry_out->value;      // later column-band pass: same tile, same key
            *carry_out = c;
            if (lb && publish && !leader) lb_publish<V>(*lb, tile, v, announced);   // (only when some tile will take it)
            if (leader) s_wave_val[0] = v;
        }
        if (leader) {
            // (block-uniform) the group's total: the LB_GROUP - 1 records before this tile, by one wave, + this tile's own sum
            __syncthreads();
            const V own = s_wave_val[0];
            bool ok = true;
            V before = 0;
            if (tid < WAVE) before = lb_take_wave<V>(*lb, tile, LB_GROUP - 1, ok);
            if (__syncthreads_or(ok ? 0 : 1)) {
                // a poll ran out: the nonzeros of the group's earlier tiles, from the matrix (they are the row's, one tile's worth each --
                // or from the row's first nonzero, when the group begins with the row's first piece)
                const int j0 = first_row_tile == tile - (LB_GROUP - 1) ? first_row_start : c0.y - (LB_GROUP - 1) * (BLOCK * IPT);
                before = recompute_row_head<V, BLOCK>(p, j0, c0.y, s_wave_val);
                if (tid == 0 && lb->error) {
                    __hip_atomic_store(lb->error, (int) lb->call_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(lb->error + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (tid == 0) lb_publish<V>(*lb, lb->group_base + tile / LB_GROUP, before + own, announced);
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < UPT; ++u) st_unit(&s_prod_raw[prod_unit<V, CPT>(tid * UPT + u) * EPU], &s[u * EPU]);
    __syncthreads();
    if (tr && tid == 0) tr[7] = clock64();
    if (tid < FLAG_WORDS) s_flag[tid] = 0u;       // clean for the next tile's staging (after the loop's barrier)

    // ---- the row left open at the tile end (ref: agent_spmv_orig.cuh:906-913): the carry.  Written -- and, on the
    //      single-launch path, published for the tile in which that row ends -- BEFORE the row phase, as early as it exists
    if (tid == BLOCK - 1) {
        const int e_last = tile_rows > 0 ? s_end[tile_rows - 1] : 0;
        Carry<V> c; c.key = c0.x + tile_rows;
        c.value = tile_nnz > e_last ? s_prod_raw[prod_slot<V, CPT>(pshift + tile_nnz - 1)] : (V) 0;
        if (p.band_pass > 0) c.value += carry_out->value;
        *carry_out = c;
        if (lb && publish) lb_publish<V>(*lb, tile, c.value, announced);
    }
    // single-launch path: the pieces of this tile's first row held by tiles [first_row_tile, tile) (thread 0 stores row 0).
    // Up to 64 of them: wave 0 alone, while the other waves go on with the row phase; more: the whole block.
    V first_row_carry = 0;
    if (lb) {
        const int pieces = tile - first_row_tile;            // block-uniform
        if (pieces > 0) {
            bool ok = true;
            // the complete groups among the pieces [first_row_tile, tile): their leaders folded them into group records
            const int g_first = (first_row_tile + LB_GROUP - 1) / LB_GROUP, g_end = tile / LB_GROUP;
            if (g_first < g_end) {
                const int n_tail = tile - g_end * LB_GROUP, n_groups = g_end - g_first, n_head = g_first * LB_GROUP - first_row_tile;
                const int gb = lb->group_base;
                // nearest first: the single records after the last complete group, the group records, the singles before the first
                first_row_carry = lb_take_block<V, BLOCK, TAKE_BATCH>(*lb, n_tail + n_groups + n_head, s_wave_val, ok, [=](int i) {
                    return i < n_tail ? tile - 1 - i : i < n_tail + n_groups ? gb + g_end - 1 - (i - n_tail) : g_first * LB_GROUP - 1 - (i - n_tail - n_groups); });
            } else if (pieces > WAVE) first_row_carry = lb_take_block<V, BLOCK, TAKE_BATCH>(*lb, pieces, s_wave_val, ok, [=](int i) { return tile - 1 - i; });
            else if (tid < WAVE) first_row_carry = lb_take_wave<V>(*lb, tile, pieces, ok);
            // a poll ran out (see "NOTHING DEPENDS ON A RECORD ARRIVING" above): the whole block computes the sum from the
            // matrix -- the row's nonzeros before this tile (its cancelled slots are wiped by their publishers when those come)
            if (__syncthreads_or(ok ? 0 : 1)) {
                first_row_carry = recompute_row_head<V, BLOCK>(p, first_row_start, c0.y, s_wave_val);
                if (tid == 0 && lb->error) {
                    __hip_atomic_store(lb->error, (int) lb->call_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(lb->error + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }

    // ---- row phase
    V *__restrict__ y = p.y + c0.x;
    for (int r = tid; r < tile_rows; r += BLOCK) {
        const int e = s_end[r];
        const int e0 = r > 0 ? s_end[r - 1] : 0;
        V sum = e > e0 ? s_prod_raw[prod_slot<V, CPT>(pshift + e - 1)] : (V) 0;
        if (r == 0) sum += first_row_carry;
        if (AXPBY) y[r] = p.alpha * sum + (p.beta == (V) 0 ? (V) 0 : p.beta * y[r]);
        else y[r] = sum;
    }
}

// ---------------------------------------------------------------------------
// LEAN in-tile reduction for tiles of short rows (tile_kernel_snap): when a row-snapped tile is CLOSED -- it starts at the
// first nonzero of a row and ends with the last nonzero of a row: nothing flows in, nothing is left open -- and its rows
// are short on average, the flag bits, the per-thread running sums, the block-wide segmented scan, the write-back of the
// running sums and two of the three barriers of consume_tile_flags buy nothing: a thread takes a whole row and adds
// up its products straight from LDS.  What bounds a matrix of 4-7 nonzeros per row that lives in the Infinity Cache (the
// reference's --dense=5, the 5/7-point grids) is VALU issue, not bytes (profiles/r03_short_rows/): the general reduction
// costs ~290 vector instructions per wave and tile, this one ~60 on such a tile.
//   * products were staged at their RAW positions (prod_unit(u, plain = true)), so row r's products are the consecutive
//     LDS elements [pshift + e0, pshift + e1), read with immediate offsets from one address per row;
//   * summation order: strictly left to right from +0.0 -- for every row of at most LEAN_SERIAL nonzeros that IS the order of
//     the reference's SpmvGold (gpu_spmv.cu:262-278: partial = 0; partial += v * x, one product at a time), so such rows
//     are bit for bit the sequential definition's; longer rows (rare in a tile that qualifies) are summed by the 16 lanes
//     of a DPP row, strided, then folded: another fixed order;
//   * which reduction a tile takes depends only on its boundaries (closed, nonzeros <= lean_avg * rows), i.e. on the
//     matrix and the tile shape -- never on hints, timing or the device.
// ---------------------------------------------------------------------------
// which tile shapes take the branch likelihoods below: the small ones (what a small problem runs on).  In the fp64 256x11 kernel the
// same hints changed nothing on a full device (dense5, the grids, dense32, circuit, Orkut-sized within +-0.5 %), in the fp32 one
// they cost spills.
template <typename V, int IPT> constexpr bool layout_hints() { return IPT <= 7; }
#define MSPMV_LIKELY(on, c) ((on) ? __builtin_expect(!!(c), 1) : !!(c))      // (on: a constant -- which tile shapes take the hint)
#define MSPMV_UNLIKELY(on, c) ((on) ? __builtin_expect(!!(c), 0) : !!(c))
// acc += v[j] for j < len - base, j = 0 .. 7, left to right, under a SHRINKING EXEC MASK: v_cmpx + v_add per product -- the lanes
// leave as their row ends -- instead of compare + select(s) + add (3 instructions per product in fp32, 4 in fp64).  The same additions
// in the same order as `acc = j < len ? acc + v[j] : acc`; EXEC is restored.  (The lean reductions of short-row tiles are bound by
// VALU issue, not bytes: profiles/r04_short_rows/.)
template <typename V> struct CompactChain;
template <> struct CompactChain<double> {
    // the first half alone: rows of at most four products (the caller knows that no lane of the wave has more)
    static __device__ __forceinline__ void add4(double &acc, const double (&v)[4], int len)
    {
        unsigned long long sv;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 0, %[l]\n\tv_add_f64 %[a], %[a], %[v0]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 1, %[l]\n\tv_add_f64 %[a], %[a], %[v1]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 2, %[l]\n\tv_add_f64 %[a], %[a], %[v2]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 3, %[l]\n\tv_add_f64 %[a], %[a], %[v3]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(sv)
                     : [l] "v"(len), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3])
                     : "vcc");
    }
    // acc (+0.0 on entry) += v[j] for j < len, left to right, lanes leaving as their row ends (EXEC restored)
    static __device__ __forceinline__ void add8(double &acc, const double (&v)[8], int len, int base)
    {
        unsigned long long sv;
        const int l = len - base;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 0, %[l]\n\tv_add_f64 %[a], %[a], %[v0]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 1, %[l]\n\tv_add_f64 %[a], %[a], %[v1]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 2, %[l]\n\tv_add_f64 %[a], %[a], %[v2]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 3, %[l]\n\tv_add_f64 %[a], %[a], %[v3]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 4, %[l]\n\tv_add_f64 %[a], %[a], %[v4]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 5, %[l]\n\tv_add_f64 %[a], %[a], %[v5]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 6, %[l]\n\tv_add_f64 %[a], %[a], %[v6]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 7, %[l]\n\tv_add_f64 %[a], %[a], %[v7]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(sv)
                     : [l] "v"(l), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7])
                     : "vcc");
    }
};
template <> struct CompactChain<float> {
    // the first half alone: rows of at most four products (the caller knows that no lane of the wave has more)
    static __device__ __forceinline__ void add4(float &acc, const float (&v)[4], int len)
    {
        unsigned long long sv;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 0, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v0]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 1, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v1]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 2, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v2]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 3, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v3]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(sv)
                     : [l] "v"(len), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3])
                     : "vcc");
    }
    static __device__ __forceinline__ void add8(float &acc, const float (&v)[8], int len, int base)
    {
        unsigned long long sv;
        const int l = len - base;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 0, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v0]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 1, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v1]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 2, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v2]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 3, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v3]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 4, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v4]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 5, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v5]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 6, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v6]\n\t"
                     "v_cmpx_lt_i32_e32 vcc, 7, %[l]\n\tv_add_f32_e32 %[a], %[a], %[v7]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(sv)
                     : [l] "v"(l), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7])
                     : "vcc");
    }
};
constexpr int LEAN_SERIAL = 16;        // rows up to this long are summed by one thread
constexpr int LEAN_GROUP_MAX = 256;    // rows up to this long (and longer than LEAN_SERIAL) are summed by the 16 lanes of one DPP row; longer ones by the wave
// Rows of a lean tile that are longer than LEAN_SERIAL -- every lane of the wave calls this with ITS row's first product position
// `pos` (LDS element index), length `len` (0: no row) and running sum `acc`; the owners of such rows get their row's total in `acc`.
//   * LEAN_SERIAL < len <= LEAN_GROUP_MAX: four rows at a time, each by the 16 lanes of one DPP row -- lane j adds products j, j + 16, ...
//     from +0.0, the 16 partial sums are folded left to right (row_shr 1, 2, 4, 8): summation depth <= len / 16 + 4 <= 20;
//   * longer rows (a closed lean tile can hold ONE row of ~2000-3000 nonzeros next to hundreds of empty ones): one at a time by the
//     whole wave -- lane j keeps FOUR running sums, over products j + 64 u + 256 r (u = 0 .. 3), adds them pairwise, and the 64
//     totals are folded by row_shr 1, 2, 4, 8, row_bcast 15 and 31: depth <= len / 256 + 2 + 6 <= 20 for any row a tile can hold.
//     (Until round 5 these rows went to a 16-lane group as well: depth len / 16 + 4, ~150 for such a row -- beyond the
//      2 (ceil(log2(len + 1)) + items_per_thread + 8) of the stated bound, which the tests then only met statistically.)
// A fixed order either way; shared by consume_tile_rows and the compact front end, so the two write the same bits.
template <typename V>
__device__ __forceinline__ void lean_long_rows(const V *s_prod, int pos, int len, int lane, V &acc)
{
    unsigned long long pending = __ballot(len > LEAN_SERIAL && len <= LEAN_GROUP_MAX);
    while (pending != 0ull) {                                           // wave-uniform
        int owner[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            owner[g] = pending != 0ull ? __ffsll((long long) pending) - 1 : -1;
            pending &= pending - 1ull;                                  // (0 stays 0)
        }
        const int grp = lane >> 4, j = lane & 15;
        const int own = grp == 0 ? owner[0] : grp == 1 ? owner[1] : grp == 2 ? owner[2] : owner[3];
        // (both shuffles by ALL lanes, the selection afterwards: a lane switched off by a branch cannot be read from)
        const int g_pos = __shfl(pos, own < 0 ? 0 : own, WAVE);
        const int g_len_any = __shfl(len, own < 0 ? 0 : own, WAVE);
        const int g_len = own < 0 ? 0 : g_len_any;
        const V *gsrc = s_prod + g_pos;
        V part = (V) 0;
        for (int k = j; __ballot(k < g_len) != 0ull; k += 64) {
            V u4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) u4[u] = k + 16 * u < g_len ? gsrc[k + 16 * u] : (V) 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) part += u4[u];
        }
        part += dpp_move<0x111, 0xf>((V) 0, part);                     // row_shr:1
        part += dpp_move<0x112, 0xf>((V) 0, part);                     // row_shr:2
        part += dpp_move<0x114, 0xf>((V) 0, part);                     // row_shr:4
        part += dpp_move<0x118, 0xf>((V) 0, part);                     // row_shr:8  -> lane 15 of every row holds its total
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const V total = __shfl(part, 16 * g + 15, WAVE);
            if (lane == owner[g]) acc = total;
        }
    }
    pending = __ballot(len > LEAN_GROUP_MAX);
    while (pending != 0ull) {                                           // wave-uniform: one row per round, by the whole wave
        const int own = __ffsll((long long) pending) - 1;
        pending &= pending - 1ull;
        const int g_pos = __shfl(pos, own, WAVE), g_len = __shfl(len, own, WAVE);
        const V *gsrc = s_prod + g_pos;
        V a4[4] = {(V) 0, (V) 0, (V) 0, (V) 0};
        for (int k = lane; k < g_len; k += 4 * WAVE) {                  // (the lanes leave as the row runs out: no ballot needed, nothing is exchanged inside)
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] += k + WAVE * u < g_len ? gsrc[k + WAVE * u] : (V) 0;
        }
        V part = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        part += dpp_move<0x111, 0xf>((V) 0, part);                     // row_shr:1
        part += dpp_move<0x112, 0xf>((V) 0, part);                     // row_shr:2
        part += dpp_move<0x114, 0xf>((V) 0, part);                     // row_shr:4
        part += dpp_move<0x118, 0xf>((V) 0, part);                     // row_shr:8  -> lane 15 of every row of 16 holds that row's total
        part += dpp_move<0x142, 0xa>((V) 0, part);                     // row_bcast15 into rows 1 and 3
        part += dpp_move<0x143, 0xc>((V) 0, part);                     // row_bcast31 into rows 2 and 3 -> lane 63 holds the total
        const V total = __shfl(part, WAVE - 1, WAVE);
        if (lane == own) acc = total;
    }
}

constexpr int LEAN_BATCH = 8;          // products of a row requested before any is looked at
template <typename V, int BLOCK, int IPT, bool AXPBY>
__device__ __forceinline__ void consume_tile_rows(const Params<V> &p, const Coord c0, int tile_rows, const end16_t *s_end,
                                                  const V *s_prod_raw, int pshift, Carry<V> *__restrict__ carry_out,
                                                  unsigned long long *tr = nullptr)
{
    const int tid = threadIdx.x;
#ifdef MSPMV_DEV
#define MSPMV_LEAN_TR(i) do { if (tr && tid == 0) tr[i] = wall_clock64(); } while (0)
#else
#define MSPMV_LEAN_TR(i) do { } while (0)
#endif
    MSPMV_LEAN_TR(8);
    if (tid == BLOCK - 1) { Carry<V> c; c.key = c0.x + tile_rows; c.value = (V) 0; *carry_out = c; }     // (nothing open: what mspmv_debug_read_tiles reports)
    V *__restrict__ y = p.y + c0.x;
    // Two rows per thread and iteration (r, r + BLOCK), and every LDS read of a step requested before the first is waited for:
    // the row ends of both rows, then the first LEAN_BATCH products of both (read whether the row has them or not: they lie
    // inside the LDS arrays for any lane, pshift + e0 + 15 < SLOTS) -- two LDS round trips per pair of rows instead of six.
    for (int r0 = 0; r0 < tile_rows; r0 += 2 * BLOCK) {           // block-uniform trip count
        int e0[2], len[2]; bool valid[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + h * BLOCK + tid;
            valid[h] = r < tile_rows;
            const int rr = valid[h] ? r : 0;                       // (row 0 exists: tile_rows > 0 here)
            const int e1 = s_end[rr];
            const int ep = s_end[rr > 0 ? rr - 1 : 0];             // (read whatever rr is: a select afterwards, not a branch around a second LDS round trip)
            e0[h] = rr > 0 ? ep : 0;
            len[h] = valid[h] ? e1 - e0[h] : 0;
        }
        if (r0 == 0) MSPMV_LEAN_TR(9);
        // (a wave none of whose rows -- of this pair of rounds -- has more than four products reads and adds the first four of each only: the
        //  row-strided product reads are bank-conflict-laden, 57 % of the LDS cycles of a 5-point grid's launch, and the LDS is busy for more
        //  than half of such a kernel: profiles/r05_lds_counters.txt)
        const bool more = __ballot(len[0] > LEAN_BATCH / 2 || len[1] > LEAN_BATCH / 2) != 0ull;      // wave-uniform
        V v[2][LEAN_BATCH];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < LEAN_BATCH / 2; ++j) v[h][j] = s_prod_raw[pshift + e0[h] + j];
        V acc[2] = {(V) 0, (V) 0};
        static_assert(LEAN_BATCH == 8, "CompactChain adds eight");
        if (more) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = LEAN_BATCH / 2; j < LEAN_BATCH; ++j) v[h][j] = s_prod_raw[pshift + e0[h] + j];
            // (all of them requested here, in one go: without this the compiler sinks a read into the branch of the first addition that uses it)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < LEAN_BATCH; ++j) asm volatile("" : "+v"(v[h][j]));
#pragma unroll
            for (int h = 0; h < 2; ++h) CompactChain<V>::add8(acc[h], v[h], len[h], 0);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < LEAN_BATCH / 2; ++j) asm volatile("" : "+v"(v[h][j]));
#pragma unroll
            for (int h = 0; h < 2; ++h) { const V q[4] = {v[h][0], v[h][1], v[h][2], v[h][3]}; CompactChain<V>::add4(acc[h], q, len[h]); }
        }
        if (r0 == 0) MSPMV_LEAN_TR(10);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const V *src = s_prod_raw + (pshift + e0[h]);
            // rows of LEAN_BATCH + 1 ... LEAN_SERIAL nonzeros: the same left-to-right order, continued
            if (MSPMV_UNLIKELY((layout_hints<V, IPT>()), __ballot(len[h] > LEAN_BATCH) != 0ull)) {
                V w[LEAN_SERIAL - LEAN_BATCH];
#pragma unroll
                for (int j = 0; j < LEAN_SERIAL - LEAN_BATCH; ++j) w[j] = src[LEAN_BATCH + j];
                CompactChain<V>::add8(acc[h], w, len[h], LEAN_BATCH);
            }
            // rows longer than that: by 16-lane groups, the longest by the whole wave (lean_long_rows)
            if (MSPMV_UNLIKELY((layout_hints<V, IPT>()), __ballot(len[h] > LEAN_SERIAL) != 0ull))
                lean_long_rows<V>(s_prod_raw, pshift + e0[h], len[h], tid & (WAVE - 1), acc[h]);
            if (r0 == 0 && h == 0) MSPMV_LEAN_TR(11);
            if (valid[h]) {
                const int r = r0 + h * BLOCK + tid;
                if (AXPBY) y[r] = p.alpha * acc[h] + (p.beta == (V) 0 ? (V) 0 : p.beta * y[r]);
                else y[r] = acc[h];
            }
        }
    }
    MSPMV_LEAN_TR(12);
#undef MSPMV_LEAN_TR
}

// ---------------------------------------------------------------------------
// One block per tile, dword-per-lane staging: the fallback for CSR arrays
// whose base addresses are not 16-byte aligned and for tiny matrices.
// ref: DeviceSpmvKernel / AgentSpmv::ConsumeTile, dispatch_spmv_orig.cuh:157-186,
// agent_spmv_orig.cuh:413-639,856-914.
// ---------------------------------------------------------------------------
template <typename V, int BLOCK, int IPT, bool AXPBY>
__global__ __launch_bounds__(BLOCK) void tile_kernel(Params<V> p, const Coord *__restrict__ coords,
                                                     Carry<V> *__restrict__ carries, int num_tiles)
{
    constexpr int TILE = BLOCK * IPT;
    constexpr int NW = BLOCK / WAVE;
    constexpr int PAD = IPT + 4;
    __shared__ int s_end[TILE + PAD];
    __shared__ V s_prod[TILE + PAD];
    __shared__ int s_wave_key[NW];
    __shared__ V s_wave_val[NW];

    const int tile = blockIdx.x;
    const int tid = threadIdx.x;
    const Coord c0 = coords[tile];
    const Coord c1 = coords[tile + 1];
    const int tile_rows = c1.x - c0.x;
    const int tile_nnz = c1.y - c0.y;

    const int *__restrict__ cols = p.cols + c0.y;
    const V *__restrict__ vals = p.values + c0.y;
    int col_r[IPT];
    V val_r[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * BLOCK;
        if (j < tile_nnz) { col_r[k] = cols[j]; val_r[k] = vals[j]; }
    }
    const int *__restrict__ row_end = p.row_end + c0.x;
    for (int r = tid; r < TILE + PAD; r += BLOCK) s_end[r] = r < tile_rows ? row_end[r] - c0.y : 0x3fffffff;
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * BLOCK;
        s_prod[j] = j < tile_nnz ? val_r[k] * p.x[col_r[k]] : (V) 0;
    }
    if (tid < PAD) s_prod[TILE + tid] = (V) 0;
    __syncthreads();
    consume_tile_lds<V, BLOCK, IPT, AXPBY>(p, c0, tile_rows, tile_nnz, s_end, s_prod, s_prod, s_wave_key, s_wave_val,
                                           carries + tile);
}

// ---------------------------------------------------------------------------
// The production kernel: 16-byte streaming loads, LDS-staged products, flag/segmented-scan
// reduction (consume_tile_flags).  PERSIST = false (default): one tile per block; the
// hardware's block scheduler balances the load and de-phases the blocks of a CU.
// PERSIST = true (tuning flags): a block walks tiles blockIdx.x, +gridDim.x, ... and requests
// the NEXT tile's nonzeros (into the registers the current tile has just drained into LDS)
// before the LDS phases of the current one.  The persistent form was the faster one while the
// in-tile reduction was the per-thread path walk; measured with the present reduction a resident
// grid is 7-10 % slower on streaming matrices and 2-4 tiles per block are within +-3 % of one
// (tools/sweep.py with SWEEP_FLAGS=0x100000..0x800000), so the simpler launch is the default.
// What bounds the kernel is the CU's vector-memory pipeline: cycle stamps (tools/trace_tiles.py)
// show ~70 % of a block's life spent issuing into / waiting on it, and throughput follows the
// number of cache lines requested per tile, not the instruction count of the LDS phases.
//
// Staging works on 4-element chunks aligned in ARRAY index space (the CSR
// arrays are 16-byte aligned -- checked by the dispatcher -- but a tile starts
// anywhere): chunk addresses are clamped to the last full chunk of the array
// instead of being branched around, so every load is unconditional,
// straight-line code (a branch per chunk made hipcc wait for each load before
// issuing the next); elements outside the tile are predicated off, and the
// <= 3 elements of a ragged array tail are read by scalar loads in a rarely
// taken branch.  Nothing outside [0,nnz) / [0,rows] is ever read.
// ---------------------------------------------------------------------------
// Resident blocks per CU the vectorised tile kernels are compiled for: what their LDS
// footprint admits (160 KiB per CU), at most 32 waves per CU.  Passed to __launch_bounds__ as
// waves per SIMD: left alone, hipcc spent 92-104 VGPRs on these kernels (4-5 waves/SIMD); told
// the target it fits 64-80 without spilling, which is what actually sets the residency.
template <typename V, int BLOCK, int IPT, bool WIDE_ENDS = false>
constexpr int tile_blocks_per_cu()
{
    constexpr int slots = (IPT / 4 + 1) * BLOCK * 4;
    constexpr int lds = slots * ((WIDE_ENDS ? 4 : 2) + (int) sizeof(V)) + slots / 8 + 256;      // products, row ends (16-bit; 32 for the walk), flag bits
    constexpr int by_lds = 163840 / lds;
    constexpr int by_waves = 2048 / BLOCK;
    return by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves;
}
template <typename V, int BLOCK, int IPT, bool RELAX = false, bool WIDE_ENDS = false>
constexpr int tile_waves_per_simd()
{
    // one wave less than the LDS footprint admits (never more than 6 for fp64): with both
    // staging paths in the kernel the tighter budget spills 3 registers per lane, and the
    // measured difference between the two choices is within noise
    int w = (tile_blocks_per_cu<V, BLOCK, IPT, WIDE_ENDS>() * BLOCK + 255) / 256;
    if (sizeof(V) == 8 && w > 6) w = 6;
    if (w > 4 && (!RELAX || WIDE_ENDS)) w -= WIDE_ENDS && w > 5 ? 2 : 1;
    return w;
}

// Column-band passes, fp64: the values of a chunk are fetched only when one of its four columns lies in the pass's band
// (with sorted columns a band's nonzeros are a run inside each row: C2 fp64 skips 57 % of the 32-byte value loads per pass,
// 1.447 -> 1.342 ms).  Not for fp32: the extra live state spills there (0.87 -> 0.95 ms, and the scratch set-up costs the
// ordinary body of the same kernel 20 %).
template <typename V, bool BAND> constexpr bool band_lazy_values() { return BAND && sizeof(V) == 8; }

template <typename V, int BLOCK, int IPT>
struct TileRegs {
    static constexpr int CPT = IPT / 4 + 1;   // 4-element chunks per thread: covers TILE + 3
    Vec4<int> col[CPT];
    Vec4<V> val[CPT];
};

// fp64 values of a 4-element chunk are 32 bytes: fetched as two 16-byte loads per lane, each load instruction of a wave would
// touch every 128-byte line of the wave's 2 KB twice, half a line at a time -- and the second touch of a line read with a
// non-temporal load is another request to L2 (measured: making the second half an ordinary load, which then hits the CU's
// cache, is worth 5-9 % on dense32 / 7-point-grid streams, but gives the lines ordinary retention and costs the
// gather-bound matrices 2-3 % of their x hits).  Instead the two loads are laid out line by line: in the first, lanes 0-31
// fetch the first half of their own chunk and lanes 32-63 the SECOND half of the chunk of lane - 32 (8 whole lines); in the
// second, lanes 0-31 fetch the first half of the chunk of lane + 32 and lanes 32-63 their own second half; four
// v_permlane32_swap (gfx950) then hand every lane its own 32 bytes.  Wave-uniform control flow required.
// (The swaps are a separate step, linewise_own, done by the staging right before the products: next to the loads the compiler
//  waits for each pair before it requests the next one.)
template <bool NT>
__device__ __forceinline__ Vec4<double> ld_stream4_linewise(const double *base, int e_own, int e_partner, int tid)
{
    const bool hi = (tid & 32) != 0;
    const double2v *pa = reinterpret_cast<const double2v *>(base + (hi ? e_partner + 2 : e_own));
    const double2v *pb = reinterpret_cast<const double2v *>(base + (hi ? e_own + 2 : e_partner));
    Vec4<double> r;
    r.a = NT ? __builtin_nontemporal_load(pa) : *pa;
    r.b = NT ? __builtin_nontemporal_load(pb) : *pb;
    return r;
}
// what ld_stream4_linewise fetched -> every lane's own four values.  Wave-uniform control flow required.
__device__ __forceinline__ Vec4<double> linewise_own(const Vec4<double> &w)
{
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    uint4v ua = __builtin_bit_cast(uint4v, w.a), ub = __builtin_bit_cast(uint4v, w.b);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        // lanes 32-63 of the first operand <-> lanes 0-31 of the second
        const auto sw = __builtin_amdgcn_permlane32_swap(ua[d], ub[d], false, false);
        ua[d] = sw[0]; ub[d] = sw[1];
    }
    Vec4<double> r; r.a = __builtin_bit_cast(double2v, ua); r.b = __builtin_bit_cast(double2v, ub);
    return r;
}
__device__ __forceinline__ Vec4<float> linewise_own(const Vec4<float> &w) { return w; }
// values as issue_nonzero_loads left them in the tile's registers -> the values of the lane's own chunk k
template <typename V, bool NT, bool LAZY>
constexpr bool vals_linewise() { return sizeof(V) == 8 && NT && !LAZY; }
// WIRE staging (the production kernels): the values stay as ld_stream4_linewise fetched them -- val.a = the first half of the own chunk
// (lanes 0-31) / the second half of the chunk 32 lanes down (lanes 32-63), val.b = the first half of the chunk 32 lanes up / the own
// second half -- and the COLUMNS are swapped to match (two v_permlane32_swap instead of four): every lane then multiplies two pairs that
// belong to two different chunks, and writes each pair where it belongs in the LDS tile.  Lanes 0-31 and 32-63 of one store
// instruction together cover 1 KB of consecutive 16-byte units (a lane writing its own 32 bytes as two stores 32 bytes apart
// collided two-way with the lane four further on).  cols[0..1] belong to val.a, cols[2..3] to val.b.
__device__ __forceinline__ void wire_cols(const Vec4<int> &col, int (&c)[4])
{
    const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned) col.v.x, (unsigned) col.v.z, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned) col.v.y, (unsigned) col.v.w, false, false);
    c[0] = (int) s0[0]; c[1] = (int) s1[0]; c[2] = (int) s0[1]; c[3] = (int) s1[1];
}
// chunk index (in units of 4 elements, relative to the tile's first chunk) and element offset inside it of the pair in val.a / val.b
struct WirePos { int qa, qb, off; };
__device__ __forceinline__ WirePos wire_pos(int q, int tid)
{
    const bool hi = (tid & 32) != 0;
    WirePos w; w.qa = hi ? q - 32 : q; w.qb = hi ? q : q + 32; w.off = hi ? 2 : 0;
    return w;
}

// (A wave-uniform skip of chunks that lie past the tile altogether -- one or two of the four waves of a short-row tile -- was built for
//  the interior staging in round 5 and not kept: every branch around a group of loads makes the compiler wait at the join, and the
//  result was mixed -- grid3d-200 -2.3 %, band5 -3 %, but the circuit-shaped matrix +8 %, dense32 fp64 +3.5 %, dense5 +2 %:
//  profiles/r05_ab_wave_skip_general_kernel.txt.  The compact front end, whose loads are few and whose tiles share a CU in small
//  numbers, keeps it.)
template <typename V, int BLOCK, int IPT, bool NT, bool VALS = true, bool LINEWISE_OK = true>
__device__ __forceinline__ void issue_nonzero_loads(const Params<V> &p, const Coord c0, const Coord c1,
                                                    TileRegs<V, BLOCK, IPT> &r, int tid_in = -1)
{
    const int tid = tid_in < 0 ? (int) threadIdx.x : tid_in;
    constexpr int CPT = IPT / 4 + 1;
    const int a0 = c0.y & ~3;
    const int last_full = (p.nnz & ~3) - 4;        // first element of the array's last full chunk (nnz >= 4)
    const int safe = a0 < last_full ? a0 : last_full;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        int e0 = a0 + 4 * (tid + k * BLOCK);
        // a chunk past the tile (it belongs to the next tile) or past the last full chunk of
        // the array is not fetched: those lanes re-read the tile's first chunk instead (one
        // cached address), so no byte of HBM traffic is spent on data this tile does not use
        e0 = (e0 < c1.y && e0 <= last_full) ? e0 : safe;
        r.col[k] = ld_stream4<NT>(p.cols + e0);
        if constexpr (VALS && LINEWISE_OK && vals_linewise<V, NT, false>()) {     // (ordinary loads: the second half hits the CU's cache anyway, and the swaps cost 3 % on dense5)
            int e1 = a0 + 4 * ((tid ^ 32) + k * BLOCK);          // the chunk of the lane 32 away, by the same rule
            e1 = (e1 < c1.y && e1 <= last_full) ? e1 : safe;
            r.val[k] = ld_stream4_linewise<NT>(p.values, e0, e1, tid);
        } else if constexpr (VALS) r.val[k] = ld_stream4<NT>(p.values + e0);      // (!VALS: column-band passes fetch the values later, by band)
    }
}

// Stage one tile whose nonzeros are already in `regs` (requested earlier): row offsets
// and x gathers are issued, tile-relative row ends and products land in LDS (every slot of
// both arrays is written: +inf / 0 outside the tile), the ragged array tails are patched,
// and the block is synchronised.
// (after_gather: called once the x gathers have been requested -- scalar work of the caller that then runs in the shadow of their
//  latency instead of after the staging barrier: the hint verdict of tile_kernel_snap)
struct NoAfterGather { __device__ __forceinline__ void operator()() const {} };
template <typename V, int BLOCK, int IPT, bool NT, bool FL, bool XL = false, bool BAND = false, typename AG = NoAfterGather>
__device__ __forceinline__ void stage_tile_careful(const Params<V> &p, const Coord c0, const Coord c1,
                                           const TileRegs<V, BLOCK, IPT> &regs, typename EndType<FL>::type *s_end_raw, V *s_prod_raw,
                                           int last_full_nz, int last_full_ro, unsigned *s_flag, const V *s_x = nullptr, int tid_in = -1,
                                           bool lean = false, AG after_gather = AG())
{
    // lean (block-uniform; FL only): the tile will be reduced row by row (consume_tile_rows) -- no row-start bits, products at their raw positions
    constexpr int CPT = IPT / 4 + 1;
    const int tid = tid_in < 0 ? (int) threadIdx.x : tid_in;
    const int *__restrict__ row_offsets = p.row_end - 1;
    const int tile_rows = c1.x - c0.x;
    const int tile_nnz = c1.y - c0.y;
    const int a0 = c0.y & ~3;
    const int first = c0.x + 1;                // d_row_offsets index of the tile's first row end
    const int i0 = first & ~3;
    const int eshift = first - i0;
    // ---- row ends of the current tile -> LDS (chunks of d_row_offsets)
    Vec4<int> ro[CPT];
    const int ro_chunks = (tile_rows + eshift + 3) / 4;           // chunks holding a row end of this tile
    const int ro_safe = i0 < last_full_ro ? i0 : last_full_ro;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int q = tid + k * BLOCK;
        int i = i0 + 4 * q;
        i = (q < ro_chunks && i <= last_full_ro) ? i : ro_safe;
        ro[k] = ld_stream4<NT>(row_offsets + i);
    }
    // ---- gather x for the current tile (its nonzeros were requested one iteration ago)
    V xv[CPT][4];
    unsigned in_band = 0u;                     // BAND: one bit per staged nonzero of this thread (its column lies in the pass's band)
    constexpr bool LAZY = band_lazy_values<V, BAND>();
    Vec4<V> bval[LAZY ? CPT : 1];              // LAZY: the values, fetched here and only for chunks with a nonzero of the band
    constexpr bool WIRE = vals_linewise<V, NT, LAZY>() && FL;
    unsigned wire_in = 0u;                     // WIRE: one bit per held nonzero that lies in the tile
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int e0 = a0 + 4 * (tid + k * BLOCK);
        if constexpr (WIRE) {
            int gc[4]; wire_cols(regs.col[k], gc);
            const WirePos wp = wire_pos(tid + k * BLOCK, tid);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int eb = a0 + 4 * (i < 2 ? wp.qa : wp.qb);             // first element of the chunk this value belongs to
                const bool in = (unsigned) (eb + wp.off + (i & 1) - c0.y) < (unsigned) tile_nnz && eb <= last_full_nz;
                wire_in |= in ? 1u << (4 * k + i) : 0u;
                xv[k][i] = XL ? s_x[in ? gc[i] : 0] : p.x[in ? gc[i] : 0];
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = (unsigned) (e0 + i - c0.y) < (unsigned) tile_nnz && e0 <= last_full_nz;
            if (BAND) {
                xv[k][i] = (V) 0;
                if (in && (unsigned) (regs.col[k].get(i) - p.band_lo) < (unsigned) p.band_len) { xv[k][i] = p.x[regs.col[k].get(i)]; in_band |= 1u << (4 * k + i); }
            } else xv[k][i] = XL ? s_x[in ? regs.col[k].get(i) : 0] : p.x[in ? regs.col[k].get(i) : 0];
        }
        if constexpr (LAZY) {
            zero4(bval[k]);
            if ((in_band >> (4 * k)) & 0xfu) bval[k] = ld_stream4<NT>(p.values + e0);     // (a bit set => a chunk of this tile, <= last_full_nz)
        }
    }
    after_gather();
    // ---- stage row ends
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int q = tid + k * BLOCK;
        const int i = i0 + 4 * q;
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * q - eshift + j;
            const bool in = r >= 0 && r < tile_rows && i <= last_full_ro;
            v[j] = in ? ro[k].get(j) - c0.y : 0x3fffffff;
            if (FL && !lean && (unsigned) v[j] < (unsigned) tile_nnz) {
                const int b = (c0.y - a0) + v[j];
                atomicOr(&s_flag[b >> 5], 1u << (b & 31));
            }
        }
        st_lds4(&s_end_raw[4 * q], v);
    }
    if (FL && !lean && tid == 0) atomicOr(&s_flag[0], 1u << (c0.y - a0));
    // ---- stage products
    Vec4<V> own_val[LAZY || WIRE ? 1 : CPT];
    if constexpr (!LAZY && !WIRE) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) own_val[k] = vals_linewise<V, NT, LAZY>() ? linewise_own(regs.val[k]) : regs.val[k];
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int chunk = tid + k * BLOCK;
        const int e0 = a0 + 4 * chunk;
        if constexpr (WIRE) {
            const WirePos wp = wire_pos(chunk, tid);
            const V w[4] = {regs.val[k].get(0), regs.val[k].get(1), regs.val[k].get(2), regs.val[k].get(3)};
            V pa[2], pb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pa[i] = ((wire_in >> (4 * k + i)) & 1u) ? w[i] * xv[k][i] : (V) 0;
                pb[i] = ((wire_in >> (4 * k + 2 + i)) & 1u) ? w[2 + i] * xv[k][2 + i] : (V) 0;
            }
            st_unit(&s_prod_raw[2 * prod_unit<V, CPT>(2 * wp.qa + (wp.off >> 1), lean)], pa);
            st_unit(&s_prod_raw[2 * prod_unit<V, CPT>(2 * wp.qb + (wp.off >> 1), lean)], pb);
            continue;
        }
        V prod[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = BAND ? ((in_band >> (4 * k + i)) & 1u) != 0u
                                 : (unsigned) (e0 + i - c0.y) < (unsigned) tile_nnz && e0 <= last_full_nz;
            if constexpr (LAZY) prod[i] = in ? bval[k].get(i) * xv[k][i] : (V) 0;
            else prod[i] = in ? own_val[k].get(i) * xv[k][i] : (V) 0;
        }
        if (FL) st_prod_chunk<CPT>(s_prod_raw, chunk, prod, lean);
        else st_lds4(&s_prod_raw[swz_prod(4 * chunk)], prod);
    }
    // ---- ragged array tails (at most 3 elements each; only the tile that reaches the array
    //      end).  Block-uniform branch; the barrier orders these writes after the zeros /
    //      sentinels the chunk owners stored into the same slots above.
    const bool nz_tail = c1.y > last_full_nz + 4;
    const bool ro_tail = first + tile_rows > last_full_ro + 4;
    if (nz_tail || ro_tail) {
        __syncthreads();
        const int j = last_full_nz + 4 + tid;                       // absolute nonzero index
        if (nz_tail && j < c1.y && j >= c0.y) {
            const int c = ld_stream<NT>(p.cols + j);
            const bool inb = !BAND || (unsigned) (c - p.band_lo) < (unsigned) p.band_len;
            s_prod_raw[FL ? prod_slot<V, CPT>(j - a0, lean) : swz_prod(j - a0)] = inb ? ld_stream<NT>(p.values + j) * (XL ? s_x[c] : p.x[c]) : (V) 0;
        }
        const int i = last_full_ro + 4 + tid;                       // absolute d_row_offsets index
        const int r = i - first;
        if (ro_tail && r >= 0 && r < tile_rows) {
            const int v = ld_stream<NT>(row_offsets + i) - c0.y;
            s_end_raw[r + eshift] = (typename EndType<FL>::type) v;
            if (FL && !lean && (unsigned) v < (unsigned) tile_nnz) atomicOr(&s_flag[((c0.y - a0) + v) >> 5], 1u << (((c0.y - a0) + v) & 31));
        }
    }
    __syncthreads();
}

// Interior tiles (all but the handful whose chunk loads would touch the last, ragged chunk of
// an array -- the matrix's final tile in particular): no per-element predicates at all.
//  * every column index in a loaded chunk is a valid index even when the element belongs to
//    the neighbouring tile, so x is gathered unconditionally;
//  * products of elements outside the tile land in LDS slots the walk never reads as
//    nonzeros (an item past the tile's nonzeros is always a row end there, and a row end's
//    product slot is read but discarded by a select, so even a NaN cannot leak);
//  * row offsets past the tile's last row are REAL offsets (>= the tile's nonzero count), which
//    serve as the "+inf" the search and the walk need: a row end of a later tile lies at
//    least IPT path items past every thread of a full tile.
// Only whole chunks beyond the needed range are redirected to a cached address (no HBM bytes
// for data the tile does not use).  This removes ~200 of the ~1100 instructions per wave per tile.
template <typename V, int BLOCK, int IPT, bool NT, bool FL, bool XL = false, bool BAND = false, typename AG = NoAfterGather>
__device__ __forceinline__ void stage_tile_interior(const Params<V> &p, const Coord c0, const Coord c1,
                                                    const TileRegs<V, BLOCK, IPT> &regs,
                                                    typename EndType<FL>::type *s_end_raw, V *s_prod_raw, unsigned *s_flag,
                                                    const V *s_x = nullptr, int tid_in = -1, bool lean = false, AG after_gather = AG())
{
    constexpr int CPT = IPT / 4 + 1;
    const int tid = tid_in < 0 ? (int) threadIdx.x : tid_in;
    const int *__restrict__ row_offsets = p.row_end - 1;
    const int tile_rows = c1.x - c0.x;
    const int first = c0.x + 1;
    const int i0 = first & ~3;
    const int eshift = first - i0;
    Vec4<int> ro[CPT];
    // row ends of the tile (+ the IPT the per-thread walk may peek at)
    const int ro_chunks = (tile_rows + eshift + (FL ? 0 : IPT) + 3) / 4;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        // block-uniform: a tile of long rows needs one chunk per thread at most; every vector-memory
        // instruction spared is a slot in the CU's memory pipeline, which is what this kernel queues on
        if (k == 0 || k * BLOCK < ro_chunks) {
            const int q = tid + k * BLOCK;
            const int i = q < ro_chunks ? i0 + 4 * q : i0;
            ro[k] = ld_stream4<NT>(row_offsets + (unsigned) i);
        }
    }
    V xv[CPT][4];
    unsigned in_band = 0u;                     // BAND: one bit per staged nonzero of this thread
    constexpr bool LAZY = band_lazy_values<V, BAND>();
    Vec4<V> bval[LAZY ? CPT : 1];              // LAZY: the values, fetched here and only for chunks with a nonzero of the band
    constexpr bool WIRE = vals_linewise<V, NT, LAZY>() && FL;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        if constexpr (WIRE) {
            int gc[4]; wire_cols(regs.col[k], gc);
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[k][i] = XL ? s_x[(unsigned) gc[i]] : p.x[(unsigned) gc[i]];
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (BAND) {
                xv[k][i] = (V) 0;
                if ((unsigned) (regs.col[k].get(i) - p.band_lo) < (unsigned) p.band_len) { xv[k][i] = p.x[(unsigned) regs.col[k].get(i)]; in_band |= 1u << (4 * k + i); }
            } else xv[k][i] = XL ? s_x[(unsigned) regs.col[k].get(i)] : p.x[(unsigned) regs.col[k].get(i)];
        }
        if constexpr (LAZY) {
            zero4(bval[k]);
            if ((in_band >> (4 * k)) & 0xfu) {
                // the address issue_nonzero_loads used for this chunk's columns (a chunk past the tile re-reads the tile's first)
                const int a0 = c0.y & ~3, last_full = (p.nnz & ~3) - 4;
                int e0 = a0 + 4 * (tid + k * BLOCK);
                e0 = (e0 < c1.y && e0 <= last_full) ? e0 : (a0 < last_full ? a0 : last_full);
                bval[k] = ld_stream4<NT>(p.values + e0);
            }
        }
    }
    after_gather();
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int q = tid + k * BLOCK;
        if (q < ro_chunks) {
            int v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ro[k].get(j) - c0.y;
            st_lds4(&s_end_raw[4 * q], v);
            if (FL && !lean) {
                // row starts inside the tile (rows before the tile give v <= 0, rows after it
                // v >= tile_nnz; v == 0 is the tile's first nonzero, flagged anyway)
                const int tile_nnz = c1.y - c0.y, pshift = c0.y - (c0.y & ~3);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((unsigned) v[j] < (unsigned) tile_nnz) atomicOr(&s_flag[(pshift + v[j]) >> 5], 1u << ((pshift + v[j]) & 31));
            }
        }
    }
    if (FL && !lean && tid == 0) atomicOr(&s_flag[0], 1u << (c0.y - (c0.y & ~3)));
    Vec4<V> own_val[LAZY || WIRE ? 1 : CPT];
    if constexpr (!LAZY && !WIRE) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) own_val[k] = vals_linewise<V, NT, LAZY>() ? linewise_own(regs.val[k]) : regs.val[k];
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int chunk = tid + k * BLOCK;
        if constexpr (WIRE) {
            const WirePos wp = wire_pos(chunk, tid);
            V pa[2], pb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { pa[i] = regs.val[k].get(i) * xv[k][i]; pb[i] = regs.val[k].get(2 + i) * xv[k][2 + i]; }
            st_unit(&s_prod_raw[2 * prod_unit<V, CPT>(2 * wp.qa + (wp.off >> 1), lean)], pa);
            st_unit(&s_prod_raw[2 * prod_unit<V, CPT>(2 * wp.qb + (wp.off >> 1), lean)], pb);
            continue;
        }
        V prod[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (LAZY) prod[i] = ((in_band >> (4 * k + i)) & 1u) ? bval[k].get(i) * xv[k][i] : (V) 0;
            else if constexpr (BAND) prod[i] = ((in_band >> (4 * k + i)) & 1u) ? own_val[k].get(i) * xv[k][i] : (V) 0;
            else prod[i] = own_val[k].get(i) * xv[k][i];
        }
        if (FL) st_prod_chunk<CPT>(s_prod_raw, chunk, prod, lean);
        else st_lds4(&s_prod_raw[swz_prod(4 * chunk)], prod);
    }
    __syncthreads();
}

// true when every vector load of the interior path stays inside full chunks of the arrays
template <int IPT>
__device__ __forceinline__ bool tile_is_interior(const Coord c0, const Coord c1, int tile, int num_tiles,
                                                 int last_full_nz, int last_full_ro)
{
    const int first = c0.x + 1;
    const int i0 = first & ~3;
    const int ro_chunks = ((c1.x - c0.x) + (first - i0) + IPT + 3) / 4;
    return tile != num_tiles - 1 && c1.y <= last_full_nz + 4 && i0 + 4 * ro_chunks <= last_full_ro + 4;
}

template <typename V, int BLOCK, int IPT, bool NT, bool FL, bool BAND = false, typename AG = NoAfterGather>
__device__ __forceinline__ void stage_tile(const Params<V> &p, const Coord c0, const Coord c1, int tile, int num_tiles,
                                           const TileRegs<V, BLOCK, IPT> &regs, typename EndType<FL>::type *s_end_raw,
                                           V *s_prod_raw, int last_full_nz, int last_full_ro, unsigned *s_flag, const V *s_x = nullptr, int tid_in = -1,
                                           bool lean = false, AG after_gather = AG())
{
    const bool interior = tile_is_interior<IPT>(c0, c1, tile, num_tiles, last_full_nz, last_full_ro);      // block-uniform
    if constexpr (BAND) {              // (a banded pass is for an x beyond L2: never the LDS copy)
        if (interior) stage_tile_interior<V, BLOCK, IPT, NT, FL, false, true>(p, c0, c1, regs, s_end_raw, s_prod_raw, s_flag, nullptr, tid_in);
        else stage_tile_careful<V, BLOCK, IPT, NT, FL, false, true>(p, c0, c1, regs, s_end_raw, s_prod_raw, last_full_nz, last_full_ro, s_flag, nullptr, tid_in);
    } else if (MSPMV_LIKELY((layout_hints<V, IPT>()), s_x == nullptr)) {
        // (the likelihoods order the code of the SMALL tile shapes: x in memory, interior tile, usable hints, lean reduction first and in
        //  one piece -- a block that runs alone, i.e. a small problem, waits for every stretch of instructions it jumps to:
        //  instruction-cache misses per 28-tile launch 169 -> 87, wave-cycles -21 %, profiles/r04_small_call_counters.txt; a full
        //  device has them all in its caches, and the large fp32 shape sits at its register limit: the hints cost it spills)
        if (MSPMV_LIKELY((layout_hints<V, IPT>()), interior)) stage_tile_interior<V, BLOCK, IPT, NT, FL, false, false, AG>(p, c0, c1, regs, s_end_raw, s_prod_raw, s_flag, nullptr, -1, lean, after_gather);
        else stage_tile_careful<V, BLOCK, IPT, NT, FL, false, false, AG>(p, c0, c1, regs, s_end_raw, s_prod_raw, last_full_nz, last_full_ro, s_flag, nullptr, -1, lean, after_gather);
    } else {                                  // block-uniform: x lives in LDS (tiny x only)
        if (MSPMV_LIKELY((layout_hints<V, IPT>()), interior)) stage_tile_interior<V, BLOCK, IPT, NT, FL, true, false, AG>(p, c0, c1, regs, s_end_raw, s_prod_raw, s_flag, s_x, -1, lean, after_gather);
        else stage_tile_careful<V, BLOCK, IPT, NT, FL, true, false, AG>(p, c0, c1, regs, s_end_raw, s_prod_raw, last_full_nz, last_full_ro, s_flag, s_x, -1, lean, after_gather);
    }
}

// x -> LDS (dynamic shared memory, p.x_lds entries) at block start; nullptr when the call does not use it.
// The copy is visible after the block's first barrier.
template <typename V>
__device__ __forceinline__ const V *stage_x_in_lds(const Params<V> &p, unsigned char *s_dyn, int block)
{
    if (p.x_lds <= 0) return nullptr;
    V *s_x = reinterpret_cast<V *>(s_dyn);
    for (int i = threadIdx.x; i < p.x_lds; i += block) s_x[i] = p.x[i];
    return s_x;
}

// The same copy in two steps (tile_kernel_snap, scalar-hint shapes): the loads are REQUESTED at the very head of the block --
// before the hints are waited for, before the tile's streams -- and WRITTEN to LDS after the streams have been requested
// (vector loads return in order: waiting for the oldest does not wait for the streams).  In one step the block sat through
// hint latency + x latency with nothing else in flight: 0.5 us of a 5.7 us block life on the reference's --dense inputs.
template <typename V, int BLOCK> struct XRegs { static constexpr int N = X_LDS_MAX_BYTES / (int) sizeof(V) / BLOCK; V v[N]; };
template <typename V, int BLOCK>
__device__ __forceinline__ void request_x_for_lds(const Params<V> &p, XRegs<V, BLOCK> &xr)
{
    static_assert(XRegs<V, BLOCK>::N >= 1, "x_lds entries per thread");
#pragma unroll
    for (int j = 0; j < XRegs<V, BLOCK>::N; ++j) {
        const int i = (int) threadIdx.x + j * BLOCK;
        xr.v[j] = i < p.x_lds ? p.x[i] : (V) 0;
    }
}
template <typename V, int BLOCK>
__device__ __forceinline__ void commit_x_to_lds(const Params<V> &p, const XRegs<V, BLOCK> &xr, unsigned char *s_dyn)
{
    V *s_x = reinterpret_cast<V *>(s_dyn);
#pragma unroll
    for (int j = 0; j < XRegs<V, BLOCK>::N; ++j) {
        const int i = (int) threadIdx.x + j * BLOCK;
        if (i < p.x_lds) s_x[i] = xr.v[j];
    }
}

// Block -> tile mapping of the one-tile-per-block launches.  Blocks are dealt round-robin to the 8
// XCDs (block b runs on XCD b % 8: observed; only speed depends on it).  Inside every group of 8*G
// blocks, XCD k takes G CONSECUTIVE tiles: neighbouring tiles gather neighbouring x (which then stays in
// that XCD's private L2) and each XCD reads longer contiguous pieces of the CSR arrays, while the XCDs
// still advance through the matrix together.  Bijective for any num_tiles; log2 G = 0 turns it off.
constexpr int TILE_MAP_CONTIGUOUS = 30;       // chunk_log2 value selecting one contiguous tile range per XCD
__device__ __forceinline__ int xcd_chunked_tile(int b, int num_tiles, int chunk_log2)
{
    if (chunk_log2 == TILE_MAP_CONTIGUOUS) {
        // XCD k takes tiles [k*T/8, (k+1)*T/8): what the band-major prepared plan wants (each XCD's L2
        // then only ever sees the x slice of the band(s) its range covers); bijective for any T
        const int q = num_tiles / 8, r = num_tiles % 8;
        const int xcd = b & 7, idx = b >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if (chunk_log2 > 0) {
        // groups of span = 8 * 2^chunk_log2 blocks; shifts, not divisions (the compiler does not see that span is a power of
        // two, and this sits at the very head of every block, before its first load can be requested); b, num_tiles >= 0
        const int sh = chunk_log2 + 3;
        if (b < ((num_tiles >> sh) << sh)) {
            const int q = b >> sh, r = b & ((1 << sh) - 1);
            return (((q << 3) | (r & 7)) << chunk_log2) + (r >> 3);
        }
    }
    return b;
}

// the same map without a branch (the head of tile_kernel_snap: scalar selects, nothing between the wave's start and its hint request)
__device__ __forceinline__ int xcd_chunked_tile_flat(int b, int num_tiles, int chunk_log2)
{
    const int q8 = num_tiles >> 3, r8 = num_tiles & 7, xcd = b & 7;
    const int contiguous = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int sh = (chunk_log2 & 15) + 3, cl = chunk_log2 & 15;                     // (& 15: the shifts stay defined when chunk_log2 is the contiguous code)
    const int q = b >> sh, r = b & ((1 << sh) - 1);
    const int chunked = b < ((num_tiles >> sh) << sh) ? (((q << 3) | (r & 7)) << cl) + (r >> 3) : b;
    return chunk_log2 == TILE_MAP_CONTIGUOUS ? contiguous : chunk_log2 > 0 ? chunked : b;
}

// The compact launches (small problems: up to 2304 tiles): ONE CONTIGUOUS TILE RANGE PER XCD (map != 0), the same range in every call.
// Blocks are dealt round-robin to the XCDs (block b -> XCD b % 8), so with tile = block index neighbouring tiles sit on eight different
// XCDs and each fetches the x window of its rows for itself (a 5-point grid of 500^2: 10.9 KB of x beside the 17 KB of the tile's own
// streams); with one range per XCD the window is fetched once per XCD, the lines that straddle tile boundaries once instead of twice
// -- and a matrix of up to ~4 MB per XCD appears to be still in that XCD's L2 at the next call of a solver's loop (where the advantage
// ends: 2300-2800 tiles, 4-6 MB per XCD; mspmv_api.hip: compact_max_tiles).  Measured (tools/ab_driver, interleaved with the block-index map, fp64 / fp32 5-point grids):
// 697 tiles 4.35 -> 4.13 / 3.29 -> 3.27 us per call, 1003 tiles 5.31 -> 4.69 / 3.75 -> 3.53, 1366 tiles 7.05 -> 6.09 / 4.64 -> 4.07.
// Forward progress is not a matter of order here either: a tile that needs another workgroup's record (a long row ending in it) polls a
// bounded number of times and then computes the sum itself; inside a range a lower-numbered tile sits on an earlier block, only the
// first tiles of a range (and group leaders at a range's edge) look at blocks dispatched after them.
__device__ __forceinline__ int compact_tile(int b, int num_tiles, int map)
{
    const int q8 = num_tiles >> 3, r8 = num_tiles & 7, xcd = b & 7;
    const int contiguous = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    return map ? contiguous : b;                  // (scalar arithmetic and a select: nothing between the wave's start and its hint request)
}

// Clock-scheduled column bands (mspmv_tdm.hpp): the one-pass form of the same organisation.  band_shift == 0: the passes.
struct TdmArgs {
    int band_shift;        // band = column >> band_shift
    int bands;             // ceil(cols / 2^band_shift) <= TDM_MAX_BANDS
    float inv_slot;        // 1 / (ticks of the 100 MHz clock a band stays on air)
    int lookahead;         // bands after the one on air that may be taken too
};
template <typename V, int BLOCK, int IPT, bool NT>
__device__ __forceinline__ void stage_tile_tdm(const Params<V> &p, const Coord c0, const Coord c1, const TileRegs<V, BLOCK, IPT> &regs,
                                               end16_t *s_end_raw, V *s_prod_raw, int last_full_nz, int last_full_ro, unsigned *s_flag,
                                               int *s_start, int *s_wave_sum, const TdmArgs &ta, int tid);
constexpr int TDM_MAX_BANDS = 32;

// What the tile kernel needs to know about them (BAND variants of tile_kernel_vec; verdict == nullptr otherwise)
struct BandArgs {
    const int *verdict;        // BAND_WINDOWS verdicts of band_detect_block
    int *counters;             // 8 claim counters, BAND_COUNTER_STRIDE ints apart, zeroed by band_detect_block
    int *next;                 // num_tiles ints: next[t] = the tile the block that ran t in pass 0 took after it
    int grid;                  // blocks that run the passes (4-5 per CU, a multiple of 8); the others return
    int bands, band_cols;
    int force;                 // 1: passes whatever the verdicts say (mspmv_set_band_passes)
    TdmArgs tdm;               // band_shift > 0: clock-scheduled bands instead of the passes
};

// The column-band passes of one call, run by the first `grid` blocks of the tile kernel's launch.
// Pass 0 hands out the tiles dynamically -- as the hardware's block dispatcher does for the one-tile-per-block form; a
// static split measured 18 % slower, and so did anything that lets the blocks work far apart in the matrix -- in
// ascending order: block b starts with tile b, then takes tile 8 * n + (b & 7) for the next n of counter (b & 7)
// (8 counters 256 bytes apart: one device-scope atomic per tile on ONE word serialises at ~10 ns, 0.3 ms for C2;
// b & 7 is the block's XCD, and when its own sequence runs out a block helps with the others').  The claim for the
// next tile is issued before the current tile is staged.  Each block chains what it took in next[] and the later
// passes walk the block's own chain, so y[r] and carries[tile] are re-read by the very thread that wrote them: nothing
// travels between CUs or XCDs (next[] is written and read by thread 0 of the same block, past the caches).
// Pass 0 applies the caller's alpha and beta, the later ones add to y and to the stored carries; the ordinary fix-up
// launch follows.  Blocks drift from one band to the next without a barrier: two slices share L2 only while they do.
template <typename V, int BLOCK, int IPT, bool NT>
__device__ __forceinline__ void run_band_passes(Params<V> p, const Coord *__restrict__ coords, Carry<V> *__restrict__ carries,
                                                int num_tiles, const BandArgs &ba, end16_t *s_end_raw, V *s_prod_raw, unsigned *s_flag,
                                                int *s_wave_key, V *s_wave_val)
{
    constexpr int CPT = IPT / 4 + 1;
    constexpr int SLOTS = CPT * BLOCK * 4;
    __shared__ int s_next, s_first;
    const int tid = threadIdx.x;
    if (tid < SLOTS / 32 + 1) s_flag[tid] = 0u;
    // (the block's index is needed again at the head of every pass; kept in LDS, not in a register across the tile loop -- the
    //  fp32 kernel sits at its 64-VGPR cap there, and what does not fit went to scratch memory: .vgpr_spill_count 1-2 until round 6)
    __shared__ V s_beta0;
    if (tid == 0) { s_first = blockIdx.x; s_beta0 = p.beta; }
    const int last_full_nz = (p.nnz & ~3) - 4;
    const int last_full_ro = ((p.rows + 1) & ~3) - 4;
    const int first_n = ba.grid / 8;           // sequence positions the blocks' first tiles used up (grid: a multiple of 8, or < 8 = all the tiles)
    int seq = (int) blockIdx.x & 7;            // thread 0: the sequence it claims from (moves on when one is exhausted)
    for (int b = 0; b < ba.bands; ++b) {
        p.band_lo = b * ba.band_cols; p.band_len = ba.band_cols; p.band_pass = b;
        __syncthreads();
        if (tid == 0) s_next = s_first;        // every pass starts with the block's own first tile: no claim
        __syncthreads();
        p.beta = b == 0 ? s_beta0 : (V) 1;
        for (;;) {
            // (s_next: written before the barrier that ended the previous tile, or the one above)
            const int tile = s_next;
            if (tile >= num_tiles) break;
            int following = num_tiles;
            if (tid == 0) {
                // the tile after this one: the atomic / the load is in flight while this tile is staged
                if (b > 0) following = __hip_atomic_load(ba.next + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (ba.grid >= 8)
                    for (int tries = 0; tries < 8; ++tries) {
                        const int n = first_n + atomicAdd(ba.counters + seq * BAND_COUNTER_STRIDE, 1);
                        following = 8 * n + seq;
                        if (following < num_tiles) break;
                        following = num_tiles; seq = (seq + 1) & 7;
                    }
            }
            // the thread index goes through an empty asm once per tile: what the tile body derives from it (LDS addresses,
            // lane offsets) is then recomputed per tile instead of being hoisted out of the loop and kept live across it,
            // which costs ~40 registers per lane and with them a quarter of the resident waves.
            int t = tid;
            asm volatile("" : "+v"(t));
            __builtin_assume(t >= 0 && t < BLOCK);
            const Coord c0 = coords[tile];
            const Coord c1 = coords[tile + 1];
            TileRegs<V, BLOCK, IPT> regs;
            issue_nonzero_loads<V, BLOCK, IPT, NT, !band_lazy_values<V, true>()>(p, c0, c1, regs, t);     // (fp64: columns now, values by band in the staging)
            stage_tile<V, BLOCK, IPT, NT, true, true>(p, c0, c1, tile, num_tiles, regs, s_end_raw, s_prod_raw, last_full_nz, last_full_ro, s_flag, nullptr, t);
            const int pshift = c0.y - (c0.y & ~3);
            const int eshift = (c0.x + 1) - ((c0.x + 1) & ~3);
            consume_tile_flags<V, BLOCK, IPT, true>(p, c0, c1.x - c0.x, c1.y - c0.y, s_end_raw + eshift, s_prod_raw, s_flag,
                                                    s_wave_key, s_wave_val, carries + tile, pshift, nullptr, nullptr, 0, false, 0, t);
            if (tid == 0) {
                s_next = following;
                if (b == 0) __hip_atomic_store(ba.next + tile, following, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();          // all LDS reads of this tile done before the next tile's staging writes; s_next visible
        }
    }
}

#ifdef MSPMV_DEV
// development: per-phase cycle stamps of the first 16 tiles of every block (ABLATE == 6 variant)
__device__ unsigned long long *g_mspmv_trace = nullptr;
#endif

template <typename V, int BLOCK, int IPT, bool AXPBY, bool XCD_REMAP, bool NT, int ABLATE = 0, bool PERSIST = false, bool BAND = false, bool TDM = false>
__global__ __launch_bounds__(BLOCK, (tile_waves_per_simd<V, BLOCK, IPT, !PERSIST, ABLATE == 7>())) void tile_kernel_vec(Params<V> p, const Coord *__restrict__ coords,
                                                                Carry<V> *__restrict__ carries, int num_tiles, int xcd_chunk_log2,
                                                                BandArgs ba)
{
    constexpr int NW = BLOCK / WAVE;
    constexpr int CPT = IPT / 4 + 1;
    constexpr int SLOTS = CPT * BLOCK * 4;         // >= TILE + 8
    constexpr bool FL = ABLATE != 7;

    __shared__ __attribute__((aligned(16))) typename EndType<FL>::type s_end_raw[SLOTS];
    __shared__ __attribute__((aligned(16))) V s_prod_raw[SLOTS];
    __shared__ unsigned s_flag[SLOTS / 32 + 1];
    __shared__ int s_wave_key[NW];
    __shared__ V s_wave_val[NW];               // ABLATE 7 (development): the per-thread path walk instead
    // BAND: a call that the column-band passes may serve better; the sampled windows decide, here, on the device.  The
    // verdicts are requested now and looked at once the tile's coordinates are there too (one load latency, not two:
    // with the residency capped by LDS every microsecond a block waits before streaming is throughput lost)
    int band_v = 0;
    if constexpr (BAND) {
        static_assert(FL && !PERSIST && ABLATE == 0 && !XCD_REMAP, "band passes: production variant only");
    static_assert(BAND || !TDM, "clock-scheduled bands: a BAND variant");
        band_v = ba.force ? 1 : ba.verdict[threadIdx.x & (WAVE - 1)];
    }
    constexpr bool TRACE = ABLATE == 6;
    int trace_iter = 0;
#ifdef MSPMV_DEV
    unsigned long long *const trace = TRACE ? g_mspmv_trace : nullptr;
#else
    static_assert(ABLATE == 0 && !PERSIST && !XCD_REMAP, "development variants need -DMSPMV_DEV");
    unsigned long long *const trace = nullptr;
#endif
#define MSPMV_TR(i) do { if (TRACE && trace && threadIdx.x == 0 && trace_iter < 16) trace[((size_t) blockIdx.x * 16 + trace_iter) * 8 + (i)] = clock64(); } while (0)

    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];     // x, when it is tiny (p.x_lds)
    const int tid = threadIdx.x;
    if (tid < SLOTS / 32 + 1) s_flag[tid] = 0u;
    const V *const s_x = stage_x_in_lds<V>(p, s_dyn, BLOCK);
    __syncthreads();
    // XCD_REMAP: blocks are dealt round-robin to the 8 XCDs (block b -> XCD
    // b % 8, observed; only speed depends on it); give each XCD's private L2 a
    // contiguous range of tiles.  Bijective for any num_tiles.
    auto physical = [&](int t) {
        if (XCD_REMAP) {
            const int q = num_tiles / 8, r = num_tiles % 8;
            const int xcd = t % 8, idx = t / 8;
            return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        return xcd_chunked_tile(t, num_tiles, xcd_chunk_log2);
    };
    int seq = blockIdx.x;
    if (seq >= num_tiles) return;
    int tile = physical(seq);
    Coord c0 = coords[tile];
    Coord c1 = coords[tile + 1];
    if constexpr (BAND) {
        // (the empty asm pins the order: coordinates requested, THEN the verdict awaited -- left alone the compiler
        // compares it the moment it is loaded, a full memory latency at the head of every block.  Measured on a banded
        // 99 M-nonzero matrix the passes do not serve: +8 us per call that way, +2 us this way.)
        asm volatile("" : "+s"(c0.x), "+v"(band_v));
        if (__popcll(__ballot(band_v != 0)) >= BAND_MAJORITY) {
            if constexpr (TDM) {
                // clock-scheduled column bands (mspmv_tdm.hpp): one tile per block as in the ordinary body, the x gathers band by band
                // (a kernel of its own, chosen by the host: with the passes in the same kernel the scalar registers do not go round)
                __shared__ int s_tdm_start[TDM_MAX_BANDS + 1];
                p.x_lds = 0;
                TileRegs<V, BLOCK, IPT> tregs;
                // (values by plain 16-byte loads, also where the ordinary body fetches them line by line: the two bodies then share no
                //  loads, the ordinary one compiles as it does in the kernel of the passes, and this one needs no lane swaps)
                asm volatile("" : "+s"(p.nnz));       // (nothing of this body is merged with the ordinary one's)
                issue_nonzero_loads<V, BLOCK, IPT, NT, true, false>(p, c0, c1, tregs);
                stage_tile_tdm<V, BLOCK, IPT, NT>(p, c0, c1, tregs, s_end_raw, s_prod_raw, (p.nnz & ~3) - 4, ((p.rows + 1) & ~3) - 4, s_flag,
                                                  s_tdm_start, s_wave_key, ba.tdm, (int) threadIdx.x);
                consume_tile_flags<V, BLOCK, IPT, AXPBY>(p, c0, c1.x - c0.x, c1.y - c0.y, s_end_raw + ((c0.x + 1) - ((c0.x + 1) & ~3)), s_prod_raw, s_flag,
                                                         s_wave_key, s_wave_val, carries + tile, c0.y - (c0.y & ~3));
                return;
            } else {
                // the passes instead: run by the first ba.grid blocks of this launch, the others return
                if ((int) blockIdx.x < ba.grid) {
                    if (!AXPBY) { p.alpha = (V) 1; p.beta = (V) 0; }
                    p.x_lds = 0;
                    run_band_passes<V, BLOCK, IPT, NT>(p, coords, carries, num_tiles, ba, s_end_raw, s_prod_raw, s_flag, s_wave_key, s_wave_val);
                }
                return;
            }
        }
    }
    TileRegs<V, BLOCK, IPT> regs;
    issue_nonzero_loads<V, BLOCK, IPT, NT>(p, c0, c1, regs);
    const int last_full_nz = (p.nnz & ~3) - 4;
    const int last_full_ro = ((p.rows + 1) & ~3) - 4;   // rows + 1 >= 4

    for (;;) {
        const int tile_rows = c1.x - c0.x;
        const int tile_nnz = c1.y - c0.y;
        const int a0 = c0.y & ~3;
        const int pshift = c0.y - a0;
        const int first = c0.x + 1;                // d_row_offsets index of the tile's first row end
        const int i0 = first & ~3;
        const int eshift = first - i0;

        // ---- next tile's coordinates (scalar loads, in flight during the staging)
        const int next_seq = seq + (int) gridDim.x;
        const bool has_next = PERSIST ? next_seq < num_tiles : false;
        const int next = has_next ? physical(next_seq) : tile;
        const Coord n0 = coords[next];
        const Coord n1 = coords[next + 1];
        MSPMV_TR(0);
        if (TRACE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MSPMV_TR(1);
        stage_tile<V, BLOCK, IPT, NT, FL>(p, c0, c1, tile, num_tiles, regs, s_end_raw, s_prod_raw, last_full_nz, last_full_ro, s_flag, s_x);
        MSPMV_TR(2);
        // ---- the next tile's nonzero stream goes in flight, then the LDS phases of this tile
        if (has_next) issue_nonzero_loads<V, BLOCK, IPT, NT>(p, n0, n1, regs);
        MSPMV_TR(3);
        if constexpr (ABLATE == 1) {
            // ablation (development): staging only -- keep the LDS data live, skip search/walk/scan
            if (s_prod_raw[tid] == (V) 12345.678 && s_end_raw[tid] == 77) carries[tile].key = 1;
        } else if constexpr (FL)
            consume_tile_flags<V, BLOCK, IPT, AXPBY>(p, c0, tile_rows, tile_nnz, s_end_raw + eshift, s_prod_raw, s_flag,
                                                     s_wave_key, s_wave_val, carries + tile, pshift,
                                                     (TRACE && trace && trace_iter < 16) ? trace + ((size_t) blockIdx.x * 16 + trace_iter) * 8 : nullptr);
        else
            consume_tile_lds<V, BLOCK, IPT, AXPBY, true>(p, c0, tile_rows, tile_nnz, s_end_raw + eshift, s_prod_raw,
                                                            s_prod_raw, s_wave_key, s_wave_val, carries + tile, pshift);
        MSPMV_TR(4);
        if (!has_next) break;
        __syncthreads();          // all LDS reads of this tile done before the next tile's staging writes
        MSPMV_TR(5);
        ++trace_iter;
        seq = next_seq; tile = next; c0 = n0; c1 = n1;
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
__device__ __forceinline__ void fixup_chunk(const Carry<V> *in, int n, int chunk, bool multi, Carry<V> *out, V *y,
                                            int rows, V alpha, int *s_wave_key, V *s_wave_val)
{
    constexpr int CHUNK = BLOCK * IPT;
    const int tid = threadIdx.x;
    const int base = chunk * CHUNK;
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
    // Segments this thread closes are collected (slot k = item index that closed them) and
    // applied afterwards in one batch of independent loads followed by the stores: a
    // load-add-store per segment in sequence made each thread wait for up to IPT dependent
    // memory round trips.
    int ekey[IPT]; V esum[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        ekey[k] = -1; esum[k] = 0;
        if (keys[k] != cur) {
            if (!have_first) { have_first = true; first_key = cur; first_total = total; }
            else { ekey[k] = cur; esum[k] = total; }
            cur = keys[k]; total = vals[k];
        } else total += vals[k];
    }
    int prev_key, agg_key; V carry_in, agg_val;
    block_exclusive_rbk<V, BLOCK>(cur, total, s_wave_key, s_wave_val, prev_key, carry_in, agg_key, agg_val);
    int fkey = -1; V fsum = 0;
    if (have_first) { fkey = first_key; fsum = first_total + ((tid > 0 && prev_key == first_key) ? carry_in : (V) 0); }
    // threads whose whole range is one key contributed through the scan only.
    int lkey = -1; V lsum = 0;                         // the chunk's open last segment (last thread)
    if (tid == BLOCK - 1) {
        if (multi) {
            Carry<V> c; c.key = agg_key; c.value = agg_val; out[2 * chunk + 1] = c;
            if (agg_key == key0) { Carry<V> z; z.key = key0; z.value = 0; out[2 * chunk] = z; }
        } else { lkey = agg_key; lsum = agg_val; }
    }
    // a segment with the chunk's first key may continue from the previous chunk: next level
    if (multi && fkey == key0) { Carry<V> c; c.key = fkey; c.value = fsum; out[2 * chunk] = c; fkey = -1; }
#pragma unroll
    for (int k = 0; k < IPT; ++k)
        if (multi && ekey[k] == key0) { Carry<V> c; c.key = ekey[k]; c.value = esum[k]; out[2 * chunk] = c; ekey[k] = -1; }
    // batched y[key] += alpha * sum (keys of one thread are distinct)
    V old[IPT], fold = 0, lold = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) { const bool ok = ekey[k] >= 0 && ekey[k] < rows; old[k] = ok ? y[ekey[k]] : (V) 0; }
    if (fkey >= 0 && fkey < rows) fold = y[fkey];
    if (lkey >= 0 && lkey < rows && lkey != fkey) lold = y[lkey];
#pragma unroll
    for (int k = 0; k < IPT; ++k) if (ekey[k] >= 0 && ekey[k] < rows) y[ekey[k]] = old[k] + alpha * esum[k];
    if (fkey >= 0 && fkey < rows) { fold += alpha * fsum; if (lkey == fkey) fold += alpha * lsum; y[fkey] = fold; }
    if (lkey >= 0 && lkey < rows && lkey != fkey) y[lkey] = lold + alpha * lsum;
}

template <typename V, int BLOCK, int IPT>
__global__ __launch_bounds__(BLOCK) void fixup_kernel(const Carry<V> *__restrict__ in, int n,
                                                      Carry<V> *__restrict__ out, V *__restrict__ y, int rows,
                                                      V alpha)
{
    constexpr int NW = BLOCK / WAVE;
    __shared__ int s_wave_key[NW];
    __shared__ V s_wave_val[NW];
    fixup_chunk<V, BLOCK, IPT>(in, n, blockIdx.x, gridDim.x > 1, out, y, rows, alpha, s_wave_key, s_wave_val);
}

// ---------------------------------------------------------------------------
// One-launch fix-up.  Same reduce-by-key over the carry pairs, but every segment (maximal run of
// equal keys) is owned by the block whose chunk holds its FIRST pair: that block sums the part
// inside its chunk (thread-local fold + one block scan, as above) and, when the run continues
// past the chunk end, reads on until the key changes -- keys are non-decreasing, so the run is
// contiguous; this only happens for rows that span several tiles.  A block skips the leading
// pairs of its chunk that continue a run begun earlier.  Every y[key] therefore has exactly one
// writer and a fixed association order (bitwise reproducible), with no second level: one launch
// instead of two or three, ~5 us less on every call.
// ---------------------------------------------------------------------------
template <typename V, int BLOCK, int IPT>
__global__ __launch_bounds__(BLOCK) void fixup_onepass_kernel(const Carry<V> *__restrict__ in, int n, V *__restrict__ y,
                                                              int rows, V alpha)
{
    constexpr int CHUNK = BLOCK * IPT;
    constexpr int NW = BLOCK / WAVE;
    __shared__ int s_wave_key[NW];
    __shared__ V s_wave_val[NW];
    __shared__ int s_need;
    const int tid = threadIdx.x;
    const int base = blockIdx.x * CHUNK;
    const int key_before = base > 0 ? in[base - 1].key : -1;       // -1: no run continues into this chunk (keys are >= 0)

    int keys[IPT]; V vals[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int i = base + tid * IPT + k;
        if (i < n) { const Carry<V> c = in[i]; keys[k] = c.key; vals[k] = c.value; }
        else { keys[k] = 0x7fffffff; vals[k] = 0; }
    }
    const int first_i = base + tid * IPT;
    int cur = tid == 0 ? key_before : (first_i - 1 < n ? in[first_i - 1].key : 0x7fffffff);   // the run this thread starts inside
    if (tid == BLOCK - 1) {
        // does the chunk's last run continue in the next chunk, and is it ours (begun in this chunk)?
        const int last_key = keys[IPT - 1];
        const int next_i = base + CHUNK;
        s_need = (next_i < n && last_key != key_before && last_key < rows && in[next_i].key == last_key) ? 1 : 0;
    }
    V total = 0;
    int first_key = -1; V first_total = 0; bool have_first = false;
    int ekey[IPT]; V esum[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        ekey[k] = -1; esum[k] = 0;
        if (keys[k] != cur) {
            if (!have_first) { have_first = true; first_key = cur; first_total = total; }
            else { ekey[k] = cur; esum[k] = total; }
            cur = keys[k]; total = vals[k];
        } else total += vals[k];
    }
    int prev_key, agg_key; V carry_in, agg_val;
    block_exclusive_rbk<V, BLOCK>(cur, total, s_wave_key, s_wave_val, prev_key, carry_in, agg_key, agg_val);
    int fkey = -1; V fsum = 0;
    if (have_first) { fkey = first_key; fsum = first_total + ((tid > 0 && prev_key == first_key) ? carry_in : (V) 0); }
    if (fkey == key_before) fkey = -1;                 // that run began in an earlier chunk: its owner adds these pairs
    // the chunk's open last run (last thread); its continuation beyond the chunk, if any
    int lkey = -1; V lsum = 0;
    if (tid == BLOCK - 1 && agg_key != key_before) { lkey = agg_key; lsum = agg_val; }
    if (s_need) {                                      // block-uniform (written before the scan's barrier)
        const int akey = in[base + CHUNK].key;
        V part = 0;
        // 16 independent loads per thread and round (a round costs one memory latency whatever
        // its width): a run of 24 000 pairs -- one row spanning 24 000 tiles -- takes 6 rounds
        constexpr int U = 16;
        for (int pos = base + CHUNK;; pos += U * BLOCK) {
            Carry<V> c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = pos + tid + u * BLOCK;
                if (i < n) c[u] = in[i]; else { c[u].key = -2; c[u].value = 0; }
            }
            bool ended = false;
#pragma unroll
            for (int u = 0; u < U; ++u) { if (c[u].key == akey) part += c[u].value; else ended = true; }
            if (__syncthreads_or(ended ? 1 : 0)) break;
        }
        // block sum in a fixed order: wave scan, then the wave totals in wave order
        const V wsum = wave_segmented_inclusive_sum<V>(0, part);
        __syncthreads();                               // s_wave_val is free again
        if ((tid & (WAVE - 1)) == WAVE - 1) s_wave_val[tid / WAVE] = wsum;
        __syncthreads();
        if (tid == BLOCK - 1) {
            V ext = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) ext += s_wave_val[w];
            lsum += ext;
        }
    }
    // batched y[key] += alpha * sum (the keys of one thread are distinct; every key has one owner)
    V old[IPT], fold = 0, lold = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) { const bool ok = ekey[k] >= 0 && ekey[k] < rows; old[k] = ok ? y[ekey[k]] : (V) 0; }
    if (fkey >= 0 && fkey < rows) fold = y[fkey];
    if (lkey >= 0 && lkey < rows) lold = y[lkey];
#pragma unroll
    for (int k = 0; k < IPT; ++k) if (ekey[k] >= 0 && ekey[k] < rows) y[ekey[k]] = old[k] + alpha * esum[k];
    if (fkey >= 0 && fkey < rows) y[fkey] = fold + alpha * fsum;
    if (lkey >= 0 && lkey < rows) y[lkey] = lold + alpha * lsum;
}

// ---------------------------------------------------------------------------
// ONE launch for problems of any size: tile_kernel_snap (the default of every call that the vectorised staging can serve,
// and that is not a candidate for the column-band passes).  No coordinate launch,
// no fix-up launch, no dependency between workgroups except for rows longer than HEAD_MAX.
//  * COORDINATES ARE HINTS, VERIFIED.  The point (x, y) of a tile boundary is characterised by two row offsets:
//    row_offsets[x] <= y <= row_offsets[x + 1].  A tile therefore takes whatever the caller's temp storage holds for its
//    two boundaries -- left there by an earlier call on the same matrix, by mspmv_csrmv_prepare, or garbage --, starts
//    streaming on that assumption, and checks the four row offsets that decide it (requested first, so they are back
//    before the nonzeros).  Right (every call but the first on a given temp buffer): the tile has paid what a stored
//    coordinate costs, one cached load, and the call has no coordinate pass at all.  Wrong: waves 0 and 1 search the two
//    boundaries (wave_merge_path_guess: one window of row offsets around the arithmetic guess on a regular matrix, the
//    64-ary sampled search on any other), the tile is staged again, and the corrected hints are stored for the next call.
//    Nothing depends on the hints being right, so the call stays stateless in the reference's sense (temp storage is
//    scratch); what a second call on the same buffer saves is the search, 8-19 us per call as a launch of its own.
//  * ROW-SNAPPED TILES: no carries, every y[r] written exactly once.  A merge-path boundary falls inside some row x,
//    `head` nonzeros after the row's first one.  When head <= HEAD_MAX the boundary is moved back to the row's first
//    nonzero: the tile BEFORE it stops at the end of its last complete row and the tile AFTER it also multiplies those
//    <= HEAD_MAX nonzeros -- both sides decide from the same two numbers, so they agree.  Tiles stay merge-path tiles
//    (their work differs by at most HEAD_MAX items, 5-10 % of a tile; LDS has that much room in every compiled shape), but
//    a matrix of short rows -- every grid, band, dense-block or FEM matrix -- has no row left open at any tile end:
//    nothing to publish, nothing to wait for, nothing to fix up.
//  * Long rows (head > HEAD_MAX): the tiles holding a piece of the row publish their partial sums
//    as tagged records and the tile in which the row ENDS takes exactly those (up to 64 by wave 0 alone, longer lists by
//    the whole block, 8 records per thread and round), adds them in a fixed order and clears them.  A publisher never
//    waits before it publishes, so there are no chains; a waiting tile needs lower-numbered TILES to have been dispatched,
//    which the block -> tile mapping guarantees as long as one of its XCD runs fits the resident blocks twice over
//    (mspmv_api.hip: safe_chunk_log2) -- and otherwise recomputes the missing sum itself after a bounded poll, so the call
//    never depends on another workgroup's progress.  Deterministic and bitwise reproducible (a recomputed sum is the one
//    re-association that can differ): the result does not depend on whether the hints were right.
//  * Closed tiles of short rows take the lean row-by-row reduction (consume_tile_rows) instead of flags + scan.
// ---------------------------------------------------------------------------
// Merge-path point of a diagonal WITHOUT a coordinate pass and, on a regular matrix, at the price of the one memory round
// trip that loading a stored coordinate costs: the row is guessed arithmetically -- x ~ diagonal * rows / (rows + nnz),
// exact when every row has the same length -- and the wave looks at the 64 rows around the guess (two cache lines of
// row_offsets, which the tile is about to read anyway).  M(p) = row_end[p] + p + 1 is strictly increasing and the point
// is the first p with M(p) > diagonal, so the window holds the answer iff its first row fails the test (or the window
// starts at row 0) and some row passes it.  Otherwise the window's own slope gives a second guess (a matrix whose row
// lengths vary smoothly: grids, bands, FEM meshes), and if that window misses as well the 64-ary search over fixed
// samples runs (wave_merge_path_search_interp: any matrix, 3-4 dependent loads).  Exact; all lanes return the same
// point; row_start = row_offsets[x], the first nonzero of the row the point falls in.
__device__ __forceinline__ Coord wave_merge_path_guess(int diagonal, const int *__restrict__ row_end, int rows, int nnz, int &row_start)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int inf = rows + nnz + 1;                                  // < 2^31
    long long g = (long long) diagonal * rows / ((long long) rows + nnz);
    for (int attempt = 0; attempt < 2; ++attempt) {
        int w0 = (int) g - WAVE / 2;
        const int w_max = rows - (WAVE - 1) > 0 ? rows - (WAVE - 1) : 0;
        w0 = w0 < 0 ? 0 : w0 > w_max ? w_max : w0;
        const int p = w0 + lane;
        const int e = p < rows ? row_end[p] : 0;
        const int m = p < rows ? e + p + 1 : inf;                    // (p == rows: M = +inf)
        const unsigned long long mask = __ballot(m > diagonal);
        const bool first_fails = !(mask & 1ull);
        if (mask != 0ull && (first_fails || w0 == 0)) {
            const int f = __ffsll((long long) mask) - 1;
            const int x = w0 + f;                                    // <= rows
            // row_offsets[x] = row_end[x - 1]: in the window unless x == w0 (then w0 == 0: row 0 starts at 0)
            row_start = f > 0 ? __shfl(e, f - 1, WAVE) : 0;
            Coord c; c.x = x < rows ? x : rows; c.y = diagonal - c.x;
            if (x >= rows) row_start = nnz;
            return c;
        }
        if (mask == 0ull && w0 + WAVE >= rows) break;                // (cannot happen: lane of p == rows always passes) -- fall back
        // second guess from the window's slope
        const int m_a = __shfl(m, 0, WAVE), m_b = __shfl(m, WAVE - 1, WAVE);
        if (m_b >= inf || m_b <= m_a) break;
        g = w0 + (long long) ((double) (diagonal - m_a) * (double) (WAVE - 1) / (double) (m_b - m_a));
        if (g < 0) g = 0; if (g > rows) g = rows;
    }
    const Coord c = wave_merge_path_search_interp(diagonal, row_end, rows, nnz);
    row_start = c.x >= rows ? nnz : c.x > 0 ? row_end[c.x - 1] : 0;
    return c;
}

template <int BLOCK, int IPT>
constexpr int snap_head_max()
{
    constexpr int slack = (IPT / 4 + 1) * BLOCK * 4 - BLOCK * IPT - 16;
    return slack < 192 ? slack : 192;
}

// ---------------------------------------------------------------------------
// COMPACT FRONT END of the one-launch kernel, for problems of ONE BLOCK GENERATION (mspmv_api.hip: compact_max_tiles).
// The reference special-cases small problems too (dispatch_spmv_orig.cuh:674-679: one tile -> no search, no fix-up;
// agent_spmv_orig.cuh:867-891).  Below a few hundred tiles a call is one generation of blocks that each run alone on their CU:
// what it costs is the block's dependent memory trips (hints -> streams -> x) PLUS ONE INSTRUCTION PER ~5 CYCLES of a lone wave and
// every stretch of code the block has to fetch -- the general kernel issues ~440 instructions per wave on its likeliest path
// (hints, checks for every shape of tile, 16-bit row ends, row-start bits, the two-rows-at-a-time lean reduction) out of 19-37 KB
// of code.  This front end is that likeliest path and nothing else, written for instruction count (~200 per wave, ~2 KB of
// straight-line code at the head of the kernel):
//   * the tile is the block index (no XCD chunking: a generation that is resident all at once gains nothing from it);
//   * hints by two scalar loads; the tile is taken here only if they describe a CLOSED LEAN tile (exactly the tiles the general
//     kernel hands to consume_tile_rows) whose every speculative access stays inside the arrays and LDS -- anything else (no usable
//     hints yet, a long row, x in LDS, hints that fail the four-word verification) returns false BEFORE anything but LDS and registers
//     was touched, and the block runs the general body from its start: the compact kernel is complete, this is only its fast lane;
//   * CSR chunks are clamped to the last full chunk of their array (min, no branch); the one ragged chunk an array can end with is
//     re-aligned in a wave-uniform, rarely taken branch -- so the LAST tile of a problem stays in the fast lane (a launch is as slow
//     as its slowest block);
//   * row offsets go to LDS as they are (32 bits, x0 .. x1: a row's start and end are ONE ds_read2_b32), products at their raw
//     positions; ONE barrier;
//   * a thread sums a row left to right from +0.0 -- the first eight products under a shrinking EXEC mask (v_cmpx + v_add per
//     product instead of compare + select + add), nine to sixteen the same way in a second batch, longer rows by the 16 lanes of a
//     DPP row exactly as consume_tile_rows does: THE SAME ADDITIONS IN THE SAME ORDER, so y is bit for bit what the general kernel
//     writes (tests/test_gpu_parity.py: the `compact` paths; tools/fuzz.py).
// ---------------------------------------------------------------------------
constexpr int COMPACT_BLOCK = 256, COMPACT_IPT = 7;
// element i of a chunk that was loaded `s` elements too early (the last, ragged chunk of an array is fetched at n - 4): what
// belongs at position i is what was loaded at position i + s (positions that fall off the end are never used)
template <typename T>
__device__ __forceinline__ void compact_realign(Vec4<T> &c, int s)
{
    T l[4], o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) l[i] = c.get(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int j = i + s; o[i] = j <= 0 ? l[0] : j == 1 ? l[1] : j == 2 ? l[2] : l[3]; }
    if constexpr (sizeof(T) == 8) { c.a[0] = o[0]; c.a[1] = o[1]; c.b[0] = o[2]; c.b[1] = o[3]; }
    else { c.v[0] = o[0]; c.v[1] = o[1]; c.v[2] = o[2]; c.v[3] = o[3]; }
}
// (the ragged chunk sits at n - 4, which need not be 16-byte aligned: dword-aligned 16-byte global loads are fine on gfx9; the
//  packed attribute keeps the compiler from assuming more)
typedef int int4v_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float4v_u __attribute__((ext_vector_type(4), aligned(4)));
typedef double double2v_u __attribute__((ext_vector_type(2), aligned(8)));
typedef int int2v_u __attribute__((ext_vector_type(2), aligned(4)));

// loads through a 32-bit unsigned BYTE offset from a uniform base (global_load ... v_off, s[base]: one shift, no 64-bit address
// arithmetic per lane).  The dispatcher takes the compact kernel only when every array is < 4 GB (mspmv_api.hip).
template <typename T>
__device__ __forceinline__ T compact_ld(const void *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ Vec4<int> compact_ld4(const int *base, int e) { Vec4<int> r; r.v = compact_ld<int4v_u>(base, (unsigned) e << 2); return r; }
__device__ __forceinline__ Vec4<float> compact_ld4(const float *base, int e) { Vec4<float> r; r.v = compact_ld<float4v_u>(base, (unsigned) e << 2); return r; }
__device__ __forceinline__ Vec4<double> compact_ld4(const double *base, int e)
{
    Vec4<double> r; r.a = compact_ld<double2v_u>(base, (unsigned) e << 3); r.b = compact_ld<double2v_u>(base, ((unsigned) e << 3) + 16u); return r;
}

// Written for the ORDER OF ITS CODE as much as for its instruction count: the translation unit that instantiates the compact kernel
// (mspmv_compact.hip) is compiled with the block placement pass off, so the machine code keeps the order of this source -- the
// fast lane from the hint request to its s_endpgm in one piece, what is rarely needed (the ragged end of the nonzero arrays, tiles
// of more than 511 rows, rows of more than 8 nonzeros) behind it, reached by `goto` and left by `goto`.
template <typename V, bool AXPBY>
__device__ __forceinline__ void compact_front(const Coord *coords, const int *rstart, int num_tiles, const Params<V> &p,
                                              Carry<V> *__restrict__ carries, int lean_avg, int tile, V *s_prod, int *s_ro, int4v &hc, int2v &hr)
{
    constexpr int BLOCK = COMPACT_BLOCK, IPT = COMPACT_IPT, TILE = BLOCK * IPT, CPT = IPT / 4 + 1;
    constexpr int HEAD_MAX = snap_head_max<BLOCK, IPT>();
    constexpr int RO_ROUNDS = (TILE + 1 + BLOCK - 1) / BLOCK;         // row offsets x0 .. x1, one per lane and round: <= 8 rounds
    static_assert(CPT == 2 && TILE + HEAD_MAX + 16 <= CPT * BLOCK * 4 && RO_ROUNDS * BLOCK <= CPT * BLOCK * 4, "two chunks per thread; room for the snapped rows");
    static_assert(LEAN_SERIAL == 2 * LEAN_BATCH && LEAN_BATCH == 8, "two batches of eight");
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    // (everything the cold sections at the end touch is declared up here: a goto may not jump past an initialisation)
    Vec4<int> col[CPT]; Vec4<V> val[CPT];
    int rov[RO_ROUNDS];
    int r0, r, start, len; bool valid; V acc; const V *src;
    int2v vw0 = {0, 0}, vw1 = {0, 0};
    __shared__ int s_verdict;
    // ---- hints (scalar cache; the tile index is uniform).  Request and wait in ONE asm statement (see tile_kernel_snap).  They go
    // back to the caller: a tile that is not taken runs the general body on THESE registers instead of waiting for the same 24 bytes again
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(hc), "=&s"(hr) : "s"(coords + tile), "s"(rstart + tile) : "memory");
    // (used in this very basic block: the kernel-argument loads are requested before the wait above)
    asm volatile("" :: "s"(p.row_end), "s"(p.cols), "s"(p.values), "s"(p.x), "s"(p.y), "s"(p.rows), "s"(p.nnz), "s"(p.x_lds), "s"(lean_avg), "s"(carries));
    const int x0 = hc.x, x1 = hc.z, rs0 = hr.x, rs1 = hr.y;
    const int total = p.rows + p.nnz;                                  // < 2^31
    const int d0 = tile * TILE, d1 = d0 + TILE < total ? d0 + TILE : total;
    const int y0 = d0 - x0, y1 = d1 - x1;
    const int tile_rows = x1 - x0, tile_nnz = rs1 - rs0;
    // A closed lean tile (exactly the general kernel's `lean`: both boundaries snap, nonzeros <= lean_avg * rows) -- judged on the HINTS,
    // which may hold anything: what keeps every access before the verdict inside the arrays is 0 <= x0, x1 <= rows and 0 <= rs0 <= nnz
    // (chunk addresses are clamped from above, row-offset indices lie in [x0, x1], LDS is written at thread positions only); the
    // verdict then either proves the hints to be THE merge-path points of d0 and d1 -- and with them everything else assumed here:
    // x0 <= x1, rs0 <= rs1 <= nnz, the tile's sizes -- or sends the block to the general body before LDS is read or y written.
    // (x in memory and more than one tile: the dispatcher launches this variant only then)
    const bool take = ((unsigned) (y0 - rs0) <= (unsigned) HEAD_MAX) & ((unsigned) (y1 - rs1) <= (unsigned) HEAD_MAX) &
                      ((unsigned) x0 <= (unsigned) p.rows) & ((unsigned) x1 <= (unsigned) p.rows) & ((unsigned) rs0 <= (unsigned) p.nnz) &
                      ((unsigned) tile_nnz <= (unsigned) lean_avg * (unsigned) tile_rows);
    // ONE way out to the general body (the caller's code behind this function): `go` stays 0 when the tile is not taken or its hints
    // fail the verification, and is tested once, after the staging -- two exits would be merged by the compiler through flag
    // variables and a common block at the far end of the kernel, which puts two far jumps into the fast lane
    int go = 0;
    const int *__restrict__ row_offsets = p.row_end - 1;
    const int a0 = rs0 & ~3;
    const int nz4 = p.nnz - 4;                                          // (nnz >= 4 on this path)
    const int a0c = a0 < nz4 ? a0 : nz4;
    const int n_ro = tile_rows + 1;
    const int wave_base = __builtin_amdgcn_readfirstlane(tid) & ~(WAVE - 1);
    const bool live1 = a0 + 4 * (wave_base + BLOCK) < rs1;             // wave-uniform (scalar): see the nonzero loads below
    if (__builtin_expect(!take, 0)) goto staged;
    // ---- the four row offsets that decide whether (x0, rs0), (x1, rs1) are the points of diagonals d0, d1: two scalar loads of the
    // pairs row_end[x - 1], row_end[x] (uniform addresses; row_end[-1] = row_offsets[0] exists, and a boundary at x == rows reads
    // the pair one lower), requested first, looked at in the shadow of the gathers
    // (by the block's FIRST WAVE alone: its verdict reaches the others through one LDS word behind the staging barrier that is there
    //  anyway -- two vector-memory requests and ~25 scalar instructions fewer in three of the four waves)
    if (wave_base == 0) {
        const int b0 = x0 - 1 < p.rows - 2 ? x0 - 1 : p.rows - 2, b1 = x1 - 1 < p.rows - 2 ? x1 - 1 : p.rows - 2;      // (rows >= 3)
        // (plain loads on uniform addresses of memory the kernel has not written: the compiler makes them s_load_dwordx2 and keeps
        //  track of them itself -- a hand-issued scalar load whose wait sits in another asm statement leaves its destination
        //  registers open to copies while the load is in flight)
        vw0 = *reinterpret_cast<const int2v_u *>(p.row_end + b0);
        vw1 = *reinterpret_cast<const int2v_u *>(p.row_end + b1);
    }
    // ---- the tile's nonzeros: 4-element chunks aligned in array index space, clamped to the array's last full chunk.
    // A WAVE whose second chunk lies past the tile altogether (a tile of 5-point rows fills 1500 of the 2048 slots: waves 2 and 3)
    // issues nothing for it -- no stream loads, no gathers, no products: a redirected load costs no bytes, but it costs the CU's
    // vector-memory pipeline the same cycles, and that pipeline is what several tiles on one CU queue on
    {
        const int e = a0 + 4 * tid;
        const int ee = e < rs1 ? (e < nz4 ? e : nz4) : a0c;             // (a chunk past the tile re-reads the tile's first: no bytes for data it does not use)
        col[0] = compact_ld4(p.cols, ee);
        val[0] = compact_ld4(p.values, ee);
    }
    if (live1) {
        const int e = a0 + 4 * (tid + BLOCK);
        const int ee = e < rs1 ? (e < nz4 ? e : nz4) : a0c;
        col[1] = compact_ld4(p.cols, ee);
        val[1] = compact_ld4(p.values, ee);
    }
    // ---- row offsets x0 .. x1 (a row's start and end both come from here), one per lane and round: no alignment, no ragged end
    rov[0] = compact_ld<int>(row_offsets, (unsigned) (x0 + (tid < n_ro ? tid : 0)) << 2);
    // (a round is taken by the waves that have entries in it: a tile of 300 rows has 45 in the second round, all in the first wave)
    if (wave_base + BLOCK < n_ro) rov[1] = compact_ld<int>(row_offsets, (unsigned) (x0 + (tid + BLOCK < n_ro ? tid + BLOCK : 0)) << 2);      // wave-uniform
    if (__builtin_expect(2 * BLOCK < n_ro, 0)) goto more_row_offsets;
have_row_offsets:
    // the one ragged chunk the nonzero arrays can end with was fetched at nnz - 4: its elements are put where they belong
    if (__builtin_expect(((rs1 + 3) & ~3) > p.nnz, 0)) goto ragged_end;
aligned:
    {
        // ---- x gathers
        V xv[CPT][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[0][i] = compact_ld<V>(p.x, (unsigned) col[0].get(i) * (unsigned) sizeof(V));
        if (live1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[1][i] = compact_ld<V>(p.x, (unsigned) col[1].get(i) * (unsigned) sizeof(V));
        }
        // ---- the verdict on the hints, in the shadow of the gathers (scalar: v_readlane).  Acted upon after the barrier -- a branch
        // here would hold the gathers back behind it --: until then nothing but registers and LDS is touched
        if (wave_base == 0) {
        const int before0 = x0 == p.rows ? vw0.y : vw0.x, at0 = vw0.y, before1 = x1 == p.rows ? vw1.y : vw1.x, at1 = vw1.y;
        // (x, rs) is the point of diagonal d  <=>  rs == row_offsets[x] <= y = d - x  and  (x == rows ? d == total : y <= row_offsets[x + 1])
        // (written with integer selects: the tests stay on the scalar unit); and the stored y's are the derived ones (what
        //  mspmv_debug_read_tiles reports)
        const int lim0 = x0 < p.rows ? at0 : (d0 == total ? 0x7fffffff : -1);
        const int lim1 = x1 < p.rows ? at1 : (d1 == total ? 0x7fffffff : -1);
        const bool verdict = ((x0 > 0 ? before0 : 0) == rs0) & (y0 <= lim0) & ((x1 > 0 ? before1 : 0) == rs1) & (y1 <= lim1) & (hc.y == y0) & (hc.w == y1);
        if (lane == 0) s_verdict = verdict ? 1 : 0;
        }
        // (nothing open at the end of a closed tile: what mspmv_debug_read_tiles reports.  By the last wave, here in the shadow of the
        //  gathers; a tile that fails the verdict has it overwritten by the general body)
        if (__builtin_amdgcn_readfirstlane(tid) >= BLOCK - WAVE) { if (tid == BLOCK - 1) { Carry<V> c; c.key = x1; c.value = (V) 0; carries[tile] = c; } }
        // ---- LDS: row offsets as they are, products at their raw positions (element e of the array -> slot e - a0)
        s_ro[tid] = rov[0];
        if (wave_base + BLOCK < n_ro) s_ro[tid + BLOCK] = rov[1];
#pragma unroll
        for (int k = 0; k < CPT; ++k)
            if (k == 0 || live1) {                                      // (slots of a chunk nobody staged are never a row's products: reads past a row's end are discarded)
                V prod[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) prod[i] = val[k].get(i) * xv[k][i];
                st_lds4(&s_prod[4 * (tid + k * BLOCK)], prod);
            }
        __syncthreads();
        go = s_verdict;
    }
staged:
    asm volatile("" : "+v"(go));                                       // (opaque: the test below is not threaded back into the two places `go` comes from)
    go = __builtin_amdgcn_readfirstlane(go);
    if (__builtin_expect(go == 0, 0)) return;                          // (every wave alike; the general body starts from scratch)
    // ---- row by row: consume_tile_rows' arithmetic (left to right from +0.0; > LEAN_SERIAL: 16-lane groups), one row per thread and round
    r0 = wave_base;                                                    // (the wave's first row of the round: scalar -- the compiler has to SEE that the test below is wave-uniform)
next_round:
    // (the wave's program ENDS here -- s_endpgm, not a return: a return value is merged with the general body's exit and the
    //  reduction then sits behind two far jumps.  The test has to be visibly wave-uniform for the same reason: an exit the compiler
    //  takes for divergent is routed through flag variables and a common exit block)
    if (r0 >= tile_rows) __builtin_amdgcn_endpgm();                    // this wave's rows are done
    r = r0 + lane;
    valid = r < tile_rows;
    {
        const int rr = valid ? r : 0;                                   // (row 0 exists: tile_rows > 0 here)
        start = s_ro[rr]; const int end = s_ro[rr + 1];
        len = valid ? end - start : 0;
        src = s_prod + (start - a0);                                   // (start - a0 + 15 < SLOTS for any lane)
        // THE PRODUCT READS ARE WHAT THE LANE WAITS FOR: lanes a row apart read 8-byte words 32 bytes (rows of 4) or 64 bytes (rows of 8)
        // apart, 4 or 2 distinct bank groups for the 16 lanes of an LDS cycle -- SQ_LDS_BANK_CONFLICT is 65 % of the LDS cycles of a
        // 5-point grid's call, and the LDS is busy for ~3/4 of such a kernel (profiles/r05_lds_counters.txt).  So a wave none of whose rows
        // has more than four products (wave-uniform: one ballot) reads and adds the first four only.  (A third level -- six -- for rows of
        // 5 and 6, whose strides conflict far less, measured no gain: profiles/r05_ab_half_batch.txt.)
        V v[LEAN_BATCH];
        const bool more = __ballot(len > LEAN_BATCH / 2) != 0ull;
#pragma unroll
        for (int j = 0; j < LEAN_BATCH / 2; ++j) v[j] = src[j];
        acc = (V) 0;
        if (more) {
#pragma unroll
            for (int j = LEAN_BATCH / 2; j < LEAN_BATCH; ++j) v[j] = src[j];
            CompactChain<V>::add8(acc, v, len, 0);
        } else {
            const V h[4] = {v[0], v[1], v[2], v[3]};
            CompactChain<V>::add4(acc, h, len);
        }
    }
    if (__builtin_expect(__ballot(len > LEAN_BATCH) != 0ull, 0)) goto longer_rows;
store_row:
    if (valid) {
        V *__restrict__ y = p.y + x0;
        if (AXPBY) y[r] = p.alpha * acc + (p.beta == (V) 0 ? (V) 0 : p.beta * y[r]);
        else y[r] = acc;
    }
    r0 += BLOCK;
    goto next_round;

    // ======== what is rarely needed ========
more_row_offsets:                                                      // tiles of more than 511 rows (rows that average < 3.5 nonzeros)
#pragma unroll
    for (int k = 2; k < RO_ROUNDS; ++k)
        if (k * BLOCK < n_ro) rov[k] = compact_ld<int>(row_offsets, (unsigned) (x0 + (tid + k * BLOCK < n_ro ? tid + k * BLOCK : 0)) << 2);
#pragma unroll
    for (int k = 2; k < RO_ROUNDS; ++k)
        if (k * BLOCK < n_ro) s_ro[tid + k * BLOCK] = rov[k];           // (the barrier of the fast lane follows)
    goto have_row_offsets;
ragged_end:
#pragma unroll
    for (int k = 0; k < CPT; ++k)
        if (k == 0 || live1) {
            const int e = a0 + 4 * (tid + k * BLOCK);
            const int s = (e < rs1 && e > nz4) ? e - nz4 : 0;
            if (__ballot(s != 0) != 0ull) { compact_realign(col[k], s); compact_realign(val[k], s); }
        }
    goto aligned;
longer_rows:
    {
        // (a lane index the compiler cannot hoist out of the round loop: everything derived from it -- the DPP groups' bookkeeping --
        //  stays down here instead of being computed ahead of the fast lane's loop)
        int lane_c = tid; asm volatile("" : "+v"(lane_c)); lane_c &= WAVE - 1;
        V w[LEAN_BATCH];
#pragma unroll
        for (int j = 0; j < LEAN_BATCH; ++j) w[j] = src[LEAN_BATCH + j];
        CompactChain<V>::add8(acc, w, len, LEAN_BATCH);
        // rows longer than LEAN_SERIAL: the code consume_tile_rows runs for them
        if (__ballot(len > LEAN_SERIAL) != 0ull) lean_long_rows<V>(s_prod, start - a0, len, lane_c, acc);
    }
    goto store_row;
}

template <typename V, int BLOCK, int IPT, bool AXPBY, bool NT, bool COMPACT = false>
__global__ __launch_bounds__(BLOCK, (COMPACT ? 4 : tile_waves_per_simd<V, BLOCK, IPT, true>())) void tile_kernel_snap(Coord *__restrict__ coords, int *__restrict__ rstart,
                                                           int num_tiles, int xcd_chunk_log2,
                                                           Params<V> p, Carry<V> *__restrict__ carries, LookBack lb, int lean_avg)
{
    constexpr int TILE = BLOCK * IPT;
    constexpr int NW = BLOCK / WAVE;
    constexpr int CPT = IPT / 4 + 1;
    constexpr int SLOTS = CPT * BLOCK * 4;
    constexpr int HEAD_MAX = snap_head_max<BLOCK, IPT>();
    static_assert(HEAD_MAX >= 64 && TILE + HEAD_MAX + 16 <= SLOTS, "room for the snapped rows");
    // (COMPACT: the fast lane's 32-bit row offsets and the general body's 16-bit row ends share one array -- a block uses one of the two)
    __shared__ __attribute__((aligned(16))) int s_end_words[COMPACT ? SLOTS : SLOTS / 2];
    end16_t *const s_end_raw = reinterpret_cast<end16_t *>(s_end_words);
    __shared__ __attribute__((aligned(16))) V s_prod_raw[SLOTS];
    __shared__ unsigned s_flag[SLOTS / 32 + 1];
    __shared__ int s_wave_key[NW];
    __shared__ V s_wave_val[NW];
    __shared__ int s_bnd[6];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];     // x, when it is tiny (p.x_lds)
    // COMPACT (small problems, mspmv_api.hip: compact_max_tiles): the fast lane for closed lean tiles on good hints comes first -- its own LDS layout
    // (32-bit row offsets) in an array of its own, the product array shared --; a tile it does not take runs the general body below
    // from the start, on the hints the front end loaded.  Same tile (compact_tile) in both.
    int4v front_hc = {0, 0, 0, 0}; int2v front_hr = {0, 0};
    if constexpr (COMPACT) {
        static_assert(BLOCK == COMPACT_BLOCK && IPT == COMPACT_IPT && !NT, "the compact front end is written for the small tile shape");
        compact_front<V, AXPBY>(coords, rstart, num_tiles, p, carries, lean_avg, compact_tile((int) blockIdx.x, num_tiles, xcd_chunk_log2), s_prod_raw, s_end_words,
                                front_hc, front_hr);      // (a tile it takes ends there)
    }

    const int tid = threadIdx.x;
#ifdef MSPMV_DEV
    // development (tools/trace_snap.py): 100 MHz wall-clock stamps of this block's phases, 8 words per block
    const unsigned long long t_entry = wall_clock64();           // (before the load of the trace pointer: that is a memory round trip)
    unsigned long long *const snap_tr = g_mspmv_trace ? g_mspmv_trace + (size_t) blockIdx.x * 16 : nullptr;     // (8 words of the block + 8 of the lean reduction)
#define MSPMV_SNAP_TR(i) do { if (snap_tr && tid == 0) snap_tr[i] = wall_clock64(); } while (0)
    if (snap_tr && tid == 0) { snap_tr[6] = __builtin_amdgcn_s_getreg(63492); snap_tr[7] = __builtin_amdgcn_s_getreg(6164); }
#else
#define MSPMV_SNAP_TR(i) do { } while (0)
#endif
#ifdef MSPMV_DEV
    if (snap_tr && tid == 0) snap_tr[0] = t_entry;
#endif
    static_assert(sizeof(coords) + sizeof(rstart) + sizeof(num_tiles) + sizeof(xcd_chunk_log2) <= 32, "the four leading arguments lie inside the 8 preloaded dwords");
    // THE FIRST FOUR ARGUMENTS -- all the hint request needs -- are in SGPRs when the wave starts (kernel-argument preload, 8 dwords:
    // the Makefile's -amdgpu-kernarg-preload-count=8), the tile index below is branch-free scalar arithmetic on them, and the other
    // arguments are pinned behind the hint request: the block's first memory round trip is the hints AND the rest of the kernel
    // arguments together, where the arguments (in three batches, as the compiler sank them to their uses) came first.
    const int tile = COMPACT ? compact_tile((int) blockIdx.x, num_tiles, xcd_chunk_log2) : xcd_chunked_tile_flat((int) blockIdx.x, num_tiles, xcd_chunk_log2);
    // the hints: the tile's two boundaries (x, y) and their row starts, read THROUGH THE SCALAR
    // CACHE -- the tile index is uniform, and a scalar load neither queues behind the vector-memory traffic of the CU's other
    // blocks nor needs an LDS hop to reach every wave: 2-6 % on matrices streamed from HBM (grid2d-4096, dense32, band5, C4;
    // same-box A/B in profiles/r03_scalar_hints.txt), and since round 4 in the small tile shape as well (two lanes' vector
    // loads + an LDS broadcast until then): 4.1-4.2 -> 3.8-4.0 us per call below 1 M nonzeros, 19.9 -> 18.0 us at 5.8 M, with
    // the rest of this prologue in 32-bit, branch-free scalar code.  (The hints were written by vector stores of an earlier
    // launch; the scalar cache is invalidated at every kernel start.)  Request and wait are ONE asm statement: the compiler never
    // sees destination registers whose load is still in flight, so it cannot copy, spill or re-assign them under the load (a
    // later, separate s_waitcnt left exactly that open).  What used to sit between the two -- clearing the flag words,
    // requesting a tiny x -- comes first; the block has nothing else to do until its hints are there anyway.
    const bool single = num_tiles == 1;                             // one tile: its boundaries are (0, 0) and (rows, nnz)
    int4v hint_c; int2v hint_r;
    if constexpr (COMPACT) { hint_c = front_hc; hint_r = front_hr; }          // (the compact front end's: it requested them)
    else asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                      : "=&s"(hint_c), "=&s"(hint_r) : "s"(coords + tile), "s"(rstart + tile) : "memory");
    // (uses in this very basic block: the loads of these arguments stay up here, requested before the wait above)
    asm volatile("" :: "s"(p.row_end), "s"(p.cols), "s"(p.values), "s"(p.x), "s"(p.y), "s"(p.rows), "s"(p.nnz), "s"(p.x_lds), "s"(lean_avg),
                 "s"(carries), "s"(lb.rec), "s"(lb.tag_a), "s"(lb.tag_b), "s"(lb.error), "s"(lb.call_tag), "s"(lb.max_polls), "s"(p.band_pass));
    if (tid < SLOTS / 32 + 1) s_flag[tid] = 0u;                     // (the row-start bits)
    // a tiny x goes to LDS: requested now, written after the streams have been requested
    XRegs<V, BLOCK> xr;
    const V *s_x = nullptr;
    if (MSPMV_UNLIKELY((layout_hints<V, IPT>()), p.x_lds > 0)) { request_x_for_lds<V, BLOCK>(p, xr); s_x = reinterpret_cast<const V *>(s_dyn); }
    // (With a tiny x being copied into LDS, every wave requests its share of the streams first and the barrier -- which waits for
    //  that copy -- comes after the requests: dense32 fp32 -3 %, fp64 -6.5 %; without the copy the early barrier is the better
    //  place, by 1-2 %)
    const bool late_barrier = s_x != nullptr;                       // block-uniform
    if constexpr (!layout_hints<V, IPT>()) { if (!late_barrier) __syncthreads(); }      // (the large shapes: see below)
    const LookBack &lbe = lb;
    MSPMV_SNAP_TR(1);
    const int total = p.rows + p.nnz;                               // < 2^31
    // (32-bit: tile < num_tiles = ceil(total / TILE), so tile * TILE < total <= 2^31 - 65537 and adding TILE cannot overflow)
    const int d0 = tile * TILE, d1 = d0 + TILE < total ? d0 + TILE : total;
    const int last_full_nz = (p.nnz & ~3) - 4;
    const int last_full_ro = ((p.rows + 1) & ~3) - 4;
    int x0 = single ? 0 : hint_c.x, rs0 = single ? 0 : hint_r.x, x1 = single ? p.rows : hint_c.z, rs1 = single ? p.nnz : hint_r.y;
    const int hint_y0 = single ? 0 : hint_c.y, hint_y1 = single ? p.nnz : hint_c.w;
    int y0 = d0 - x0, y1 = d1 - x1;
    bool snap0 = y0 - rs0 <= HEAD_MAX, snap1 = y1 - rs1 <= HEAD_MAX;
    Coord c0, c1;
    c0.x = x0; c0.y = snap0 ? rs0 : y0;
    c1.x = x1; c1.y = snap1 ? rs1 : y1;
    // closed tile of short rows (block-uniform; consume_tile_rows): no row-start bits, raw product positions, one LDS phase
    // (unsigned: with garbage hints the product may wrap -- harmless, `lean` is only used once the hints have passed `good`,
    //  and then a tile has at most TILE rows)
    bool lean = snap0 && snap1 && (unsigned) (c1.y - c0.y) <= (unsigned) lean_avg * (unsigned) (c1.x - c0.x);
    // the clearing of the row-start bits is fenced from the staging's ORs by a barrier only in a tile that will use them -- a lean
    // tile neither sets nor reads them, and spares its waves the meeting (3.5-3.7 -> 3.4-3.6 us per small call).  As early as `lean`
    // is known (whatever the hints are worth: hints that fail the checks below send the block through the search, which clears
    // and fences for itself).  In the small shapes only: in the large one the circuit-shaped matrix, a mix of lean and other tiles,
    // was 1 % slower in every variant tried and dense5 / the grids as much faster (same-box A/B, profiles/r04_small_call_lean_barrier.txt)
    if constexpr (layout_hints<V, IPT>()) { if (!lean && !late_barrier) __syncthreads(); }
    // hints: anything may be in there.  Only values that keep every speculative access inside the arrays and the LDS tile
    // are tried at all (block-uniform)
    // 0 <= x0 <= x1 <= rows, 0 <= y0 <= y1 <= nnz, 0 <= rs0 <= y0, rs0 <= rs1 <= y1, as unsigned comparisons without branches (each
    // chain's last link bounds the earlier ones from below by zero); the last: the snapped boundaries in order
    bool good = ((unsigned) x0 <= (unsigned) x1) & ((unsigned) x1 <= (unsigned) p.rows) & ((unsigned) y0 <= (unsigned) y1) &
                ((unsigned) y1 <= (unsigned) p.nnz) & ((unsigned) rs0 <= (unsigned) y0) & ((unsigned) rs0 <= (unsigned) rs1) &
                ((unsigned) rs1 <= (unsigned) y1) & (c1.y >= c0.y);
    // the hints of the next call on this temp storage (and what mspmv_debug_read_tiles returns): stored when they were not there
    auto store_hints = [&](bool found_here) {
        if (tid == 0) {
            if (found_here || single || hint_y0 != y0) { Coord h; h.x = x0; h.y = y0; coords[tile] = h; rstart[tile] = rs0; }
            if (tile == num_tiles - 1 && (found_here || single || hint_y1 != y1)) { Coord h; h.x = x1; h.y = y1; coords[num_tiles] = h; rstart[num_tiles] = rs1; }
        }
    };
    if (MSPMV_LIKELY((layout_hints<V, IPT>()), good)) {
        // the four row offsets that decide whether (x0, rs0) and (x1, rs1) are the points of diagonals d0 and d1: requested
        // BEFORE the tile's streams, so they are back first
        // (by four lanes of EVERY wave: each wave then decides for itself -- all from the same four words, so all alike --
        //  and the verdict needs neither an LDS word nor a barrier of its own)
        int vre = 0;
        const int lane = tid & (WAVE - 1);
        if (lane < 4) {
            int idx = (lane < 2 ? x0 : x1) - 1 + (lane & 1);        // x0 - 1, x0, x1 - 1, x1
            idx = idx < 0 ? 0 : idx >= p.rows ? p.rows - 1 : idx;   // (rows >= 3 on this path)
            vre = p.row_end[idx];
        }
        TileRegs<V, BLOCK, IPT> regs;
        issue_nonzero_loads<V, BLOCK, IPT, NT>(p, c0, c1, regs);
        MSPMV_SNAP_TR(2);
        if (MSPMV_UNLIKELY((layout_hints<V, IPT>()), late_barrier)) { commit_x_to_lds<V, BLOCK>(p, xr, s_dyn); __syncthreads(); }
        // The verdict on the hints: looked at once the tile's x gathers have been requested -- the four words came back before the
        // streams did, and the scalar arithmetic on them runs in the shadow of the gathers (after the staging barrier it was ~100
        // instructions of a lone wave's critical path; before the staging it would hold the wave's share of the streams back).
        // (v_readlane: the four words and everything derived from them are scalars -- a shuffle goes through LDS and leaves vectors)
        bool verdict = true;
        auto check_hints = [&]() {
            const int before0 = __builtin_amdgcn_readlane(vre, 0), at0 = __builtin_amdgcn_readlane(vre, 1), before1 = __builtin_amdgcn_readlane(vre, 2), at1 = __builtin_amdgcn_readlane(vre, 3);
            // (x, rs) is the point of diagonal d  <=>  rs == row_offsets[x] <= y = d - x  and  (x == rows ? d == total : y <= row_offsets[x + 1])
            const bool ok0 = (x0 > 0 ? before0 == rs0 : rs0 == 0) & (x0 < p.rows ? y0 <= at0 : d0 == total);
            const bool ok1 = (x1 > 0 ? before1 == rs1 : rs1 == 0) & (x1 < p.rows ? y1 <= at1 : d1 == total);
            verdict = single | (ok0 & ok1);
        };
        stage_tile<V, BLOCK, IPT, NT, true, false, decltype(check_hints)>(p, c0, c1, tile, num_tiles, regs, s_end_raw, s_prod_raw, last_full_nz, last_full_ro, s_flag, s_x, -1, lean, check_hints);
        MSPMV_SNAP_TR(3);
        good = verdict;
    }
    if (MSPMV_UNLIKELY((layout_hints<V, IPT>()), !good)) {
        // no usable hints (the first call on this temp storage, or another matrix since): find the two boundaries, stage (again)
        // (after a failed check every wave is past the staging barrier, which touched nothing but LDS; the bits are cleared whatever
        //  was done to them, and the barrier below comes before the new staging)
        if (tid < SLOTS / 32 + 1) s_flag[tid] = 0u;
        if (late_barrier) commit_x_to_lds<V, BLOCK>(p, xr, s_dyn);      // (again, or for the first time: same values)
        __syncthreads();
        const int wave = tid / WAVE;
        if (wave < 2) {
            int rs = 0;
            const Coord c = wave_merge_path_guess(wave == 0 ? d0 : d1, p.row_end, p.rows, p.nnz, rs);
            if ((tid & (WAVE - 1)) == 0) { s_bnd[3 * wave] = c.x; s_bnd[3 * wave + 1] = rs; }
        }
        __syncthreads();
        x0 = s_bnd[0]; rs0 = s_bnd[1]; x1 = s_bnd[3]; rs1 = s_bnd[4];
        y0 = d0 - x0; y1 = d1 - x1;
        snap0 = y0 - rs0 <= HEAD_MAX; snap1 = y1 - rs1 <= HEAD_MAX;
        c0.x = x0; c0.y = snap0 ? rs0 : y0;
        c1.x = x1; c1.y = snap1 ? rs1 : y1;
        lean = snap0 && snap1 && (unsigned) (c1.y - c0.y) <= (unsigned) lean_avg * (unsigned) (c1.x - c0.x);
        TileRegs<V, BLOCK, IPT> regs;
        issue_nonzero_loads<V, BLOCK, IPT, NT>(p, c0, c1, regs);
        stage_tile<V, BLOCK, IPT, NT, true>(p, c0, c1, tile, num_tiles, regs, s_end_raw, s_prod_raw, last_full_nz, last_full_ro, s_flag, s_x, -1, lean);
    }
    store_hints(!good);
    // the tiles that hold published pieces of this tile's first row (only when the row ends here and began > HEAD_MAX
    // nonzeros before the tile): from the tile of its first nonzero (path item x0 + rs0) -- or the one after, if that one
    // handed its short piece on by the rule above -- up to this tile
    int first_piece = tile;
    if (!snap0) {                                                   // (also for a tile without a row end: a group leader needs it)
        const unsigned item = (unsigned) x0 + (unsigned) rs0;         // (<= rows + nnz < 2^31)
        const int ft = (int) (item / (unsigned) TILE);
        const unsigned tail_ft = (unsigned) (ft + 1) * (unsigned) TILE - item;
        first_piece = ft + (tail_ft <= (unsigned) HEAD_MAX ? 1 : 0);
    }
    const int pshift = c0.y - (c0.y & ~3);
    const int eshift = (c0.x + 1) - ((c0.x + 1) & ~3);
#ifdef MSPMV_DEV
#define MSPMV_SNAP_END() do { MSPMV_SNAP_TR(4); if (snap_tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MSPMV_SNAP_TR(5); } } while (0)    // (5: y stores acknowledged)
    unsigned long long *const lean_tr = snap_tr;
#else
#define MSPMV_SNAP_END() do { } while (0)
    unsigned long long *const lean_tr = nullptr;
#endif
    if (MSPMV_LIKELY((layout_hints<V, IPT>()), lean)) {
        consume_tile_rows<V, BLOCK, IPT, AXPBY>(p, c0, c1.x - c0.x, s_end_raw + eshift, s_prod_raw, pshift, carries + tile, lean_tr);
        MSPMV_SNAP_END();
        return;                                                     // (the program ends HERE, not after a jump over the general reduction)
    }
    consume_tile_flags<V, BLOCK, IPT, AXPBY, 4>(p, c0, c1.x - c0.x, c1.y - c0.y, s_end_raw + eshift, s_prod_raw, s_flag,
                                                s_wave_key, s_wave_val, carries + tile, pshift, nullptr, &lbe, tile, !snap1, first_piece, -1, rs0);
    MSPMV_SNAP_END();
#undef MSPMV_SNAP_END
#undef MSPMV_SNAP_TR
}

// mspmv_probe_read_stream: a bare 16-byte-per-lane read stream over a buffer, one 256 x 11-chunk piece per block like a tile's
// nonzero stream, ordinary or non-temporal loads -- the rate a measured kernel's algorithmic bytes are to be read against when
// its arrays live in the Infinity Cache (bench.py: records whose matrix fits 256 MB), where the HBM peak is not the bound.
__device__ int g_probe_sink;
template <bool NT>
__global__ __launch_bounds__(256) void probe_read_kernel(const int4v *__restrict__ p, unsigned long long n16)
{
    constexpr int PER = 11;
    const unsigned long long base = (unsigned long long) blockIdx.x * (256ull * PER) + threadIdx.x;
    int4v acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const unsigned long long i = base + 256ull * k;
        if (i < n16) { const int4v v = NT ? __builtin_nontemporal_load(p + i) : p[i]; acc ^= v; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x5a5a5a5a) g_probe_sink = 1;        // (keeps the loads; practically never true)
}

// ONE runtime call per launch.  `kernel<<<...>>>(...)` / hipLaunchKernelGGL is three (push the launch configuration, pop it in the stub,
// hipLaunchKernel) and leaves its status to a fourth (hipGetLastError); every entry into the HIP runtime costs the calling thread
// 50-100 ns, and the reference's timing loop is bound by the enqueueing thread below a few hundred tiles (tools/ab_driver: host
// enqueue time per call == loop time per call there).  The arguments are converted to the kernel's exact parameter types.
template <typename... P, typename... A, size_t... I>
static inline hipError_t launch_exact_impl(void (*kernel)(P...), dim3 grid, dim3 block, size_t dyn_lds, hipStream_t stream, std::index_sequence<I...>, A &&...a)
{
    std::tuple<P...> vals{static_cast<P>(a)...};
    void *ptrs[] = {static_cast<void *>(&std::get<I>(vals))...};
    return hipLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, dyn_lds, stream);
}
template <typename... P, typename... A>
static inline hipError_t launch_exact(void (*kernel)(P...), dim3 grid, dim3 block, size_t dyn_lds, hipStream_t stream, A &&...a)
{
    static_assert(sizeof...(P) == sizeof...(A), "one argument per kernel parameter");
    return launch_exact_impl(kernel, grid, block, dyn_lds, stream, std::index_sequence_for<P...>{}, static_cast<A &&>(a)...);
}

// host side of the compact variant (defined and instantiated in mspmv_compact.hip, the translation unit compiled for it)
template <typename V>
hipError_t launch_snap_compact(bool axpby, unsigned grid, size_t dyn_lds, hipStream_t stream, Coord *coords, int *rstart, int num_tiles,
                         const Params<V> &p, Carry<V> *carries, const LookBack &lb, int lean_avg, int tile_map);

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

#include "mspmv_tdm.hpp"
