// mspmv_tdm.hpp -- clock-scheduled column bands: the ONE-pass form of the column-band organisation (round 6).
//
// The column-band passes (run_band_passes, mspmv_kernels.hpp) read the whole CSR stream once per band so that at any moment
// every XCD gathers from one slice of x, which its 4 MiB L2 keeps.  Here the stream is read ONCE.  A block
//   1. loads its merge-path tile's (column, value) chunks as every tile kernel does,
//   2. sorts the tile's nonzeros by column band THROUGH LDS (replicated counters filled by LDS atomics, one exclusive scan, a second
//      round of atomics that hands out positions; the sorted (slot | column-in-band) words pass through the product array and come
//      back into registers, striped over the block: no LDS beyond what the tile kernel has anyway),
//   3. gathers band by band -- and WHEN it may gather from a band is read off the chip-wide 100 MHz clock (s_memrealtime):
//      band (t / slot) % B is "on air".  Blocks never talk to each other: whatever tile a block holds and whenever it got
//      it, its gathers of band b happen while the other blocks of its XCD gather from band b (or the ones just after), so the
//      XCD's L2 holds a band or two of x and the gathers hit.  The x values land in the product array at the nonzeros'
//      own slots,
//   4. multiplies (every thread its own chunks again), stages the row ends, and hands over to the ordinary reduction
//      (consume_tile_flags): y is written once, with the caller's alpha and beta, one carry per tile, ordinary fix-up.
// The clock is a cache-affinity schedule, not a protocol: a block may gather any band at any time and the result is the
// same -- bit for bit the one-sweep tile_kernel_vec's, since products and reduction are the same (unlike the passes,
// which add band partials).  Nothing waits for another workgroup; a wave with nothing on air sleeps and looks again.
// Carrier: tile_kernel_vec<..., BAND, TDM> (mspmv_kernels.hpp) -- one tile per block, the verdict of the 64 sampled windows chooses
// between this staging and the ordinary one; the slot length comes from the dispatcher (mspmv_api.hip).
// Measured: C2 fp32 0.64 ms against the passes' 0.83, fp64 0.98-1.00 against 1.30 (profiles/r06_c2_f32, r06_c2_f64); the prototype
// (tools/tdm_spmv.hip), a persistent kernel of its own and the staging inside the one-launch kernel, all measured and dropped, are
// in docs/history/round6.md.
#pragma once
#include "mspmv_kernels.hpp"

namespace mspmv {

constexpr int TDM_COPIES = 8;              // counters per (band, wave): lanes 8 apart share one
constexpr int TDM_SLOT_SHIFT = 20;         // sorted word = slot in the tile's raw product array (12 bits) << 20 | column inside its band

// the tile's products, row ends and row-start bits in LDS, as stage_tile_careful leaves them -- x gathered band by band
template <typename V, int BLOCK, int IPT, bool NT>
__device__ __forceinline__ void stage_tile_tdm(const Params<V> &p, const Coord c0, const Coord c1, const TileRegs<V, BLOCK, IPT> &regs,
                                               end16_t *s_end_raw, V *s_prod_raw, int last_full_nz, int last_full_ro, unsigned *s_flag,
                                               int *s_start, int *s_wave_sum, const TdmArgs &ta, int tid)
{
    constexpr int CPT = IPT / 4 + 1;
    constexpr int NW = BLOCK / WAVE;
    constexpr int SLOTS = CPT * BLOCK * 4;
    static_assert(SLOTS <= (1 << (32 - TDM_SLOT_SHIFT)), "slot bits");
    static_assert(SLOTS * (int) sizeof(end16_t) >= TDM_MAX_BANDS * NW * TDM_COPIES * (int) sizeof(int), "the counters borrow the row-end array");
    static_assert(NW * TDM_COPIES * TDM_MAX_BANDS <= 4 * BLOCK && (NW * TDM_COPIES) % 4 == 0, "four counters per thread in the scan");
    const int lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int *__restrict__ row_offsets = p.row_end - 1;
    const int tile_rows = c1.x - c0.x;
    const int tile_nnz = c1.y - c0.y;
    const int a0 = c0.y & ~3;
    const int first = c0.x + 1;
    const int i0 = first & ~3;
    const int eshift = first - i0;
    int *const s_cnt = reinterpret_cast<int *>(s_end_raw);           // [band][wave][copy]; the row ends are staged after the gathers
    unsigned *const s_exch = reinterpret_cast<unsigned *>(s_prod_raw);
    const unsigned col_mask = (1u << ta.band_shift) - 1u;
    const int ncnt = ta.bands * (NW * TDM_COPIES);

    // ---- 1. how many nonzeros of the tile fall into every (band, wave, copy) bucket
    for (int j = tid; j < ncnt; j += BLOCK) s_cnt[j] = 0;
    __syncthreads();
    unsigned in_mask = 0u;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int e0 = a0 + 4 * (tid + k * BLOCK);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = (unsigned) (e0 + i - c0.y) < (unsigned) tile_nnz && e0 <= last_full_nz;
            if (in) {
                in_mask |= 1u << (4 * k + i);
                const unsigned b = (unsigned) regs.col[k].get(i) >> ta.band_shift;
                (void) __hip_atomic_fetch_add(&s_cnt[(b * NW + wave) * TDM_COPIES + (lane & (TDM_COPIES - 1))], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    // ---- 2. exclusive scan of the counters (four per thread), band starts
    {
        int c[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = 4 * tid + j < ncnt ? s_cnt[4 * tid + j] : 0; sum += c[j]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const int o = __shfl_up(incl, d, WAVE); if (lane >= d) incl += o; }
        if (lane == WAVE - 1) s_wave_sum[wave] = incl;
        __syncthreads();
        int base = incl - sum;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) if (w < wave) base += s_wave_sum[w];
        // (a band's first bucket is the first of a thread's four: NW * TDM_COPIES is a multiple of 4)
        if ((4 * tid) % (NW * TDM_COPIES) == 0 && 4 * tid < ncnt) s_start[4 * tid / (NW * TDM_COPIES)] = base;
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (4 * tid + j < ncnt) s_cnt[4 * tid + j] = base; base += c[j]; }
        // (the sorted nonzeros: the tile's, less the <= 3 of a ragged array tail, which step 7 multiplies by itself)
        if (tid == BLOCK - 1) s_start[ta.bands] = base;
    }
    __syncthreads();
    // ---- 3. the sorted words: through the product array, back into registers (entry k * BLOCK + tid)
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((in_mask >> (4 * k + i)) & 1u) {
                const unsigned c = (unsigned) regs.col[k].get(i);
                const unsigned b = c >> ta.band_shift;
                // (the counters hold the buckets' first positions now: a second round of atomics hands the positions out)
                const int dst = __hip_atomic_fetch_add(&s_cnt[(b * NW + wave) * TDM_COPIES + (lane & (TDM_COPIES - 1))], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                s_exch[dst] = ((unsigned) (4 * (tid + k * BLOCK) + i) << TDM_SLOT_SHIFT) | (c & col_mask);
            }
        }
    }
    __syncthreads();
    unsigned ent[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) ent[k] = s_exch[k * BLOCK + tid];
    // band of the first entry of each of the wave's IPT windows (window k = entries k * BLOCK + wave * 64 ... + 63): wave-uniform
    // (five bits each, in one 64-bit scalar: eleven separate ones were eleven more scalar registers than the kernel has)
    static_assert(IPT * 5 <= 64 && TDM_MAX_BANDS <= 32, "window bands packed five bits each");
    unsigned long long wqp = 0ull;
    int lookahead = ta.lookahead;
    {
        const int st = lane < ta.bands ? s_start[lane + 1] : 0x7fffffff;
#pragma unroll
        for (int k = 0; k < IPT; ++k) wqp |= (unsigned long long) (__popcll(__ballot(st <= k * BLOCK + wave * WAVE)) & 31) << (5 * k);
        // a tile whose nonzeros lie in one or two bands (a local stretch of a matrix the sampled windows called spread) has nothing to
        // schedule: waiting for its bands' slots would cost it up to a rotation, its gathers are neighbours anyway -- it takes them at once
        const int lo = lane < ta.bands ? s_start[lane] : 0x7fffffff;
        if (__popcll(__ballot(lane < ta.bands && st > lo)) <= 2) lookahead = ta.bands;
    }
    __syncthreads();                          // every sorted word is in registers: the product array may take the x values
    // ---- 4. band by band, as the clock says; every wave on its own
    {
        const int sorted = s_start[ta.bands];
        unsigned done = 0u;
#pragma unroll
        for (int k = 0; k < IPT; ++k) if (k * BLOCK + wave * WAVE >= sorted) done |= 1u << k;
        const unsigned all = (1u << IPT) - 1u;
        // (a safety valve, not a schedule: a wave that has found nothing on air 4096 times in a row -- ~0.6 ms, twenty rotations and
        //  more -- stops asking the clock and takes what is left; it never fires under a clock that runs, and no launch can hang on one
        //  that does not)
        int idle = 0;
        while (done != all) {
            // (24 bits of the clock: exact in a float; the wrap every 0.17 s costs one odd slot)
            const unsigned slot = (unsigned) ((float) ((unsigned) wall_clock64() & 0xFFFFFFu) * ta.inv_slot);
            // (every XCD an eighth of a rotation ahead of the previous one -- HW_REG_XCC_ID --, so that the eight do not fetch the same band
            //  at once, was tried: 0.651 -> 0.647 ms fp32, 0.990 -> 0.982 fp64, within noise; not kept)
            const int on_air = (int) (slot % (unsigned) ta.bands);
            bool any = false;
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                if ((done >> k) & 1u) continue;
                const int wq = (int) ((wqp >> (5 * k)) & 31ull);
                int d = wq - on_air; if (d < 0) d += ta.bands;
                // (a window that also reaches BACK from the band on air is the same thing with the clock shifted: measured, only its width matters)
                if (d > lookahead && idle <= 4096) continue;
                const int idx = k * BLOCK + tid;
                if (idx < sorted) {
                    int q = wq;
                    while (q + 1 < ta.bands && idx >= s_start[q + 1]) ++q;         // (a window seldom spans more than two bands)
                    const unsigned e = ent[k];
                    s_prod_raw[prod_slot<V, CPT>((int) (e >> TDM_SLOT_SHIFT))] = p.x[((unsigned) q << ta.band_shift) | (e & col_mask)];
                }
                done |= 1u << k; any = true;
            }
            if (!any) { ++idle; __builtin_amdgcn_s_sleep(4); } else idle = 0;
        }
    }
    // row ends of the tile: requested behind the gathers (12 registers the gather phase does without), staged below
    Vec4<int> ro[CPT];
    const int ro_chunks = (tile_rows + eshift + 3) / 4;
    const int ro_safe = i0 < last_full_ro ? i0 : last_full_ro;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int q = tid + k * BLOCK;
        int i = i0 + 4 * q;
        i = (q < ro_chunks && i <= last_full_ro) ? i : ro_safe;
        ro[k] = ld_stream4<NT>(row_offsets + i);
    }
    __syncthreads();
    // ---- 5. products: every thread its own chunks again (x value in place -> product in place)
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int chunk = tid + k * BLOCK;
        const Vec4<V> own_val = regs.val[k];                   // (plain loads: issue_nonzero_loads<..., LINEWISE_OK = false>)
        constexpr int EPU = 16 / (int) sizeof(V);
        V xv[4], prod[4];
#pragma unroll
        for (int u = 0; u < 4 / EPU; ++u) ld_unit(&s_prod_raw[prod_unit<V, CPT>(chunk * (4 / EPU) + u) * EPU], &xv[u * EPU]);
#pragma unroll
        for (int i = 0; i < 4; ++i) prod[i] = ((in_mask >> (4 * k + i)) & 1u) ? own_val.get(i) * xv[i] : (V) 0;
        st_prod_chunk<CPT>(s_prod_raw, chunk, prod, false);
    }
    // ---- 6. row ends and row-start bits (the counters' space is free again)
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
            if ((unsigned) v[j] < (unsigned) tile_nnz) {
                const int b = (c0.y - a0) + v[j];
                atomicOr(&s_flag[b >> 5], 1u << (b & 31));
            }
        }
        st_lds4(&s_end_raw[4 * q], v);
    }
    if (tid == 0) atomicOr(&s_flag[0], 1u << (c0.y - a0));
    // ---- 7. ragged array tails (only the tile that reaches the end of an array), as stage_tile_careful patches them
    const bool nz_tail = c1.y > last_full_nz + 4;
    const bool ro_tail = first + tile_rows > last_full_ro + 4;
    if (nz_tail || ro_tail) {
        __syncthreads();
        const int j = last_full_nz + 4 + tid;
        if (nz_tail && j < c1.y && j >= c0.y) {
            const int c = ld_stream<NT>(p.cols + j);
            s_prod_raw[prod_slot<V, CPT>(j - a0)] = ld_stream<NT>(p.values + j) * p.x[c];
        }
        const int i = last_full_ro + 4 + tid;
        const int r = i - first;
        if (ro_tail && r >= 0 && r < tile_rows) {
            const int v = ld_stream<NT>(row_offsets + i) - c0.y;
            s_end_raw[r + eshift] = (end16_t) v;
            if ((unsigned) v < (unsigned) tile_nnz) atomicOr(&s_flag[((c0.y - a0) + v) >> 5], 1u << (((c0.y - a0) + v) & 31));
        }
    }
    __syncthreads();
}

}  // namespace mspmv
