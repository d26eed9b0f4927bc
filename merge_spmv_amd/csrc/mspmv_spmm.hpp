// mspmv_spmm.hpp -- Y = alpha * A * X + beta * Y for a CSR matrix and a row-major block of
// K right-hand sides ("SpMM reusing the same path walk").  No reference
// counterpart: the reference ships CsrMV only (cub/device/device_spmv.cuh:129-164).
//
// Same decomposition as the CsrMV kernels (mspmv_kernels.hpp): merge-path tiles from the same
// coordinate pass, one tile per block, 16-byte streaming of (col, val), row-start flags and a
// segmented scan inside the tile, one carry per tile, one-launch owner-computes fix-up.  What
// changes is the value type flowing through LDS, the scan and the carries: a Pack of K numbers --
// the K entries X[col, c0 .. c0+K) that one gather brings in.  One pass over A serves K vectors,
// and a gather moves K * sizeof(T) useful bytes instead of one element of a 128-byte line, which
// is what bounds CsrMV on matrices without column locality.
// Packs of 4 .. 64 bytes are compiled (fp32: K = 1, 2, 4, 8, 16; fp64: K = 1, 2, 4, 8): a wide pack
// consumes a whole 32- or 64-byte row of X per gather, at the price of small tiles (the products of a
// tile must fit LDS) and few resident waves; wider blocks are processed in groups of the widest pack.
#pragma once

#include "mspmv_kernels.hpp"

namespace mspmv {

template <typename T, int K>
struct Pack {
    T v[K];
    Pack() = default;
    __host__ __device__ __forceinline__ Pack(T s)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = s;
    }
    __host__ __device__ __forceinline__ Pack &operator+=(const Pack &o)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] += o.v[k];
        return *this;
    }
    friend __host__ __device__ __forceinline__ Pack operator+(Pack a, const Pack &b) { a += b; return a; }
    friend __host__ __device__ __forceinline__ Pack operator*(T s, Pack a)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) a.v[k] *= s;
        return a;
    }
};

// cross-lane moves of a pack: element by element (found by argument-dependent lookup from the
// scan templates of mspmv_kernels.hpp)
template <int CTRL, int ROW_MASK, typename T, int K>
__device__ __forceinline__ Pack<T, K> dpp_move(Pack<T, K> old, Pack<T, K> src)
{
    Pack<T, K> r;
#pragma unroll
    for (int k = 0; k < K; ++k) r.v[k] = dpp_move<CTRL, ROW_MASK>(old.v[k], src.v[k]);
    return r;
}
template <typename T, int K>
__device__ __forceinline__ Pack<T, K> __shfl_up(Pack<T, K> a, unsigned delta, int width)
{
    Pack<T, K> r;
#pragma unroll
    for (int k = 0; k < K; ++k) r.v[k] = ::__shfl_up(a.v[k], delta, width);
    return r;
}

template <typename T, int K>
struct CarryMM {
    int key;
    Pack<T, K> value;
};

template <typename T>
struct MMParams {
    const T *__restrict__ values;
    const int *__restrict__ row_end;      // d_row_offsets + 1
    const int *__restrict__ cols;
    const T *__restrict__ x;              // first column of this group: X + c0
    T *__restrict__ y;                    // Y + c0
    int rows, nnz;
    int ldx, ldy;                         // leading dimensions (elements) of the row-major X and Y
    int x_vec, y_vec;                     // packs of X / Y are aligned to their own size: one vector load / store each
    T alpha, beta;
};

template <typename T, int K>
__device__ __forceinline__ Pack<T, K> load_pack(const T *__restrict__ p, bool aligned)
{
    typedef Pack<T, K> P;
    P r;
    if (K > 1 && aligned) {
        struct alignas(sizeof(P)) A { P p; };
        r = reinterpret_cast<const A *>(p)->p;
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) r.v[k] = p[k];
    }
    return r;
}
template <typename T, int K>
__device__ __forceinline__ void store_pack(T *__restrict__ p, const Pack<T, K> &v, bool aligned)
{
    typedef Pack<T, K> P;
    if (K > 1 && aligned) {
        struct alignas(sizeof(P)) A { P p; };
        reinterpret_cast<A *>(p)->p = v;
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) p[k] = v.v[k];
    }
}

// The staged products / running sums live in LDS as 16-byte units.  A pack of sizeof(P) bytes is a
// fraction of a unit (4, 8 bytes), one unit, or several consecutive logical units (32, 64 bytes); a
// thread of the nonzero phase owns UPT = NPT * sizeof(P) / 16 consecutive logical units.  Logical
// unit u is stored at u ^ ((u >> 4) & (UPT - 1)) (UPT a power of two): the threads of a quarter
// wave then start in different banks, and a group of 16 logical units stays a group.
template <typename P, int NPT>
__device__ __forceinline__ int mm_unit(int u)
{
    constexpr int UPT = NPT * (int) sizeof(P) / 16;
    constexpr bool POW2 = UPT >= 2 && UPT <= 16 && (UPT & (UPT - 1)) == 0;
    return POW2 ? u ^ ((u >> 4) & (UPT - 1)) : u;
}
template <typename P, int NPT>
__device__ __forceinline__ P lds_load_pack(const int4v *s_units, int e)
{
    P r;
    if (sizeof(P) <= 16) {
        constexpr int PPU = sizeof(P) <= 16 ? 16 / (int) sizeof(P) : 1;
        const char *base = reinterpret_cast<const char *>(&s_units[mm_unit<P, NPT>(e / PPU)]) + (e % PPU) * sizeof(P);
        r = *reinterpret_cast<const P *>(base);
    } else {
        constexpr int UPP = sizeof(P) > 16 ? (int) sizeof(P) / 16 : 1;
#pragma unroll
        for (int w = 0; w < UPP; ++w) {
            const int4v v = s_units[mm_unit<P, NPT>(e * UPP + w)];
            __builtin_memcpy(reinterpret_cast<char *>(&r) + 16 * w, &v, 16);
        }
    }
    return r;
}
template <typename P, int NPT>
__device__ __forceinline__ void lds_store_pack(int4v *s_units, int e, const P &val)
{
    if (sizeof(P) <= 16) {
        constexpr int PPU = sizeof(P) <= 16 ? 16 / (int) sizeof(P) : 1;
        char *base = reinterpret_cast<char *>(&s_units[mm_unit<P, NPT>(e / PPU)]) + (e % PPU) * sizeof(P);
        *reinterpret_cast<P *>(base) = val;
    } else {
        constexpr int UPP = sizeof(P) > 16 ? (int) sizeof(P) / 16 : 1;
#pragma unroll
        for (int w = 0; w < UPP; ++w) {
            int4v v;
            __builtin_memcpy(&v, reinterpret_cast<const char *>(&val) + 16 * w, 16);
            s_units[mm_unit<P, NPT>(e * UPP + w)] = v;
        }
    }
}

// One tile per block.  BLOCK x IPT path items; the staging is the predicated ("careful") form of
// mspmv_kernels.hpp for every tile (nothing outside the tile or the arrays is used).
template <typename T, int K, int BLOCK, int IPT, bool AXPBY, bool NT>
__global__ __launch_bounds__(BLOCK) void spmm_tile_kernel(MMParams<T> p, const Coord *__restrict__ coords,
                                                          CarryMM<T, K> *__restrict__ carries, int num_tiles, int groups,
                                                          int xcd_chunk_log2)
{
    typedef Pack<T, K> P;
    static_assert(sizeof(P) == 4 || sizeof(P) == 8 || sizeof(P) % 16 == 0, "a pack is 4 or 8 bytes or whole 16-byte units");
    constexpr int NW = BLOCK / WAVE;
    constexpr int CPT = IPT / 4 + 1;
    constexpr int SLOTS = CPT * BLOCK * 4;
    constexpr int NPT = CPT * 4;
    constexpr int UPT = NPT * (int) sizeof(P) / 16;        // 16-byte units per thread of the nonzero phase
    constexpr int FLAG_WORDS = SLOTS / 32 + 1;
    static_assert(FLAG_WORDS <= BLOCK && NPT <= 16, "flag word handling");
    __shared__ __attribute__((aligned(16))) end16_t s_end_raw[SLOTS];
    __shared__ int4v s_units[SLOTS * sizeof(P) / 16];
    __shared__ unsigned s_flag[FLAG_WORDS];
    __shared__ int s_wave_flag[NW];
    __shared__ P s_wave_val[NW];

    const int tid = threadIdx.x;
    const int tile = xcd_chunked_tile((int) blockIdx.x, num_tiles, xcd_chunk_log2);
    if (tid < FLAG_WORDS) s_flag[tid] = 0u;
    const Coord c0 = coords[tile];
    const Coord c1 = coords[tile + 1];
    Params<T> q;                                   // the CSR view the shared loaders take
    q.values = p.values; q.row_end = p.row_end; q.cols = p.cols; q.x = p.x; q.y = p.y; q.rows = p.rows; q.nnz = p.nnz; q.x_lds = 0;
    q.alpha = p.alpha; q.beta = p.beta;
    TileRegs<T, BLOCK, IPT> regs;
    issue_nonzero_loads<T, BLOCK, IPT, NT>(q, c0, c1, regs);
    const int last_full_nz = (p.nnz & ~3) - 4;
    const int last_full_ro = ((p.rows + 1) & ~3) - 4;
    const int *__restrict__ row_offsets = p.row_end - 1;
    const int tile_rows = c1.x - c0.x;
    const int tile_nnz = c1.y - c0.y;
    const int a0 = c0.y & ~3;
    const int pshift = c0.y - a0;
    const int first = c0.x + 1;
    const int i0 = first & ~3;
    const int eshift = first - i0;
    __syncthreads();                               // flags cleared before anyone sets one

    // ---- staging, part 1 (once per tile): row ends and row-start flags
    {
        Vec4<int> ro[CPT];
        const int ro_chunks = (tile_rows + eshift + 3) / 4;
        const int ro_safe = i0 < last_full_ro ? i0 : last_full_ro;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (k == 0 || k * BLOCK < ro_chunks) {
                const int qd = tid + k * BLOCK;
                int i = i0 + 4 * qd;
                i = (qd < ro_chunks && i <= last_full_ro) ? i : ro_safe;
                ro[k] = ld_stream4<NT>(row_offsets + i);
            }
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int qd = tid + k * BLOCK;
            if (qd < ro_chunks) {
                const int i = i0 + 4 * qd;
                int v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * qd - eshift + j;
                    const bool in = r >= 0 && r < tile_rows && i <= last_full_ro;
                    v[j] = in ? ro[k].get(j) - c0.y : 0x3fffffff;
                    if ((unsigned) v[j] < (unsigned) tile_nnz) atomicOr(&s_flag[(pshift + v[j]) >> 5], 1u << ((pshift + v[j]) & 31));
                }
                st_lds4(&s_end_raw[4 * qd], v);
            }
        }
        if (tid == 0) atomicOr(&s_flag[0], 1u << pshift);
        const bool ro_tail = first + tile_rows > last_full_ro + 4;
        if (ro_tail) {                                 // block-uniform; <= 3 row ends of the ragged array tail
            __syncthreads();
            const int i = last_full_ro + 4 + tid;
            const int r = i - first;
            if (r >= 0 && r < tile_rows) {
                const int v = row_offsets[i] - c0.y;
                s_end_raw[r + eshift] = (end16_t) v;
                if ((unsigned) v < (unsigned) tile_nnz) atomicOr(&s_flag[(pshift + v) >> 5], 1u << ((pshift + v) & 31));
            }
        }
    }
    const end16_t *s_end = s_end_raw + eshift;
    const int base = tid * NPT;
    const bool nz_tail = c1.y > last_full_nz + 4;
    // (fp64 values read non-temporally arrive laid out line by line over the wave, mspmv_kernels.hpp: ld_stream4_linewise)
    if constexpr (vals_linewise<T, NT, false>()) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) regs.val[k] = linewise_own(regs.val[k]);
    }

    // ---- one pass of the LDS phases per group of K right-hand sides: the tile's (col, val) stay in
    //      registers, its row structure (row ends, flags) in LDS; only the gathered X packs change
    for (int g = 0; g < groups; ++g) {
        const T *__restrict__ xg = p.x + (size_t) g * K;
        T *__restrict__ yg = p.y + (size_t) g * K + (size_t) c0.x * (unsigned) p.ldy;
        if (g > 0) __syncthreads();                    // the previous group's row phase has read its sums
        // staging, part 2: gathers of X packs, products
        {
            P xv[CPT][4];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int e0 = a0 + 4 * (tid + k * BLOCK);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = (unsigned) (e0 + i - c0.y) < (unsigned) tile_nnz && e0 <= last_full_nz;
                    const int c = in ? regs.col[k].get(i) : 0;
                    xv[k][i] = load_pack<T, K>(xg + (size_t) (unsigned) c * (unsigned) p.ldx, p.x_vec != 0);
                }
            }
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int chunk = tid + k * BLOCK;
                const int e0 = a0 + 4 * chunk;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = (unsigned) (e0 + i - c0.y) < (unsigned) tile_nnz && e0 <= last_full_nz;
                    lds_store_pack<P, NPT>(s_units, 4 * chunk + i, in ? regs.val[k].get(i) * xv[k][i] : P((T) 0));
                }
            }
            if (nz_tail) {                             // block-uniform; <= 3 nonzeros of the ragged array tail
                __syncthreads();
                const int j = last_full_nz + 4 + tid;
                if (j < c1.y && j >= c0.y)
                    lds_store_pack<P, NPT>(s_units, j - a0, p.values[j] * load_pack<T, K>(xg + (size_t) (unsigned) p.cols[j] * (unsigned) p.ldx, p.x_vec != 0));
            }
            __syncthreads();
        }
        // nonzero phase: segmented running sums over this thread's NPT consecutive products
        P s[NPT];
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const int4v w = s_units[mm_unit<P, NPT>(tid * UPT + u)];
            __builtin_memcpy(reinterpret_cast<char *>(s) + 16 * u, &w, 16);
        }
        const unsigned lo = s_flag[base >> 5], hi = s_flag[(base >> 5) + 1];
        const unsigned m = (unsigned) ((((unsigned long long) hi << 32) | lo) >> (base & 31)) & ((1u << NPT) - 1u);
        P run((T) 0);
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            run = ((m >> k) & 1u) ? s[k] : run + s[k];
            s[k] = run;
        }
        const P carry_in = block_exclusive_segsum<P, BLOCK>(m != 0u, run, s_wave_flag, s_wave_val);
        const unsigned lead = (m & (0u - m)) - 1u;
#pragma unroll
        for (int k = 0; k < NPT; ++k) s[k] += ((lead >> k) & 1u) ? carry_in : P((T) 0);
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            int4v w;
            __builtin_memcpy(&w, reinterpret_cast<const char *>(s) + 16 * u, 16);
            s_units[mm_unit<P, NPT>(tid * UPT + u)] = w;
        }
        __syncthreads();
        // row phase: Y[row, group] = running sum at the row's last nonzero
        for (int r = tid; r < tile_rows; r += BLOCK) {
            const int e = s_end[r];
            const int e0 = r > 0 ? s_end[r - 1] : 0;
            P sum = e > e0 ? lds_load_pack<P, NPT>(s_units, pshift + e - 1) : P((T) 0);
            T *dst = yg + (size_t) r * (unsigned) p.ldy;
            if (AXPBY) {
                sum = p.alpha * sum;
                if (p.beta != (T) 0) sum += p.beta * load_pack<T, K>(dst, p.y_vec != 0);
            }
            store_pack<T, K>(dst, sum, p.y_vec != 0);
        }
        if (tid == BLOCK - 1) {
            const int e_last = tile_rows > 0 ? s_end[tile_rows - 1] : 0;
            CarryMM<T, K> c; c.key = c0.x + tile_rows;
            c.value = tile_nnz > e_last ? lds_load_pack<P, NPT>(s_units, pshift + tile_nnz - 1) : P((T) 0);
            carries[(size_t) g * num_tiles + tile] = c;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Slot form, for groups of SW = 8 or 16 right-hand sides (round 6).  The pack kernel above carries K-wide products through LDS
// and a block-wide scan: with 32- and 64-byte packs that is 20-45 KB of LDS traffic per kilobyte of matrix, tiles of 384-768
// path items and few resident waves (dense 32-column matrix, k = 16: 1.0 ms for a 0.13 ms stream).  Here a SLOT of
// LS = SW / VEC adjacent lanes owns one contiguous share of the tile's nonzeros and every lane of it VEC = 16 bytes' worth of
// adjacent right-hand sides: the slot reads (col, val) from the tile staged in LDS (one broadcast read per nonzero), its lanes
// fetch X[col, c0 .. c0+SW) as ONE coalesced 32/64-byte piece (16 bytes per lane) and accumulate in registers; rows that begin
// and end inside the share are written straight to Y (one piece), a share's first and last row pieces meet through an LDS table
// of NS = BLOCK / LS partial rows, added in slot order (deterministic).  The tile's open row leaves as the same CarryMM the pack
// kernel writes: one fix-up serves both.  Tiles are merge-path tiles (rows + nonzeros bounded), the nonzero shares inside a tile
// are equal whatever the row lengths.
template <typename T, int SW, int VEC, int BLOCK, int IPT, bool AXPBY, bool NT>
__global__ __launch_bounds__(BLOCK) void spmm_lane_kernel(MMParams<T> p, const Coord *__restrict__ coords,
                                                          CarryMM<T, SW> *__restrict__ carries, int num_tiles, int groups,
                                                          int xcd_chunk_log2)
{
    static_assert(SW == 8 || SW == 16, "group width");
    static_assert(VEC * sizeof(T) <= 16 && SW % VEC == 0, "right-hand sides per lane: at most 16 bytes");
    constexpr int LS = SW / VEC;                        // lanes per slot
    constexpr int ITEMS = BLOCK * IPT;
    constexpr int NS = BLOCK / LS;                      // slots per block
    constexpr int U = 4;                                // nonzeros of a slot in flight per step (two steps are)
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    // (both arrays are indexed in ARRAY space -- position sh + e for the tile's nonzero e, sh = the tile's first nonzero modulo 4 --
    //  so that a 16-byte chunk of the CSR arrays is one 16-byte LDS store)
    __shared__ __attribute__((aligned(16))) unsigned s_off[ITEMS + 8];      // byte offset of X's row: col * ldx * sizeof(T) (the host checked that X spans < 4 GB)
    __shared__ __attribute__((aligned(16))) T s_val[ITEMS + 8];
    __shared__ int s_end[ITEMS];                        // row ends, relative to the tile's first nonzero
    __shared__ int s_prow[NS];                          // every slot's last row piece: local row (-1: the slot has no nonzeros) ...
    __shared__ __attribute__((aligned(16))) T s_pacc[NS][SW];      // ... and its sums
    const int tid = threadIdx.x;
    const int tile = xcd_chunked_tile((int) blockIdx.x, num_tiles, xcd_chunk_log2);
    const Coord c0 = coords[tile];
    const Coord c1 = coords[tile + 1];
    const int tile_rows = c1.x - c0.x;
    const int tile_nnz = c1.y - c0.y;
    const unsigned ldx = (unsigned) p.ldx, ldy = (unsigned) p.ldy;
    const unsigned ldx_bytes = ldx * (unsigned) sizeof(T);
    const int sh = c0.y & 3;

    // ---- staging (once per tile): 16-byte loads of chunks aligned in array space, the ragged array tail element by element
    int empty_rows = 0;
    {
        const int a0 = c0.y & ~3;
        const int chunks = (tile_nnz + sh + 3) >> 2;
        const int full = p.nnz & ~3;
        for (int ch = tid; ch < chunks; ch += BLOCK) {
            const int g0 = a0 + 4 * ch;
            int cc[4]; T vv[4];
            if (g0 + 4 <= full) {
                const Vec4<int> c4 = ld_stream4<NT>(p.cols + g0);
                const Vec4<T> v4 = ld_stream4<NT>(p.values + g0);
#pragma unroll
                for (int i = 0; i < 4; ++i) { cc[i] = c4.get(i); vv[i] = v4.get(i); }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const bool in = g0 + i < p.nnz; cc[i] = in ? p.cols[g0 + i] : 0; vv[i] = in ? p.values[g0 + i] : (T) 0; }
            }
            int4v o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (int) ((unsigned) cc[i] * ldx_bytes);
            *reinterpret_cast<int4v *>(&s_off[4 * ch]) = o;
            st_lds4(&s_val[4 * ch], vv);
        }
        for (int r = tid; r < tile_rows; r += BLOCK) {
            const int e = p.row_end[c0.x + r] - c0.y;
            s_end[r] = e;
            const int prev = r > 0 ? p.row_end[c0.x + r - 1] - c0.y : 0;        // (the neighbour's load: the same cache line)
            empty_rows |= e == prev;
        }
    }
    const int has_empty = __syncthreads_or(empty_rows);              // (also the barrier behind the staging)

    const int slot = tid / LS, c = (tid % LS) * VEC;                 // this lane's first right-hand side within the group
    const int i_begin = (int) ((long long) tile_nnz * slot / NS);
    const int i_end = (int) ((long long) tile_nnz * (slot + 1) / NS);
    int r0;                                                          // the row of nonzero i_begin: the first row that ends beyond it
    {
        int lo = 0, hi = tile_rows;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_end[mid] > i_begin) hi = mid; else lo = mid + 1; }
        r0 = lo;
    }
    const bool xv = p.x_vec != 0, yv = p.y_vec != 0;                 // this lane's pieces of X / Y are aligned: one vector access each

    auto ldv = [&](const T *src, bool aligned) {
        vecT r;
        if (aligned) r = *reinterpret_cast<const vecT *>(src);
        else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) r[k] = src[k];
        }
        return r;
    };
    auto stv = [&](T *dst, vecT v, bool aligned) {
        if (aligned) *reinterpret_cast<vecT *>(dst) = v;
        else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) dst[k] = v[k];
        }
    };

    for (int g = 0; g < groups; ++g) {
        const char *__restrict__ xg = reinterpret_cast<const char *>(p.x + (size_t) g * SW + c);
        T *__restrict__ yg = p.y + (size_t) g * SW + c + (size_t) c0.x * ldy;
        auto put = [&](int row, vecT sum) {
            T *dst = yg + (size_t) row * ldy;
            if (AXPBY) { sum = p.alpha * sum; if (p.beta != (T) 0) sum += p.beta * ldv(dst, yv); }
            stv(dst, sum, yv);
        };
        int r = r0;
        int cur_end = r < tile_rows ? s_end[r] : 0x7fffffff;
        vecT acc = (T) 0, first_acc = (T) 0;
        int first_row = -1;
        // a step: U nonzeros of the share; their X pieces are requested one step ahead (two register sets, no copies)
        auto fetch = [&](int i, T (&vv)[U], vecT (&xx)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pos = i + u < i_end ? i + u : i_end - 1;           // (clamped: a step past the share's end re-reads its last nonzero)
                vv[u] = s_val[sh + pos];
                xx[u] = ldv(reinterpret_cast<const T *>(xg + s_off[sh + pos]), xv);
            }
        };
        auto consume = [&](int i, const T (&vv)[U], const vecT (&xx)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pos = i + u;
                if (pos < i_end) {
                    if (pos >= cur_end) {                              // the row ended before this nonzero (then empty rows, if any)
                        if (first_row < 0) { first_row = r; first_acc = acc; } else put(r, acc);
                        acc = (T) 0;
                        do { ++r; cur_end = r < tile_rows ? s_end[r] : 0x7fffffff; } while (pos >= cur_end);
                    }
                    acc += vv[u] * xx[u];
                }
            }
        };
        T va[U], vb[U]; vecT xa[U], xb[U];
        int i = i_begin;
        if (i < i_end) fetch(i, va, xa);
        while (i < i_end) {
            if (i + U < i_end) fetch(i + U, vb, xb);
            consume(i, va, xa);
            i += U;
            if (i >= i_end) break;
            if (i + U < i_end) fetch(i + U, va, xa);
            consume(i, vb, xb);
            i += U;
        }
        const bool mine = i_begin < i_end;
        if (c == 0) s_prow[slot] = mine ? r : -1;
        *reinterpret_cast<vecT *>(&s_pacc[slot][c]) = acc;
        __syncthreads();
        // the pieces of `row` that earlier slots hold (a run of slots right before this one), added in slot order
        auto before = [&](int row) {
            int j = slot - 1;
            while (j >= 0 && (s_prow[j] == row || s_prow[j] == -1)) --j;
            vecT s = (T) 0;
            for (++j; j < slot; ++j) if (s_prow[j] == row) s += *reinterpret_cast<const vecT *>(&s_pacc[j][c]);
            return s;
        };
        if (first_row >= 0) put(first_row, before(first_row) + first_acc);          // the share's first row ended inside it
        if (mine && cur_end == i_end && r < tile_rows)                                // its last row ends exactly where the share ends
            put(r, before(r) + acc);
        if (slot == NS - 1) {                                                          // the tile's open row: one carry
            vecT s = (T) 0;
            for (int j = 0; j < NS; ++j) if (s_prow[j] == tile_rows) s += *reinterpret_cast<const vecT *>(&s_pacc[j][c]);
            CarryMM<T, SW> *cr = carries + (size_t) g * num_tiles + tile;
            if (c == 0) cr->key = c0.x + tile_rows;
#pragma unroll
            for (int k = 0; k < VEC; ++k) cr->value.v[c + k] = s[k];
        }
        if (has_empty) {                                                               // rows without a nonzero in this tile
            for (int idx = tid; idx < tile_rows * SW; idx += BLOCK) {
                const int rr = idx / SW, cc = idx % SW;
                if (s_end[rr] == (rr > 0 ? s_end[rr - 1] : 0)) {
                    T *dst = p.y + (size_t) g * SW + cc + (size_t) (c0.x + rr) * ldy;
                    *dst = (AXPBY && p.beta != (T) 0) ? p.beta * *dst : (T) 0;
                }
            }
        }
        __syncthreads();                                                               // the table is read before the next group writes it
    }
}

// One-launch fix-up over pack-valued carries: fixup_onepass_kernel of mspmv_kernels.hpp with
// Y[key, :] += alpha * sum.
template <typename T, int K, int BLOCK, int IPT>
__global__ __launch_bounds__(BLOCK) void spmm_fixup_kernel(const CarryMM<T, K> *__restrict__ in, int n, T *__restrict__ y, int ldy,
                                                           int y_vec, int rows, T alpha)
{
    typedef Pack<T, K> P;
    constexpr int CHUNK = BLOCK * IPT;
    constexpr int NW = BLOCK / WAVE;
    __shared__ int s_wave_key[NW];
    __shared__ P s_wave_val[NW];
    __shared__ int s_need;
    const int tid = threadIdx.x;
    const int base = blockIdx.x * CHUNK;
    in += (size_t) blockIdx.y * n;                     // one list of carries and K columns of Y per group
    y += (size_t) blockIdx.y * K;
    const int key_before = base > 0 ? in[base - 1].key : -1;

    int keys[IPT]; P vals[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int i = base + tid * IPT + k;
        if (i < n) { keys[k] = in[i].key; vals[k] = in[i].value; }
        else { keys[k] = 0x7fffffff; vals[k] = P((T) 0); }
    }
    const int first_i = base + tid * IPT;
    int cur = tid == 0 ? key_before : (first_i - 1 < n ? in[first_i - 1].key : 0x7fffffff);
    if (tid == BLOCK - 1) {
        const int last_key = keys[IPT - 1];
        const int next_i = base + CHUNK;
        s_need = (next_i < n && last_key != key_before && last_key < rows && in[next_i].key == last_key) ? 1 : 0;
    }
    P total((T) 0);
    int first_key = -1; P first_total((T) 0); bool have_first = false;
    int ekey[IPT]; P esum[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        ekey[k] = -1; esum[k] = P((T) 0);
        if (keys[k] != cur) {
            if (!have_first) { have_first = true; first_key = cur; first_total = total; }
            else { ekey[k] = cur; esum[k] = total; }
            cur = keys[k]; total = vals[k];
        } else total += vals[k];
    }
    int prev_key, agg_key; P carry_in, agg_val;
    block_exclusive_rbk<P, BLOCK>(cur, total, s_wave_key, s_wave_val, prev_key, carry_in, agg_key, agg_val);
    int fkey = -1; P fsum((T) 0);
    if (have_first) { fkey = first_key; fsum = first_total; if (tid > 0 && prev_key == first_key) fsum += carry_in; }
    if (fkey == key_before) fkey = -1;
    int lkey = -1; P lsum((T) 0);
    if (tid == BLOCK - 1 && agg_key != key_before) { lkey = agg_key; lsum = agg_val; }
    if (s_need) {
        const int akey = in[base + CHUNK].key;
        P part((T) 0);
        constexpr int U = 8;
        for (int pos = base + CHUNK;; pos += U * BLOCK) {
            bool ended = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = pos + tid + u * BLOCK;
                if (i < n && in[i].key == akey) part += in[i].value; else ended = true;
            }
            if (__syncthreads_or(ended ? 1 : 0)) break;
        }
        const P wsum = wave_segmented_inclusive_sum<P>(0, part);
        __syncthreads();
        if ((tid & (WAVE - 1)) == WAVE - 1) s_wave_val[tid / WAVE] = wsum;
        __syncthreads();
        if (tid == BLOCK - 1) {
#pragma unroll
            for (int w = 0; w < NW; ++w) lsum += s_wave_val[w];
        }
    }
    auto apply = [&](int key, const P &sum) {
        if (key >= 0 && key < rows) {
            T *dst = y + (size_t) key * (unsigned) ldy;
            store_pack<T, K>(dst, load_pack<T, K>(dst, y_vec != 0) + alpha * sum, y_vec != 0);
        }
    };
#pragma unroll
    for (int k = 0; k < IPT; ++k) apply(ekey[k], esum[k]);
    apply(fkey, fsum);
    apply(lkey, lsum);
}

}  // namespace mspmv
