// tools/binned_spmv.hip -- prototype of a STATELESS column-binned CsrMV for matrices whose columns are spread over an x
// larger than one XCD's L2 (BASELINE config 2), measured before anything of it goes into the library (development aid).
//
// The product's column-band passes read the whole CSR stream once per band (3-4 times) and still miss L2 with 12-35 % of
// the gathers.  Here the stream is re-written ONCE per call, binned by column band inside every merge-path tile:
//   A  bin    one block per merge-path tile: (col, val) and the tile's row ends staged in LDS, every nonzero gets its local
//             row (12 bits) and its band (col / band_cols, 8 bands), a STABLE partition by band inside LDS, the tile's
//             region of the temp buffer written in one coalesced piece: 8 segments of (value, local row | column in band)
//   B  bands  blocks of XCD k (block index % 8) take segment k of every tile: all gathers of that XCD fall into ONE eighth
//             of x, which its L2 keeps; one wave per (tile, band): products, a segmented sum by local row over the wave,
//             partial y of band k
//   C  sum    y[r] = beta * y[r] + alpha * (partial 0 + ... + partial 7) in that order; open rows through per-tile carries
// Everything is deterministic (no floating-point atomics in A-C; the prototype's carry add uses one because a row may get several).
// Build: make -C tools binned_spmv ; run on the GPU box: tools/binned_spmv [rows cols nnz_per_row]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <type_traits>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int BLOCK = 256, IPT = 11, ITEMS = BLOCK * IPT;       // merge-path items per tile (the library's large shape)
constexpr int NB = 8;                                           // column bands = XCDs
constexpr int PAD = 16;                                         // segments start on multiples of 16 entries
constexpr int REGION = ITEMS + NB * PAD;                        // entries per tile region (2944)
constexpr unsigned SENT_KEY = 0xFFFu;
constexpr int COL_BITS = 20;

struct Coord { int x, y; };                                     // (row, nonzero)

__device__ __forceinline__ unsigned mix(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned) (z ^ (z >> 31));
}
template <typename V>
__global__ void k_gen(int* __restrict__ off, int* __restrict__ col, V* __restrict__ val, int rows, int cols, int npr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    off[r] = r * npr;
    if (r == rows) return;
    int c[64];
    for (int j = 0; j < npr; ++j) c[j] = (int) (((unsigned long long) mix((unsigned long long) r * npr + j) * (unsigned) cols) >> 32);
    for (int i = 1; i < npr; ++i) { int v = c[i], j = i - 1; while (j >= 0 && c[j] > v) { c[j + 1] = c[j]; --j; } c[j + 1] = v; }
    for (int j = 0; j < npr; ++j) { col[(size_t) r * npr + j] = c[j]; val[(size_t) r * npr + j] = (V) (1.0 + (double) (mix(((unsigned long long) r * npr + j) ^ 0x5555ull) & 1023) * (1.0 / 4096.0)); }
}
template <typename V> __global__ void k_fill(V* x, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = (V) (1.0 + (double) (mix(i) & 255) * (1.0 / 256.0)); }

// reference: one thread per row, double accumulation
template <typename V>
__global__ void k_ref(const int* __restrict__ off, const int* __restrict__ col, const V* __restrict__ val, const V* __restrict__ x, double* __restrict__ y, double* __restrict__ yabs, int rows)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0, a = 0;
    for (int i = off[r]; i < off[r + 1]; ++i) { const double p = (double) val[i] * (double) x[col[i]]; s += p; a += fabs(p); }
    y[r] = s; yabs[r] = a;
}
template <typename V>
__global__ void k_cmp(const V* __restrict__ y, const double* __restrict__ g, const double* __restrict__ gabs, int rows, double eps, unsigned long long* worst_bits)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const double d = fabs((double) y[r] - g[r]);
    const double ratio = gabs[r] > 0 ? d / (eps * gabs[r]) : (d == 0 ? 0 : 1e30);
    atomicMax(worst_bits, (unsigned long long) __double_as_longlong(ratio));
}

// merge-path coordinates of every tile boundary (one thread per boundary)
__global__ void k_coords(const int* __restrict__ off, int rows, int nnz, int tiles, Coord* __restrict__ coords)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > tiles) return;
    const long long dl = (long long) t * ITEMS;
    const int d = (int) (dl < (long long) rows + nnz ? dl : (long long) rows + nnz);
    int lo = d - nnz > 0 ? d - nnz : 0, hi = d < rows ? d : rows;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (off[mid + 1] <= d - mid - 1) lo = mid + 1; else hi = mid; }
    coords[t].x = lo; coords[t].y = d - lo;
}

// ------------------------------------------------------------------------------------------------ phase A: bin
// 16-bit fields: per-band counts / prefixes of one block (<= 2816 per band)
struct Cnt8 { unsigned long long a, b; };        // bands 0-3 in a, 4-7 in b
__device__ __forceinline__ unsigned field(const Cnt8 c, int band) { const unsigned long long w = band < 4 ? c.a : c.b; return (unsigned) (w >> (16 * (band & 3))) & 0xFFFFu; }
__device__ __forceinline__ Cnt8 add(Cnt8 p, Cnt8 q) { p.a += q.a; p.b += q.b; return p; }
__device__ __forceinline__ Cnt8 shfl_up(Cnt8 c, int d) { Cnt8 r; r.a = __shfl_up(c.a, d); r.b = __shfl_up(c.b, d); return r; }

template <typename V>
__global__ __launch_bounds__(BLOCK) void k_bin(const int* __restrict__ off, const int* __restrict__ col, const V* __restrict__ val,
                                               const Coord* __restrict__ coords, int band_cols, unsigned band_magic,
                                               V* __restrict__ bval, unsigned* __restrict__ bpk, int* __restrict__ hdr, int4v* __restrict__ meta, int tiles)
{
    __shared__ V s_val[REGION];
    __shared__ unsigned s_col[REGION];
    __shared__ int s_mark[ITEMS];
    __shared__ int s_wmax[4];
    __shared__ unsigned long long s_wcnt[4][2];
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Coord c0 = coords[tile], c1 = coords[tile + 1];
    const int n = c1.y - c0.y, nclosed = c1.x - c0.x;
    // stage the nonzeros (16-byte loads of chunks aligned in array space) and clear the marks
#pragma unroll
    for (int j = 0; j < IPT; ++j) s_mark[j * BLOCK + tid] = 0;
    {
        const int a0 = c0.y & ~3, sh = c0.y - a0;
#pragma unroll
        for (int j = 0; j < (ITEMS / 4 + 1 + BLOCK - 1) / BLOCK; ++j) {
            const int ch = j * BLOCK + tid;
            if (ch * 4 < n + sh) {
                const int4v cc = __builtin_nontemporal_load(reinterpret_cast<const int4v*>(col + a0) + ch);
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int e = ch * 4 + i - sh; if (e >= 0 && e < n) s_col[e] = (unsigned) cc[i]; }
            }
        }
        constexpr int EPC = 16 / (int) sizeof(V);
        const int b0 = c0.y & ~(EPC - 1), shv = c0.y - b0;
#pragma unroll
        for (int j = 0; j < (ITEMS / EPC + 1 + BLOCK - 1) / BLOCK; ++j) {
            const int ch = j * BLOCK + tid;
            if (ch * EPC < n + shv) {
                const uint4v raw = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(val + b0) + ch);
                V tmp[EPC]; __builtin_memcpy(tmp, &raw, 16);
#pragma unroll
                for (int i = 0; i < EPC; ++i) { const int e = ch * EPC + i - shv; if (e >= 0 && e < n) s_val[e] = tmp[i]; }
            }
        }
    }
    __syncthreads();
    // every closed row's end marks the position from which on the local row is >= its successor (the largest wins: empty rows)
    for (int i = tid; i < nclosed; i += BLOCK) {
        const int pos = off[c0.x + i + 1] - c0.y;
        if (pos < n) atomicMax(&s_mark[pos], i + 1);
    }
    __syncthreads();
    // blocked: thread t owns entries t*IPT .. t*IPT+IPT-1
    unsigned c[IPT]; V v[IPT]; int rl[IPT]; int bd[IPT];
    int m = 0;
    unsigned long long cnt8 = 0;                  // 8-bit counts per band of this thread (<= 11)
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int e = tid * IPT + j;
        const int mk = s_mark[e];
        m = mk > m ? mk : m; rl[j] = m;
        c[j] = e < n ? s_col[e] : 0u; v[j] = e < n ? s_val[e] : (V) 0;
        unsigned q = __umulhi(c[j], band_magic);
        unsigned r = c[j] - q * (unsigned) band_cols;
        if (r >= (unsigned) band_cols) { ++q; r -= band_cols; }
        bd[j] = e < n ? (int) q : -1;
        c[j] = r;
        if (e < n) cnt8 += 1ull << (8 * q);
    }
    // block-wide exclusive max of the marks (local rows) and exclusive sums of the band counts
    Cnt8 cnt;
    cnt.a = (cnt8 & 0xFF) | ((cnt8 >> 8 & 0xFF) << 16) | ((cnt8 >> 16 & 0xFF) << 32) | ((cnt8 >> 24 & 0xFF) << 48);
    cnt.b = (cnt8 >> 32 & 0xFF) | ((cnt8 >> 40 & 0xFF) << 16) | ((cnt8 >> 48 & 0xFF) << 32) | ((cnt8 >> 56 & 0xFF) << 48);
    Cnt8 inc = cnt; int mx = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Cnt8 o = shfl_up(inc, d); const int om = __shfl_up(mx, d);
        if (lane >= d) { inc = add(inc, o); mx = om > mx ? om : mx; }
    }
    if (lane == 63) { s_wmax[wave] = mx; s_wcnt[wave][0] = inc.a; s_wcnt[wave][1] = inc.b; }
    int mprev = __shfl_up(mx, 1); if (lane == 0) mprev = 0;
    __syncthreads();
    Cnt8 wbase = {0, 0}, total = {0, 0};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const Cnt8 t = {s_wcnt[w][0], s_wcnt[w][1]};
        if (w < wave) { wbase = add(wbase, t); mprev = s_wmax[w] > mprev ? s_wmax[w] : mprev; }
        total = add(total, t);
    }
    // padded segment starts
    int seg[NB + 1]; seg[0] = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) seg[b + 1] = seg[b] + (((int) field(total, b) + PAD - 1) & ~(PAD - 1));
    Cnt8 base;                                    // where this thread's first entry of every band goes
    base.a = wbase.a + inc.a - cnt.a + ((unsigned long long) seg[0] | (unsigned long long) seg[1] << 16 | (unsigned long long) seg[2] << 32 | (unsigned long long) seg[3] << 48);
    base.b = wbase.b + inc.b - cnt.b + ((unsigned long long) seg[4] | (unsigned long long) seg[5] << 16 | (unsigned long long) seg[6] << 32 | (unsigned long long) seg[7] << 48);
    __syncthreads();                              // every thread has its staged entries in registers: LDS becomes the output region
    if (tid < NB * PAD) {                         // sentinel entries behind every segment
        const int b = tid / PAD, i = tid % PAD;
        const int at = seg[b] + (int) field(total, b) + i;
        if (at < seg[b + 1]) { s_col[at] = 0xFFFFFFFFu; s_val[at] = (V) 0; }
    }
    unsigned long long run8 = 0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        if (bd[j] >= 0) {
            const int at = (int) field(base, bd[j]) + (int) ((run8 >> (8 * bd[j])) & 0xFF);
            run8 += 1ull << (8 * bd[j]);
            const int rloc = rl[j] > mprev ? rl[j] : mprev;
            s_col[at] = ((unsigned) rloc << COL_BITS) | c[j];
            s_val[at] = v[j];
        }
    }
    if (tid < NB) {
        hdr[tile * 16 + tid] = seg[tid]; hdr[tile * 16 + 8 + tid] = (int) field(total, tid);
        int4v m; m[0] = seg[tid]; m[1] = (int) field(total, tid); m[2] = c0.x; m[3] = nclosed;
        meta[(size_t) tid * tiles + tile] = m;           // what the band phase needs to know about (band, tile), one record
    }
    __syncthreads();
    const int used = seg[NB];
    const size_t rb = (size_t) tile * REGION;
    for (int i = tid * 4; i < used; i += BLOCK * 4) {
        const uint4v pk = *reinterpret_cast<const uint4v*>(&s_col[i]);
        __builtin_nontemporal_store(pk, reinterpret_cast<uint4v*>(bpk + rb + i));
        if constexpr (sizeof(V) == 4) {
            const uint4v vv = *reinterpret_cast<const uint4v*>(&s_val[i]);
            __builtin_nontemporal_store(vv, reinterpret_cast<uint4v*>(bval + rb + i));
        } else {
            const uint4v v0 = *reinterpret_cast<const uint4v*>(&s_val[i]), v1 = *reinterpret_cast<const uint4v*>(&s_val[i + 2]);
            __builtin_nontemporal_store(v0, reinterpret_cast<uint4v*>(bval + rb + i));
            __builtin_nontemporal_store(v1, reinterpret_cast<uint4v*>(bval + rb + i + 2));
        }
    }
}

// ------------------------------------------------------------------------------------------------ phase B: bands
// one wave per (tile, band of its XCD)
// ABL (timing ablations, results wrong): 1 no zero fill / wait, 2 no segmented scan, 3 no gathers, 4 neither scan nor gathers nor zero fill
template <typename V, int WAVES, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64) void k_bands(const V* __restrict__ bval, const unsigned* __restrict__ bpk, const int* __restrict__ hdr,
                                                      const Coord* __restrict__ coords, const V* __restrict__ x, int band_cols, int tiles,
                                                      V* __restrict__ part, size_t part_stride, V* __restrict__ carry)
{
    const int k = blockIdx.x & 7, lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x >> 3) * WAVES + (threadIdx.x >> 6), waves_total = (gridDim.x >> 3) * WAVES;
    const V* __restrict__ xb = x + (size_t) k * band_cols;
    V* __restrict__ pk_out = part + (size_t) k * part_stride;
    for (int tile = wave_global; tile < tiles; tile += waves_total) {
        const int s = __builtin_amdgcn_readfirstlane(hdr[tile * 16 + k]), n = __builtin_amdgcn_readfirstlane(hdr[tile * 16 + 8 + k]);
        const Coord c0 = coords[tile], c1 = coords[tile + 1];
        const int row0 = __builtin_amdgcn_readfirstlane(c0.x), nclosed = __builtin_amdgcn_readfirstlane(c1.x) - row0;
        if (ABL != 1 && ABL != 4) { for (int r = lane; r < nclosed; r += 64) pk_out[row0 + r] = (V) 0;
        if (lane == 0) carry[(size_t) k * tiles + tile] = (V) 0; }
        const size_t eb = (size_t) tile * REGION + s;
        int carry_key = -1; V carry_sum = (V) 0;
        // (the zero stores above and the sums below go to the same addresses from the same wave: waited for)
        if (ABL != 1 && ABL != 4) __builtin_amdgcn_s_waitcnt(0);
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int e = j0 + lane;
            const bool valid = e < n;
            const unsigned pk = valid ? __builtin_nontemporal_load(bpk + eb + e) : 0xFFFFFFFFu;
            const V v = valid ? __builtin_nontemporal_load(bval + eb + e) : (V) 0;
            const int key = (int) (pk >> COL_BITS);
            const V xv = (ABL == 3 || ABL == 4) ? (V) (pk & 7) : valid ? xb[pk & ((1u << COL_BITS) - 1)] : (V) 0;
            V p = v * xv;
            int prevkey = __shfl_up(key, 1); if (lane == 0) prevkey = carry_key;
            bool f = key != prevkey;
            if (lane == 0 && f && carry_key >= 0) {                      // the run carried over from the previous 64 ended there
                if (carry_key == nclosed) carry[(size_t) k * tiles + tile] = carry_sum; else pk_out[row0 + carry_key] = carry_sum;
            }
            if (ABL != 2 && ABL != 4)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const V op = __shfl_up(p, d); const int of = __shfl_up((int) f, d);
                if (lane >= d && !f) { p += op; f = of != 0; }
            }
            if (!f) p += carry_sum;                                       // still the carried run
            int nextkey = __shfl_down(key, 1);
            const bool is_end = lane < 63 && key != nextkey && key != (int) SENT_KEY;
            if (is_end) { if (key == nclosed) carry[(size_t) k * tiles + tile] = p; else pk_out[row0 + key] = p; }
            carry_key = __builtin_amdgcn_readlane(key, 63);
            if constexpr (sizeof(V) == 4) carry_sum = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 63));
            else {
                const unsigned long long bits = __builtin_bit_cast(unsigned long long, p);
                const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) bits, 63), hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (bits >> 32), 63);
                carry_sum = __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
            }
            if (carry_key == (int) SENT_KEY) carry_key = -1;
        }
        if (lane == 0 && carry_key >= 0) { if (carry_key == nclosed) carry[(size_t) k * tiles + tile] = carry_sum; else pk_out[row0 + carry_key] = carry_sum; }
    }
}


// ---- phase B, second form: E consecutive entries per lane (16-byte loads), the next unit's entries requested behind this unit's
// gathers, one segmented scan per 64 * E entries, the tile's partial rows collected in LDS and written in one coalesced piece
// (tiles with more closed rows than the LDS window keeps: zero fill, wait, scattered stores -- as the first form)
constexpr int ACC_ROWS = 384;
template <typename V, int E> struct Entries { unsigned pk[E]; V v[E]; };

template <typename V, int E>
__device__ __forceinline__ void load_entries(Entries<V, E>& en, const V* __restrict__ bval, const unsigned* __restrict__ bpk, size_t eb, int first, int n_pad, int lane)
{
    const int e0 = first + lane * E;
    if (e0 < n_pad) {            // (segments are padded to 16 entries with sentinels and E divides 16: a lane's E entries are all inside)
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            const uint4v q = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(bpk + eb + e0 + i));
            en.pk[i] = q[0]; en.pk[i + 1] = q[1]; en.pk[i + 2] = q[2]; en.pk[i + 3] = q[3];
        }
        constexpr int EPC = 16 / (int) sizeof(V);
#pragma unroll
        for (int i = 0; i < E; i += EPC) {
            const uint4v raw = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(bval + eb + e0 + i));
            V tmp[EPC]; __builtin_memcpy(tmp, &raw, 16);
#pragma unroll
            for (int j = 0; j < EPC; ++j) en.v[i + j] = tmp[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < E; ++i) { en.pk[i] = 0xFFFFFFFFu; en.v[i] = (V) 0; }
    }
}
template <typename V> __device__ __forceinline__ V readlane_v(V p, int l)
{
    if constexpr (sizeof(V) == 4) return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), l));
    else {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, p);
        const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) bits, l), hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (bits >> 32), l);
        return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
    }
}

template <typename V, int WAVES, int E>
__global__ __launch_bounds__(WAVES * 64) void k_bands2(const V* __restrict__ bval, const unsigned* __restrict__ bpk, const int4v* __restrict__ meta,
                                                       const V* __restrict__ x, int band_cols, int tiles,
                                                       V* __restrict__ part, size_t part_stride, V* __restrict__ carry)
{
    __shared__ V s_acc[WAVES][ACC_ROWS + 1];
    const int k = blockIdx.x & 7, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_global = (blockIdx.x >> 3) * WAVES + wave, waves_total = (gridDim.x >> 3) * WAVES;
    const V* __restrict__ xb = x + (size_t) k * band_cols;
    V* __restrict__ pout = part + (size_t) k * part_stride;
    V* __restrict__ cout = carry + (size_t) k * tiles;
    volatile V* acc = s_acc[wave];
    const int4v* __restrict__ mk = meta + (size_t) k * tiles;
    constexpr int CHUNK = 64 * E;
    for (int batch = wave_global; batch < tiles; batch += waves_total * 64) {
        // the records of this wave's next 64 units, one per lane
        const int mine = batch + lane * waves_total;
        int4v mm; mm[0] = 0; mm[1] = 0; mm[2] = 0; mm[3] = 0;
        if (mine < tiles) mm = mk[mine];
        const int units = (tiles - batch + waves_total - 1) / waves_total < 64 ? (tiles - batch + waves_total - 1) / waves_total : 64;
        Entries<V, E> nxt;
        {
            const int s0 = __builtin_amdgcn_readlane(mm[0], 0), n0 = __builtin_amdgcn_readlane(mm[1], 0);
            load_entries<V, E>(nxt, bval, bpk, (size_t) batch * REGION + s0, 0, (n0 + 15) & ~15, lane);
        }
        for (int u = 0; u < units; ++u) {
            const int tile = batch + u * waves_total;
            const int s = __builtin_amdgcn_readlane(mm[0], u), n = __builtin_amdgcn_readlane(mm[1], u);
            const int row0 = __builtin_amdgcn_readlane(mm[2], u), nclosed = __builtin_amdgcn_readlane(mm[3], u);
            const int n_pad = (n + 15) & ~15;
            const size_t eb = (size_t) tile * REGION + s;
            const bool in_lds = nclosed <= ACC_ROWS;
            if (in_lds) { for (int r = lane; r <= nclosed; r += 64) acc[r] = (V) 0; }
            else {
                for (int r = lane; r < nclosed; r += 64) pout[row0 + r] = (V) 0;
                if (lane == 0) cout[tile] = (V) 0;
                __builtin_amdgcn_s_waitcnt(0);
            }
            auto store = [&](int key, V val) {
                if (key == (int) SENT_KEY) return;
                if (in_lds) acc[key] = val;
                else if (key == nclosed) cout[tile] = val; else pout[row0 + key] = val;
            };
            int carry_key = -1; V carry_sum = (V) 0;
            Entries<V, E> cur = nxt;
            for (int first = 0; first < n_pad || first == 0; first += CHUNK) {
                if (first > 0) load_entries<V, E>(cur, bval, bpk, eb, first, n_pad, lane);
                // gathers of this chunk, then the request for the next unit's first chunk behind them
                V xv[E];
#pragma unroll
                for (int i = 0; i < E; ++i) xv[i] = cur.pk[i] != 0xFFFFFFFFu ? xb[cur.pk[i] & ((1u << COL_BITS) - 1)] : (V) 0;
                if (first == 0 && u + 1 < units) {
                    const int s1 = __builtin_amdgcn_readlane(mm[0], u + 1), n1 = __builtin_amdgcn_readlane(mm[1], u + 1);
                    load_entries<V, E>(nxt, bval, bpk, (size_t) (tile + waves_total) * REGION + s1, 0, (n1 + 15) & ~15, lane);
                }
                // inside the lane: the first run is kept for the carry, middle runs are complete, the last run goes into the scan
                int key[E]; V p[E];
#pragma unroll
                for (int i = 0; i < E; ++i) { key[i] = (int) (cur.pk[i] >> COL_BITS); p[i] = cur.v[i] * xv[i]; }
                V tail = p[0], head = (V) 0; bool boundary = false;
#pragma unroll
                for (int i = 1; i < E; ++i) {
                    if (key[i] != key[i - 1]) {
                        if (!boundary) { head = tail; boundary = true; } else store(key[i - 1], tail);
                        tail = p[i];
                    } else tail += p[i];
                }
                const int k0 = key[0], k3 = key[E - 1];
                int prev_k3 = __shfl_up(k3, 1); if (lane == 0) prev_k3 = carry_key;
                bool f = k3 != prev_k3;
                if (lane == 0 && carry_key >= 0 && k0 != carry_key) store(carry_key, carry_sum);      // the carried run ended with the previous chunk
                V incl = tail;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const V op = __shfl_up(incl, d); const int of = __shfl_up((int) f, d);
                    if (lane >= d && !f) { incl += op; f = of != 0; }
                }
                if (!f) incl += carry_sum;
                V prev_incl = __shfl_up(incl, 1); if (lane == 0) prev_incl = carry_sum;
                if (boundary) store(k0, (k0 == prev_k3 ? prev_incl : (V) 0) + head);
                const int next_k0 = __shfl_down(k0, 1);
                if (lane < 63 && k3 != next_k0) store(k3, incl);
                carry_key = __builtin_amdgcn_readlane(k3, 63);
                carry_sum = readlane_v(incl, 63);
                if (carry_key == (int) SENT_KEY) carry_key = -1;
            }
            if (lane == 0 && carry_key >= 0) store(carry_key, carry_sum);
            if (in_lds) {
                for (int r = lane; r < nclosed; r += 64) pout[row0 + r] = acc[r];
                if (lane == 0) cout[tile] = acc[nclosed];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ phase C: sum
template <typename V>
__global__ void k_sum(const V* __restrict__ part, size_t part_stride, V* __restrict__ y, int rows)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    V s = __builtin_nontemporal_load(part + r);
#pragma unroll
    for (int k = 1; k < NB; ++k) s += __builtin_nontemporal_load(part + (size_t) k * part_stride + r);
    y[r] = s;
}
template <typename V>
__global__ void k_carry(const V* __restrict__ carry, const Coord* __restrict__ coords, int tiles, int rows, V* __restrict__ y)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const int row = coords[t + 1].x;
    if (row >= rows) return;
    V s = carry[t];
#pragma unroll
    for (int k = 1; k < NB; ++k) s += carry[(size_t) k * tiles + t];
    if (s != (V) 0) atomicAdd(y + row, s);
}

__global__ void k_copy(const uint4v* __restrict__ a, uint4v* __restrict__ b, size_t n)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}

template <typename F>
static float time_ms(F launch, int reps = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

template <typename V>
static void run(int rows, int cols, int npr)
{
    const size_t nnz = (size_t) rows * npr;
    const int tiles = (int) ((rows + nnz + ITEMS - 1) / ITEMS);
    int band_cols = (((cols + NB - 1) / NB) + 31) & ~31;
    if (band_cols > (1 << COL_BITS)) { printf("x too wide for %d column bits\n", COL_BITS); return; }
    const unsigned magic = (unsigned) (0x100000000ull / (unsigned) band_cols);
    int *off, *col, *hdr; int4v* meta; V *val, *x, *y, *bval, *part, *carry; unsigned* bpk; Coord* coords; double *g, *gabs; unsigned long long* worst;
    const size_t part_stride = ((size_t) rows + 63) & ~(size_t) 63;
    CK(hipMalloc(&off, ((size_t) rows + 1) * 4)); CK(hipMalloc(&col, nnz * 4 + 64)); CK(hipMalloc(&val, nnz * sizeof(V) + 64));
    CK(hipMalloc(&x, (size_t) band_cols * NB * sizeof(V))); CK(hipMalloc(&y, (size_t) rows * sizeof(V)));
    CK(hipMalloc(&bval, (size_t) tiles * REGION * sizeof(V))); CK(hipMalloc(&bpk, (size_t) tiles * REGION * 4)); CK(hipMalloc(&hdr, (size_t) tiles * 64)); CK(hipMalloc(&meta, (size_t) tiles * NB * 16));
    CK(hipMalloc(&part, part_stride * NB * sizeof(V))); CK(hipMalloc(&carry, (size_t) tiles * NB * sizeof(V))); CK(hipMalloc(&coords, ((size_t) tiles + 1) * sizeof(Coord)));
    CK(hipMalloc(&g, (size_t) rows * 8)); CK(hipMalloc(&gabs, (size_t) rows * 8)); CK(hipMalloc(&worst, 8));
    hipLaunchKernelGGL(k_gen<V>, dim3((rows + 256) / 256), dim3(256), 0, 0, off, col, val, rows, cols, npr);
    hipLaunchKernelGGL(k_fill<V>, dim3((band_cols * NB + 255) / 256), dim3(256), 0, 0, x, band_cols * NB);
    hipLaunchKernelGGL(k_ref<V>, dim3((rows + 255) / 256), dim3(256), 0, 0, off, col, val, x, g, gabs, rows);
    CK(hipDeviceSynchronize());
    int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const double balg = (double) nnz * (sizeof(V) + 4) + ((double) rows + 1) * 4 + (double) rows * sizeof(V) + (double) cols * sizeof(V);
    printf("# binned_spmv fp%d: %d rows x %d cols, %d per row, %zu nnz, %d tiles, x = %.2f MB (band %.2f MB), B_alg = %.1f MB, temp = %.1f MB\n",
           (int) sizeof(V) * 8, rows, cols, npr, nnz, tiles, cols * sizeof(V) * 1e-6, band_cols * sizeof(V) * 1e-6, balg * 1e-6,
           ((double) tiles * REGION * (sizeof(V) + 4) + (double) part_stride * NB * sizeof(V)) * 1e-6);
    auto A0 = [&] { hipLaunchKernelGGL(k_coords, dim3((tiles + 256) / 256), dim3(256), 0, 0, off, rows, (int) nnz, tiles, coords); };
    auto A = [&] { hipLaunchKernelGGL(k_bin<V>, dim3(tiles), dim3(BLOCK), 0, 0, off, col, val, coords, band_cols, magic, bval, bpk, hdr, meta, tiles); };
    auto C = [&] {
        hipLaunchKernelGGL(k_sum<V>, dim3((rows + 255) / 256), dim3(256), 0, 0, part, part_stride, y, rows);
        hipLaunchKernelGGL(k_carry<V>, dim3((tiles + 255) / 256), dim3(256), 0, 0, carry, coords, tiles, rows, y);
    };
    A0(); A();
    const float tA0 = time_ms(A0), tA = time_ms(A), tC = time_ms(C);
    printf("coords %.4f ms | bin %.4f ms (%.2f TB/s read+write) | sum + carries %.4f ms\n", tA0, tA, (double) nnz * (sizeof(V) + 4) * 2 / tA * 1e-9, tC);
    {
        const int grid = 8 * cus;
        const float t1 = time_ms([&] { hipLaunchKernelGGL((k_bands<V, 4, 1>), dim3(grid), dim3(256), 0, 0, bval, bpk, hdr, coords, x, band_cols, tiles, part, part_stride, carry); });
        const float t2 = time_ms([&] { hipLaunchKernelGGL((k_bands<V, 4, 2>), dim3(grid), dim3(256), 0, 0, bval, bpk, hdr, coords, x, band_cols, tiles, part, part_stride, carry); });
        const float t3 = time_ms([&] { hipLaunchKernelGGL((k_bands<V, 4, 3>), dim3(grid), dim3(256), 0, 0, bval, bpk, hdr, coords, x, band_cols, tiles, part, part_stride, carry); });
        const float t4 = time_ms([&] { hipLaunchKernelGGL((k_bands<V, 4, 4>), dim3(grid), dim3(256), 0, 0, bval, bpk, hdr, coords, x, band_cols, tiles, part, part_stride, carry); });
        const size_t n16 = nnz * sizeof(V) / 16;
        const float tc = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(16 * cus), dim3(256), 0, 0, (const uint4v*) val, (uint4v*) bval, n16); });
        printf("bands ablations (4 waves, 8 blocks/CU): no zero fill %.4f | no scan %.4f | no gathers %.4f | none of them %.4f ms ; plain copy of %zu MB: %.4f ms\n", t1, t2, t3, t4, n16 * 16 / 1000000, tc);
    }
    for (int waves : {4, 8}) for (int per_cu : {2, 4, 8}) {
        if (waves * per_cu > 32) continue;
        const int grid = per_cu * cus;
        auto B = [&] {
            if (waves == 4) hipLaunchKernelGGL((k_bands<V, 4>), dim3(grid), dim3(256), 0, 0, bval, bpk, hdr, coords, x, band_cols, tiles, part, part_stride, carry);
            else hipLaunchKernelGGL((k_bands<V, 8>), dim3(grid), dim3(512), 0, 0, bval, bpk, hdr, coords, x, band_cols, tiles, part, part_stride, carry);
        };
        const float tB = time_ms(B);
        const float tAll = time_ms([&] { A0(); A(); B(); C(); });
        CK(hipMemset(worst, 0, 8));
        hipLaunchKernelGGL(k_cmp<V>, dim3((rows + 255) / 256), dim3(256), 0, 0, y, g, gabs, rows, sizeof(V) == 4 ? 5.96e-8 : 1.11e-16, worst);
        unsigned long long wb; CK(hipMemcpy(&wb, worst, 8, hipMemcpyDeviceToHost));
        double w; memcpy(&w, &wb, 8);
        printf("bands: %d waves/block, %d blocks/CU : %.4f ms | whole SpMV %.4f ms = %.3f of 8 TB/s | worst |y - gold| / (eps * sum|v x|) = %.2f\n",
               waves, per_cu, tB, tAll, balg / (tAll * 1e-3) / 8e12, w);
    }
    auto check = [&](const char* name, float tB, float tAll) {
        CK(hipMemset(worst, 0, 8));
        hipLaunchKernelGGL(k_cmp<V>, dim3((rows + 255) / 256), dim3(256), 0, 0, y, g, gabs, rows, sizeof(V) == 4 ? 5.96e-8 : 1.11e-16, worst);
        unsigned long long wb; CK(hipMemcpy(&wb, worst, 8, hipMemcpyDeviceToHost));
        double w; memcpy(&w, &wb, 8);
        printf("bands2 %s : %.4f ms | whole SpMV %.4f ms = %.3f of 8 TB/s | worst %.2f\n", name, tB, tAll, balg / (tAll * 1e-3) / 8e12, w);
    };
#define RUN2(W, EE, PC) { const int grid = PC * cus; auto B = [&] { hipLaunchKernelGGL((k_bands2<V, W, EE>), dim3(grid), dim3(W * 64), 0, 0, bval, bpk, meta, x, band_cols, tiles, part, part_stride, carry); }; \
        CK(hipMemset(part, 0xFF, part_stride * NB * sizeof(V))); const float tB = time_ms(B); const float tAll = time_ms([&] { A0(); A(); B(); C(); }); check(#W " waves, E = " #EE ", " #PC " blocks/CU", tB, tAll); }
    RUN2(4, 4, 4) RUN2(4, 4, 8) RUN2(4, 8, 4) RUN2(4, 8, 8) RUN2(8, 8, 2) RUN2(8, 8, 4) RUN2(2, 8, 8) RUN2(2, 8, 16)
    CK(hipFree(off)); CK(hipFree(col)); CK(hipFree(val)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(bval)); CK(hipFree(bpk)); CK(hipFree(hdr));
    CK(hipFree(part)); CK(hipFree(carry)); CK(hipFree(coords)); CK(hipFree(g)); CK(hipFree(gabs)); CK(hipFree(worst));
}

int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 3125000;
    const int cols = argc > 2 ? atoi(argv[2]) : rows;
    const int npr = argc > 3 ? atoi(argv[3]) : 32;
    if (npr > 64) { printf("nnz per row <= 64\n"); return 1; }
    run<float>(rows, cols, npr);
    run<double>(rows, cols, npr);
    return 0;
}
