// tools/tdm_spmv.hip -- prototype of a STATELESS one-pass CsrMV whose x gathers are TIME-DIVISION MULTIPLEXED by column band
// (round 6; development aid, measured before anything of it goes into the library).
//
// The product's column-band passes read the CSR stream once per band so that, chip-wide, every XCD's L2 only ever sees one band
// of x at a time.  Here the stream is read ONCE: a block sorts the nonzeros of its tile by column band inside LDS and then
// gathers band by band -- but WHICH band it may gather from is decided by the chip-wide 100 MHz clock (s_memrealtime):
// band (t >> slot_shift) % B is "on air".  Blocks never talk to each other; whatever tile a block holds and whenever it got it, its
// gathers of band b happen while every other block of the XCD gathers from band b too, so the XCD's L2 holds one or two bands of
// x and the gathers hit.  Correctness never depends on the clock (a block may gather any band at any time): it is a
// cache-affinity schedule, not a protocol.
// The prototype handles matrices with a fixed number of nonzeros per row that divides the tile (C2: 32); tiles are row-aligned.
// Build: make -C tools tdm_spmv ; run on the GPU box: tools/tdm_spmv [rows cols nnz_per_row]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int int4v __attribute__((ext_vector_type(4)));
template <typename V> struct Vec4T;
template <> struct Vec4T<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4T<double> { typedef double type __attribute__((ext_vector_type(4))); };

constexpr int BLOCK = 256;
constexpr int MAXB = 32;
constexpr int POS_SHIFT = 20;            // entry = position in tile (12 bits) << 20 | column inside its band (<= 20 bits)

__device__ __forceinline__ unsigned mix(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned) (z ^ (z >> 31));
}
template <typename V>
__global__ void k_gen(int* __restrict__ col, V* __restrict__ val, int rows, int cols, int npr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int c[64];
    for (int j = 0; j < npr; ++j) c[j] = (int) (((unsigned long long) mix((unsigned long long) r * npr + j) * (unsigned) cols) >> 32);
    for (int i = 1; i < npr; ++i) { int v = c[i], j = i - 1; while (j >= 0 && c[j] > v) { c[j + 1] = c[j]; --j; } c[j + 1] = v; }
    for (int j = 0; j < npr; ++j) { col[(size_t) r * npr + j] = c[j]; val[(size_t) r * npr + j] = (V) (1.0 + (double) (mix(((unsigned long long) r * npr + j) ^ 0x5555ull) & 1023) * (1.0 / 4096.0)); }
}
template <typename V> __global__ void k_fill(V* x, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = (V) (1.0 + (double) (mix(i) & 255) * (1.0 / 256.0)); }
template <typename V>
__global__ void k_ref(const int* __restrict__ col, const V* __restrict__ val, const V* __restrict__ x, double* __restrict__ y, double* __restrict__ yabs, int rows, int npr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0, a = 0;
    for (size_t i = (size_t) r * npr; i < (size_t) (r + 1) * npr; ++i) { const double p = (double) val[i] * (double) x[col[i]]; s += p; a += fabs(p); }
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

struct Args {
    int rows, npr, tiles;
    int band_shift;          // band = col >> band_shift
    int bands;               // B
    float inv_slot;          // 1 / (ticks of the 100 MHz clock per slot)
    int lookahead;           // bands after the one on air that a block may also take (0, 1, ...)
    int mode;                // 0 clocked bands | 1 bands in order 0..B-1, no clock | 2 no sort: gathers in load order | 3 clocked, gathers replaced by a constant
    int prefetch;            // 1: after its segment of the band on air a wave touches its share of the NEXT band's lines
    int *claim;              // 8 counters, 64 ints apart
};

// MODE is a run-time field: the prototype trades a few branches for one kernel per (V, IPT)
template <typename V, int IPT, bool SORTED>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((sizeof(V) == 8 && IPT == 11) ? 4 : 7))) void k_tdm(const int* __restrict__ col, const V* __restrict__ val, const V* __restrict__ x, V* __restrict__ y, const Args a)
{
    typedef typename Vec4T<V>::type V4;
    constexpr int ITEMS = BLOCK * IPT, CH = ITEMS / 4, CPT = (CH + BLOCK - 1) / BLOCK;
    __shared__ unsigned s_ent[ITEMS];
    __shared__ __attribute__((aligned(32))) V s_xv[ITEMS];
    __shared__ int s_cnt[MAXB], s_start[MAXB + 1];
    __shared__ int s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seq = blockIdx.x & 7;
    const int rows_per_tile = ITEMS / a.npr, chunks_per_row = a.npr / 4;
    const unsigned col_mask = (1u << a.band_shift) - 1u;
    for (;;) {
        if (tid == 0) s_tile = atomicAdd(a.claim + seq * 64, 1) * 8 + seq;
        if (tid < MAXB) s_cnt[tid] = 0;
        __syncthreads();
        const int tile = s_tile;
        if (tile >= a.tiles) break;
        int4v c[CPT]; V4 v[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int ch = tid + k * BLOCK;
            const size_t g = (size_t) tile * CH + (ch < CH ? ch : 0);
            c[k] = __builtin_nontemporal_load(reinterpret_cast<const int4v*>(col) + g);
            v[k] = __builtin_nontemporal_load(reinterpret_cast<const V4*>(val) + g);
        }
        if constexpr (!SORTED) {
          if (a.mode == 4) {
            // NO SORT: every thread keeps its own nonzeros and gathers each one when its band is on air (a gather instruction per item
            // and slot, a twelfth of its lanes active) -- does the texture path charge by instruction or by active lane?
            V xv[CPT][4];
            unsigned pend = 0u;
#pragma unroll
            for (int k = 0; k < CPT; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) { xv[k][i] = (V) 0; if (tid + k * BLOCK < CH) pend |= 1u << (4 * k + i); }
            while (__ballot(pend != 0u) != 0ull) {
                const unsigned slot = (unsigned) ((float) ((unsigned) wall_clock64() & 0xFFFFFFu) * a.inv_slot);
                const int on_air = (int) (slot % (unsigned) a.bands);
                bool any = false;
#pragma unroll
                for (int k = 0; k < CPT; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if ((pend >> (4 * k + i)) & 1u) {
                            int d = (int) ((unsigned) c[k][i] >> a.band_shift) - on_air; if (d < 0) d += a.bands;
                            if (d <= a.lookahead) { xv[k][i] = x[c[k][i]]; pend &= ~(1u << (4 * k + i)); any = true; }
                        }
                    }
                if (__ballot(any) == 0ull) __builtin_amdgcn_s_sleep(4);
            }
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int ch = tid + k * BLOCK;
                V p = (V) 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) p += v[k][i] * xv[k][i];
                for (int d = 1; d < chunks_per_row; d <<= 1) p += __shfl_xor(p, d);
                if (ch < CH && (ch % chunks_per_row) == 0) y[(size_t) tile * rows_per_tile + ch / chunks_per_row] = p;
            }
            __syncthreads();
            continue;
          }
            // baseline in the same harness: one pass, every gather where it falls
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int ch = tid + k * BLOCK;
                V p = (V) 0;
                if (ch < CH) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) p += v[k][i] * x[c[k][i]];
                }
                for (int d = 1; d < chunks_per_row; d <<= 1) p += __shfl_xor(p, d);
                if (ch < CH && (ch % chunks_per_row) == 0) y[(size_t) tile * rows_per_tile + ch / chunks_per_row] = p;
            }
            __syncthreads();
        } else {
        // ---- rank inside the band (LDS atomics), then the sorted entries
        unsigned short rk[CPT][4];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int ch = tid + k * BLOCK;
            if (ch < CH) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rk[k][i] = (unsigned short) atomicAdd(&s_cnt[(unsigned) c[k][i] >> a.band_shift], 1);
            }
        }
        __syncthreads();
        if (tid < 64) {
            // exclusive prefix of the B counts over one wave
            int n = tid < a.bands ? s_cnt[tid] : 0, incl = n;
#pragma unroll
            for (int d = 1; d < MAXB; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            if (tid <= a.bands) s_start[tid] = incl - n;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int ch = tid + k * BLOCK;
            if (ch < CH) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned cc = (unsigned) c[k][i];
                    s_ent[s_start[cc >> a.band_shift] + rk[k][i]] = ((unsigned) (4 * ch + i) << POS_SHIFT) | (cc & col_mask);
                }
            }
        }
        __syncthreads();
        // ---- band by band; every wave on its own (no block barrier inside)
        unsigned done = 0u;
        const unsigned full = a.bands >= 32 ? 0xFFFFFFFFu : (1u << a.bands) - 1u;
        int next_in_order = 0;
        while (done != full) {
            int q = -1;
            if (a.mode == 1) q = next_in_order++;
            else {
                // (24 bits of the clock: exact in a float; the wrap every 0.17 s costs one odd slot)
                const unsigned slot = (unsigned) ((float) ((unsigned) wall_clock64() & 0xFFFFFFu) * a.inv_slot);
                const int on_air = (int) (slot % (unsigned) a.bands);
                for (int d = 0; d <= a.lookahead; ++d) {
                    int bb = on_air + d; if (bb >= a.bands) bb -= a.bands;
                    if (!((done >> bb) & 1u)) { q = bb; break; }
                }
                if (q < 0) { __builtin_amdgcn_s_sleep(4); continue; }
            }
            const int s0 = s_start[q], s1 = s_start[q + 1];
            const V* xb = x + ((size_t) q << a.band_shift);
            for (int i = s0 + wave * 64 + lane; i < s1; i += BLOCK) {
                const unsigned e = s_ent[i];
                s_xv[e >> POS_SHIFT] = a.mode == 3 ? (V) 1 : xb[e & col_mask];
            }
            if (a.prefetch && a.mode == 0) {
                // a share of the next band's lines, so that its first gathers find them in L2: the block's waves cover
                // (blockIdx / 8 * 4 + wave) * 64 + lane of the band's 128-byte lines, modulo the band
                int nb = q + 1; if (nb >= a.bands) nb = 0;
                const unsigned lines = (unsigned) ((sizeof(V) << a.band_shift) >> 7);
                const unsigned line = (unsigned) (((blockIdx.x >> 3) * 4 + wave) * 64 + lane) % lines;
                const V t = *reinterpret_cast<const volatile V*>(reinterpret_cast<const char*>(x + ((size_t) nb << a.band_shift)) + ((size_t) line << 7));
                asm volatile("" :: "v"(t));
            }
            done |= 1u << q;
        }
        __syncthreads();
        // ---- products and rows (npr nonzeros = chunks_per_row consecutive chunks = consecutive lanes)
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int ch = tid + k * BLOCK;
            V p = (V) 0;
            if (ch < CH) {
                const V4 xv = *reinterpret_cast<const V4*>(s_xv + 4 * ch);
#pragma unroll
                for (int i = 0; i < 4; ++i) p += v[k][i] * xv[i];
            }
            for (int d = 1; d < chunks_per_row; d <<= 1) p += __shfl_xor(p, d);
            if (ch < CH && (ch % chunks_per_row) == 0) y[(size_t) tile * rows_per_tile + ch / chunks_per_row] = p;
        }
        __syncthreads();
        }
    }
}

template <typename F>
static float time_ms(F launch, int reps = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

template <typename V, int IPT>
static void run(int rows_in, int cols, int npr, const char* only)
{
    constexpr int ITEMS = BLOCK * IPT;
    if (ITEMS % npr || npr % 4 || (npr / 4 & (npr / 4 - 1))) { printf("fp%d IPT %d: %d per row does not fit the prototype's tiles\n", (int) sizeof(V) * 8, IPT, npr); return; }
    const int rpt = ITEMS / npr;
    const int rows = rows_in / rpt * rpt;                       // whole tiles only
    const size_t nnz = (size_t) rows * npr;
    const int tiles = rows / rpt;
    int *col, *claim; V *val, *x, *y; double *g, *gabs; unsigned long long* worst;
    const size_t xcap = (size_t) 1 << 26;
    CK(hipMalloc(&col, nnz * 4 + 64)); CK(hipMalloc(&val, nnz * sizeof(V) + 64));
    CK(hipMalloc(&x, (xcap > (size_t) cols ? xcap : (size_t) cols) * sizeof(V))); CK(hipMalloc(&y, (size_t) rows * sizeof(V)));
    CK(hipMalloc(&g, (size_t) rows * 8)); CK(hipMalloc(&gabs, (size_t) rows * 8)); CK(hipMalloc(&worst, 8)); CK(hipMalloc(&claim, 8 * 64 * 4));
    hipLaunchKernelGGL(k_gen<V>, dim3((rows + 255) / 256), dim3(256), 0, 0, col, val, rows, cols, npr);
    hipLaunchKernelGGL(k_fill<V>, dim3((cols + 255) / 256), dim3(256), 0, 0, x, cols);
    hipLaunchKernelGGL(k_ref<V>, dim3((rows + 255) / 256), dim3(256), 0, 0, col, val, x, g, gabs, rows, npr);
    CK(hipDeviceSynchronize());
    int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const double balg = (double) nnz * (sizeof(V) + 4) + ((double) rows + 1) * 4 + (double) rows * sizeof(V) + (double) cols * sizeof(V);
    printf("# tdm_spmv fp%d IPT %d: %d rows x %d cols, %d per row, %zu nnz, %d tiles of %d, x = %.2f MB, B_alg = %.1f MB\n",
           (int) sizeof(V) * 8, IPT, rows, cols, npr, nnz, tiles, ITEMS, cols * sizeof(V) * 1e-6, balg * 1e-6);
    auto one = [&](int mode, int band_shift, int slot_ticks, int lookahead, int per_cu, int prefetch) {
        Args a; a.rows = rows; a.npr = npr; a.tiles = tiles; a.band_shift = band_shift; a.bands = (cols + (1 << band_shift) - 1) >> band_shift;
        a.inv_slot = 1.0f / (float) slot_ticks; a.lookahead = lookahead; a.mode = mode; a.prefetch = prefetch; a.claim = claim;
        if (a.bands > MAXB || band_shift > POS_SHIFT) return;
        const int grid = per_cu * cus;
        CK(hipMemset(y, 0xFF, (size_t) rows * sizeof(V)));
        auto L = [&] {
            CK(hipMemsetAsync(claim, 0, 8 * 64 * 4));
            if (mode == 2 || mode == 4) hipLaunchKernelGGL((k_tdm<V, IPT, false>), dim3(grid), dim3(BLOCK), 0, 0, col, val, x, y, a);
            else hipLaunchKernelGGL((k_tdm<V, IPT, true>), dim3(grid), dim3(BLOCK), 0, 0, col, val, x, y, a);
        };
        const float t = time_ms(L);
        CK(hipMemset(worst, 0, 8));
        hipLaunchKernelGGL(k_cmp<V>, dim3((rows + 255) / 256), dim3(256), 0, 0, y, g, gabs, rows, sizeof(V) == 4 ? 5.96e-8 : 1.11e-16, worst);
        unsigned long long wb; CK(hipMemcpy(&wb, worst, 8, hipMemcpyDeviceToHost));
        double w; memcpy(&w, &wb, 8);
        printf("mode %d  bands %2d (%.2f MB)  slot %5.2f us  lookahead %d  prefetch %d  blocks/CU %d : %.4f ms = %.3f of 8 TB/s | worst %.2f\n",
               mode, a.bands, (double) (sizeof(V) << band_shift) * 1e-6, (double) slot_ticks * 0.01, lookahead, prefetch, per_cu, t, balg / (t * 1e-3) / 8e12, w);
        fflush(stdout);
    };
    const int max_per_cu = sizeof(V) == 4 ? (IPT == 11 ? 7 : 8) : (IPT == 11 ? 4 : 7);
    if (only && strstr(only, "nosort")) {
        for (int bs : {18, 19}) {
            const int bands = (cols + (1 << bs) - 1) >> bs;
            for (int pc : {8, 6}) {
                const double ideal_us = (double) pc * cus * ITEMS / bands / 262e3;
                for (double f : {0.85, 1.0, 1.16, 1.35, 1.6}) for (int la : {0, 1, 2}) one(4, bs, (int) (ideal_us * f * 100.0 + 0.5), la, pc, 0);
            }
        }
    }
    if (!only || strstr(only, "base")) one(2, 20, 256, 0, 8, 0);
    if (!only || strstr(only, "order")) for (int bs : {18, 19}) one(1, bs, 256, 0, max_per_cu, 0);
    // x bytes per band -> slot lengths around (resident nonzeros / bands) / 262 G gathers/s
    for (int bs : {18, 19, 20}) {
        const int bands = (cols + (1 << bs) - 1) >> bs;
        const double ideal_us = (double) max_per_cu * cus * ITEMS / bands / 262e3;
        for (double f : {0.7, 0.85, 1.0, 1.15, 1.3, 1.6, 2.0}) {
            const int ticks = (int) (ideal_us * f * 100.0 + 0.5);
            for (int la : {0, 1, 2}) if (!only || strstr(only, "tdm")) one(0, bs, ticks, la, max_per_cu, 0);
            if (!only || strstr(only, "nogather")) one(3, bs, ticks, 1, max_per_cu, 0);
        }
    }
    CK(hipFree(col)); CK(hipFree(val)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(g)); CK(hipFree(gabs)); CK(hipFree(worst)); CK(hipFree(claim));
}

int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 3125000;
    const int cols = argc > 2 ? atoi(argv[2]) : rows;
    const int npr = argc > 3 ? atoi(argv[3]) : 32;
    const char* only = argc > 4 ? argv[4] : nullptr;
    const char* prec = argc > 5 ? argv[5] : "both";
    if (npr > 64) { printf("nnz per row <= 64\n"); return 1; }
    if (strcmp(prec, "f64")) run<float, 11>(rows, cols, npr, only);
    if (strcmp(prec, "f32")) { run<double, 7>(rows, cols, npr, only); run<double, 11>(rows, cols, npr, only); }
    return 0;
}
