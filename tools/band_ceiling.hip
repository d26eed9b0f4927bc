// tools/band_ceiling.hip -- what ANY column-band scheme over the unchanged CSR arrays can reach on C2 (development aid,
// not part of the product).  The kernels below do the memory work of a banded CsrMV tile -- 16-byte streams of (col, val),
// the gather of x for the nonzeros of one column band -- and nothing else (no LDS staging, no row reduction, no y), so their
// times are ceilings for the product's tile_kernel_vec<.., BAND> under each organisation:
//   passes   chip-wide phases, one band per phase (what run_band_passes does), with / without a register prefetch of the
//            next tile and with / without fetching the values only for chunks that hold a nonzero of the band
//   xcd      every XCD (pair) owns one band for the whole call and streams the whole matrix itself: no phases
//   slice    the prepared plan's access pattern: ONE stream, gathers from an L2-sized table (upper bound for everything)
//   gather2  cols-only passes that store the gathered x per nonzero, then ONE val * xg pass
// Build: make -C tools band_ceiling ; run on the GPU box: tools/band_ceiling > gpurun_out/band_ceiling.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned) (z ^ (z >> 31));
}
// C2: `npr` uniform random sorted columns per row
__global__ void k_gen(int* __restrict__ col, float* __restrict__ val, int rows, int cols, int npr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int c[32];
    for (int j = 0; j < npr; ++j) c[j] = (int) (((unsigned long long) mix((unsigned long long) r * npr + j) * (unsigned) cols) >> 32);
    for (int i = 1; i < npr; ++i) { int v = c[i], j = i - 1; while (j >= 0 && c[j] > v) { c[j + 1] = c[j]; --j; } c[j + 1] = v; }
    for (int j = 0; j < npr; ++j) { col[(size_t) r * npr + j] = c[j]; val[(size_t) r * npr + j] = 1.0f + (float) (mix(((unsigned long long) r * npr + j) ^ 0x5555ull) & 1023) * (1.0f / 4096.0f); }
}
__global__ void k_fill(float* x, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = 1.0f + (float) (i & 7) * 0.125f; }

constexpr int BLOCK = 256, CPT = 3, TILE4 = BLOCK * CPT;      // 16-byte chunks per tile (3072 nonzeros: the product's 256x11 staging)

template <bool NT> __device__ __forceinline__ int4v ldc(const int4v* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ float4v ldv(const float4v* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// ---- chip-wide passes: for band b: every block walks tiles blockIdx.x, +gridDim.x, ...
template <bool PREFETCH, bool LAZY, bool NT>
__global__ __launch_bounds__(BLOCK) void k_passes(const int4v* __restrict__ col, const float4v* __restrict__ val, const float* __restrict__ x,
                                                  size_t n4, int bands, int band_cols, float* out)
{
    const size_t ntiles = (n4 + TILE4 - 1) / TILE4;
    float acc = 0.f;
    for (int b = 0; b < bands; ++b) {
        const unsigned lo = (unsigned) b * band_cols;
        int4v c[CPT], cn[CPT]; float4v v[CPT], vn[CPT];
        size_t t = blockIdx.x;
        if (PREFETCH && t < ntiles) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); if (!LAZY) vn[k] = ldv<NT>(val + i); }
        }
        for (; t < ntiles; t += gridDim.x) {
            if (PREFETCH) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) { c[k] = cn[k]; if (!LAZY) v[k] = vn[k]; }
                const size_t t2 = t + gridDim.x < ntiles ? t + gridDim.x : t;
#pragma unroll
                for (int k = 0; k < CPT; ++k) { size_t i = t2 * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); if (!LAZY) vn[k] = ldv<NT>(val + i); }
            } else {
#pragma unroll
                for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; c[k] = ldc<NT>(col + i); if (!LAZY) v[k] = ldv<NT>(val + i); }
            }
            float xv[CPT][4];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                bool any = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xv[k][j] = 0.f;
                    if ((unsigned) c[k][j] - lo < (unsigned) band_cols) { xv[k][j] = x[c[k][j]]; any = true; }
                }
                if (LAZY) {
                    size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1;
                    v[k] = float4v{0.f, 0.f, 0.f, 0.f};
                    if (any) v[k] = ldv<NT>(val + i);
                }
            }
#pragma unroll
            for (int k = 0; k < CPT; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += v[k][j] * xv[k][j];
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// ---- every XCD (group) owns one band: block b runs on XCD b & 7 (round-robin dispatch); band = xcd % bands; the 8 / bands
//      XCDs of a band share the tiles between them
template <bool PREFETCH, bool NT>
__global__ __launch_bounds__(BLOCK) void k_xcd(const int4v* __restrict__ col, const float4v* __restrict__ val, const float* __restrict__ x,
                                               size_t n4, int bands, int band_cols, float* out)
{
    const size_t ntiles = (n4 + TILE4 - 1) / TILE4;
    const int xcd = blockIdx.x & 7, band = xcd % bands, rep = xcd / bands, reps = 8 / bands;
    const size_t first = (size_t) (blockIdx.x >> 3) * reps + rep, stride = (size_t) (gridDim.x >> 3) * reps;
    const unsigned lo = (unsigned) band * band_cols;
    float acc = 0.f;
    int4v c[CPT], cn[CPT]; float4v v[CPT], vn[CPT];
    size_t t = first;
    if (PREFETCH && t < ntiles) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
    }
    for (; t < ntiles; t += stride) {
        if (PREFETCH) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { c[k] = cn[k]; v[k] = vn[k]; }
            const size_t t2 = t + stride < ntiles ? t + stride : t;
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = t2 * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
        } else {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; c[k] = ldc<NT>(col + i); v[k] = ldv<NT>(val + i); }
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((unsigned) c[k][j] - lo < (unsigned) band_cols) acc += v[k][j] * x[c[k][j]];
    }
    if (acc == 12345.678f) out[0] = acc;
}

// ---- one stream, every gather folded into a table of `mask + 1` entries (the prepared plan's pattern; the ceiling)
// PF 0: load tile, gather, multiply.  PF 1: the next tile's stream is requested BEFORE this tile's gathers (vmcnt counts in
// order: waiting for the gathers then also waits for the younger... no: OLDER stream -- the mis-ordered pipeline).
// PF 2: gathers of tile t issued first, THEN the stream of tile t + 1, then wait for the gathers only (vmcnt(#stream loads)):
// stream latency and gather latency overlap inside one wave.
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
template <int PF, bool NT>
__global__ __launch_bounds__(BLOCK) void k_slice(const int4v* __restrict__ col, const float4v* __restrict__ val, const float* __restrict__ x,
                                                 size_t n4, unsigned mask, float* out)
{
    const size_t ntiles = (n4 + TILE4 - 1) / TILE4;
    float acc = 0.f;
    int4v c[CPT], cn[CPT]; float4v v[CPT], vn[CPT];
    size_t t = blockIdx.x;
    if (PF && t < ntiles) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
    }
    for (; t < ntiles; t += gridDim.x) {
        const size_t t2 = t + gridDim.x < ntiles ? t + gridDim.x : t;
        if (PF == 1) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { c[k] = cn[k]; v[k] = vn[k]; }
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = t2 * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
        } else if (PF == 2) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { c[k] = cn[k]; v[k] = vn[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; c[k] = ldc<NT>(col + i); v[k] = ldv<NT>(val + i); }
        }
        float xv[CPT][4];
        sched_fence();
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[k][j] = x[(unsigned) c[k][j] & mask];
        sched_fence();
        if (PF == 2) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = t2 * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
            sched_fence();
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[k][j] * xv[k][j];
        sched_fence();
    }
    if (acc == 12345.678f) out[0] = acc;
}

// ---- every XCD group owns one band, correctly ordered pipeline (PF as in k_slice), dynamic tile claims per group so that
//      the groups' blocks stay together in the matrix (counters: one per band, zeroed by the host)
template <int PF, bool NT>
__global__ __launch_bounds__(BLOCK) void k_xcd2(const int4v* __restrict__ col, const float4v* __restrict__ val, const float* __restrict__ x,
                                                size_t n4, int bands, int band_cols, float* out)
{
    const size_t ntiles = (n4 + TILE4 - 1) / TILE4;
    const int xcd = blockIdx.x & 7, band = xcd % bands, rep = xcd / bands, reps = 8 / bands;
    const size_t first = (size_t) (blockIdx.x >> 3) * reps + rep, stride = (size_t) (gridDim.x >> 3) * reps;
    const unsigned lo = (unsigned) band * band_cols;
    float acc = 0.f;
    int4v c[CPT], cn[CPT]; float4v v[CPT], vn[CPT];
    size_t t = first;
    if (PF && t < ntiles) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
    }
    for (; t < ntiles; t += stride) {
        const size_t t2 = t + stride < ntiles ? t + stride : t;
        if (PF) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { c[k] = cn[k]; v[k] = vn[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; c[k] = ldc<NT>(col + i); v[k] = ldv<NT>(val + i); }
        }
        float xv[CPT][4];
        sched_fence();
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[k][j] = 0.f; if ((unsigned) c[k][j] - lo < (unsigned) band_cols) xv[k][j] = x[c[k][j]]; }
        sched_fence();
        if (PF) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = t2 * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
            sched_fence();
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[k][j] * xv[k][j];
        sched_fence();
    }
    if (acc == 12345.678f) out[0] = acc;
}

// ---- chip-wide passes with dynamic, ascending tile claims (as run_band_passes does it: 8 interleaved sequences, one
//      counter per XCD and pass), correctly ordered pipeline
template <int PF, bool NT>
__global__ __launch_bounds__(BLOCK) void k_passes_dyn(const int4v* __restrict__ col, const float4v* __restrict__ val, const float* __restrict__ x,
                                                      size_t n4, int bands, int band_cols, int* counters, float* out)
{
    const int ntiles = (int) ((n4 + TILE4 - 1) / TILE4);
    __shared__ int s_t[2];
    float acc = 0.f;
    const int seq = blockIdx.x & 7;
    for (int b = 0; b < bands; ++b) {
        const unsigned lo = (unsigned) b * band_cols;
        int* ctr = counters + (b * 8 + seq) * 64;
        int4v c[CPT], cn[CPT]; float4v v[CPT], vn[CPT];
        __syncthreads();
        if (threadIdx.x == 0) { s_t[0] = 8 * atomicAdd(ctr, 1) + seq; }
        __syncthreads();
        int t = s_t[0];
        int par = 0;
        if (PF && t < ntiles) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { size_t i = (size_t) t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
        }
        while (t < ntiles) {
            if (threadIdx.x == 0) s_t[par ^ 1] = 8 * atomicAdd(ctr, 1) + seq;      // the claim for the tile after this one
            if (PF) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) { c[k] = cn[k]; v[k] = vn[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < CPT; ++k) { size_t i = (size_t) t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; c[k] = ldc<NT>(col + i); v[k] = ldv<NT>(val + i); }
            }
            float xv[CPT][4];
            sched_fence();
#pragma unroll
            for (int k = 0; k < CPT; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[k][j] = 0.f; if ((unsigned) c[k][j] - lo < (unsigned) band_cols) xv[k][j] = x[c[k][j]]; }
            sched_fence();
            __syncthreads();
            const int t2 = s_t[par ^ 1];
            if (PF && t2 < ntiles) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) { size_t i = (size_t) t2 * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; cn[k] = ldc<NT>(col + i); vn[k] = ldv<NT>(val + i); }
            }
            sched_fence();
#pragma unroll
            for (int k = 0; k < CPT; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += v[k][j] * xv[k][j];
            sched_fence();
            t = t2; par ^= 1;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// ---- gather-only passes (cols in, gathered x out, each nonzero written once over all passes), then val * xg
template <bool NT>
__global__ __launch_bounds__(BLOCK) void k_gather_pass(const int4v* __restrict__ col, const float* __restrict__ x, float4v* __restrict__ xg,
                                                       size_t n4, int band, int band_cols)
{
    const size_t ntiles = (n4 + TILE4 - 1) / TILE4;
    const unsigned lo = (unsigned) band * band_cols;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int4v c[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) { size_t i = t * TILE4 + k * BLOCK + threadIdx.x; i = i < n4 ? i : n4 - 1; c[k] = ldc<NT>(col + i); }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const size_t i = t * TILE4 + k * BLOCK + threadIdx.x;
            float* dst = reinterpret_cast<float*>(xg + (i < n4 ? i : n4 - 1));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((unsigned) c[k][j] - lo < (unsigned) band_cols) dst[j] = x[c[k][j]];
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_dot(const float4v* __restrict__ val, const float4v* __restrict__ xg, size_t n4, float* out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t) BLOCK + threadIdx.x; i < n4; i += (size_t) gridDim.x * BLOCK) {
        const float4v a = __builtin_nontemporal_load(val + i), b = __builtin_nontemporal_load(xg + i);
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    if (acc == 12345.678f) out[0] = acc;
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

int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 3125000, npr = 32;
    const int cols = argc > 2 ? atoi(argv[2]) : rows;
    const size_t nnz = (size_t) rows * npr, n4 = nnz / 4;
    int* col; float *val, *x, *out, *xg;
    CK(hipMalloc(&col, nnz * 4)); CK(hipMalloc(&val, nnz * 4)); CK(hipMalloc(&xg, nnz * 4)); CK(hipMalloc(&x, (size_t) cols * 4 + 64)); CK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(k_gen, dim3((rows + 255) / 256), dim3(256), 0, 0, col, val, rows, cols, npr);
    hipLaunchKernelGGL(k_fill, dim3((cols + 255) / 256), dim3(256), 0, 0, x, cols);
    CK(hipDeviceSynchronize());
    int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("# band_ceiling: %d rows x %d cols, %d nnz/row, %zu nnz (fp32); x = %.2f MB; %d CUs; ms per whole SpMV-equivalent\n", rows, cols, npr, nnz, cols * 4e-6, cus);
    const int4v* c4 = (const int4v*) col; const float4v* v4 = (const float4v*) val;

    printf("## slice: ONE stream of (col, val) + every gather folded into an L2-sized table (the prepared plan's ceiling)\n");
    printf("##        plain | stream(t+1) requested before gathers(t) | gathers(t) first, then stream(t+1), wait for the gathers only\n");
    for (unsigned mask : {0x3ffffu, 0x7ffffu, 0xfffffu}) for (int per_cu : {2, 4, 8}) {      // (1.05, 2.10, 4.19 MB: the plan's slices are 3.1 MB at 4 bands, 1.6 MB at 8)
        const int grid = per_cu * cus;
        float a = time_ms([&] { hipLaunchKernelGGL((k_slice<0, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, mask, out); });
        float b = time_ms([&] { hipLaunchKernelGGL((k_slice<1, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, mask, out); });
        float c = time_ms([&] { hipLaunchKernelGGL((k_slice<2, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, mask, out); });
        printf("table %5.2f MB  blocks/CU %d : %.4f | %.4f | %.4f ms\n", (mask + 1) * 4e-6, per_cu, a, b, c);
    }
    int* counters; CK(hipMalloc(&counters, 8 * 8 * 64 * 4));
    printf("## passes_dyn: chip-wide phases, ascending dynamic tile claims (8 sequences), no barrier between phases: plain | pipelined\n");
    for (int bands : {2, 3, 4}) for (int per_cu : {2, 3, 4, 6, 8}) {
        const int grid = per_cu * cus, bc = (cols + bands - 1) / bands;
        float a = time_ms([&] { CK(hipMemsetAsync(counters, 0, 8 * 8 * 64 * 4, 0)); hipLaunchKernelGGL((k_passes_dyn<0, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, counters, out); });
        float b = time_ms([&] { CK(hipMemsetAsync(counters, 0, 8 * 8 * 64 * 4, 0)); hipLaunchKernelGGL((k_passes_dyn<2, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, counters, out); });
        float c = time_ms([&] { CK(hipMemsetAsync(counters, 0, 8 * 8 * 64 * 4, 0)); hipLaunchKernelGGL((k_passes_dyn<2, false>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, counters, out); });
        printf("bands %d  blocks/CU %d : %.4f | %.4f | pipelined, ordinary loads %.4f ms\n", bands, per_cu, a, b, c);
    }
    printf("## xcd2: every XCD group owns one band and streams the whole matrix: plain | pipelined\n");
    for (int bands : {2, 4, 8}) for (int per_cu : {2, 3, 4, 6, 8}) {
        const int grid = per_cu * cus, bc = (cols + bands - 1) / bands;
        float a = time_ms([&] { hipLaunchKernelGGL((k_xcd2<0, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        float b = time_ms([&] { hipLaunchKernelGGL((k_xcd2<2, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        float c = time_ms([&] { hipLaunchKernelGGL((k_xcd2<2, false>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        printf("bands %d  blocks/CU %d : %.4f | %.4f | pipelined, ordinary loads %.4f ms\n", bands, per_cu, a, b, c);
    }
    if (argc > 3) return 0;
    printf("## passes: chip-wide phases, one band per phase, all blocks walk all tiles (B full streams + in-band gathers)\n");
    for (int bands : {1, 2, 3, 4, 6}) for (int per_cu : {4, 6, 8}) {
        const int grid = per_cu * cus, bc = (cols + bands - 1) / bands;
        float a = time_ms([&] { hipLaunchKernelGGL((k_passes<false, false, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        float b = time_ms([&] { hipLaunchKernelGGL((k_passes<true, false, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        float c = time_ms([&] { hipLaunchKernelGGL((k_passes<false, true, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        float d = time_ms([&] { hipLaunchKernelGGL((k_passes<true, true, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        printf("bands %d  blocks/CU %d : plain %.4f | prefetch %.4f | lazy values %.4f | prefetch + lazy values %.4f ms\n", bands, per_cu, a, b, c, d);
    }
    printf("## xcd: every XCD (group) owns one band for the whole call and streams the whole matrix (no phases)\n");
    for (int bands : {2, 4, 8}) for (int per_cu : {4, 8}) {
        const int grid = per_cu * cus, bc = (cols + bands - 1) / bands;
        float a = time_ms([&] { hipLaunchKernelGGL((k_xcd<false, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        float b = time_ms([&] { hipLaunchKernelGGL((k_xcd<true, true>), dim3(grid), dim3(BLOCK), 0, 0, c4, v4, x, n4, bands, bc, out); });
        printf("bands %d  blocks/CU %d : plain %.4f | prefetch %.4f ms\n", bands, per_cu, a, b);
    }
    printf("## gather2: B cols-only passes storing x[col] per nonzero, then ONE val * xg stream\n");
    for (int bands : {3, 4}) for (int per_cu : {4, 8}) {
        const int grid = per_cu * cus, bc = (cols + bands - 1) / bands;
        float a = time_ms([&] {
            for (int b = 0; b < bands; ++b) hipLaunchKernelGGL((k_gather_pass<true>), dim3(grid), dim3(BLOCK), 0, 0, c4, x, (float4v*) xg, n4, b, bc);
            hipLaunchKernelGGL(k_dot, dim3(8 * cus), dim3(BLOCK), 0, 0, v4, (const float4v*) xg, n4, out);
        });
        printf("bands %d  blocks/CU %d : %.4f ms\n", bands, per_cu, a);
    }
    return 0;
}
