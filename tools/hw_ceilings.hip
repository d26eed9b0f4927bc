// tools/hw_ceilings.hip -- the hardware-ceiling microbenchmarks DESIGN.md prices the kernels against (development aid,
// not part of the product): what a pure stream, a random gather (by table size, cache policy and address pattern) and a
// column-banded CSR traversal reach on an MI355X.  Built to tools/libhw_ceilings.so by `make -C tools`; driven by
// tools/hw_ceilings.py <probe>, whose output is kept under profiles/ (r02_hw_ceilings_*.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float float4v __attribute__((ext_vector_type(4)));

// ===== stream / gather vs table size =====
// tools/ubench.hip -- hardware-ceiling microbenchmarks (development aid):
// what a pure stream and a pure random gather reach on this chip, to price
// the tile kernel against.  Built to tools/libubench.so, driven by tools/ubench.py.

// sum of a float4 stream (read-only): n4 float4 elements
__global__ __launch_bounds__(256) void k_stream_f4(const float4* __restrict__ a, size_t n4, float* out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) {
        float4 v = a[i]; acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// dword stream, coalesced, one element per thread per iteration
__global__ __launch_bounds__(256) void k_stream_f1(const float* __restrict__ a, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) acc += a[i];
    if (acc == 12345.678f) out[0] = acc;
}
// gather: idx stream (coalesced dword) -> x[idx]; UNROLL independent gathers in flight per thread
template <int UNROLL>
__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ idx, const float* __restrict__ val,
                                                const float* __restrict__ x, size_t n, float* out)
{
    float acc = 0.f;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        int c[UNROLL]; float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { c[u] = idx[i + u * stride]; v[u] = val ? val[i + u * stride] : 1.f; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u] * x[c[u]];
    }
    for (; i < n; i += stride) acc += (val ? val[i] : 1.f) * x[idx[i]];
    if (acc == 12345.678f) out[0] = acc;
}

extern "C" {
int ub_stream_f4(const void* a, size_t bytes, void* out, int blocks, void* stream)
{
    hipLaunchKernelGGL(k_stream_f4, dim3(blocks), dim3(256), 0, (hipStream_t) stream, (const float4*) a, bytes / 16, (float*) out);
    return (int) hipGetLastError();
}
int ub_stream_f1(const void* a, size_t bytes, void* out, int blocks, void* stream)
{
    hipLaunchKernelGGL(k_stream_f1, dim3(blocks), dim3(256), 0, (hipStream_t) stream, (const float*) a, bytes / 4, (float*) out);
    return (int) hipGetLastError();
}
int ub_gather(const void* idx, const void* val, const void* x, size_t n, void* out, int blocks, int unroll, void* stream)
{
    hipStream_t s = (hipStream_t) stream;
    if (unroll == 1) hipLaunchKernelGGL((k_gather<1>), dim3(blocks), dim3(256), 0, s, (const int*) idx, (const float*) val, (const float*) x, n, (float*) out);
    else if (unroll == 4) hipLaunchKernelGGL((k_gather<4>), dim3(blocks), dim3(256), 0, s, (const int*) idx, (const float*) val, (const float*) x, n, (float*) out);
    else hipLaunchKernelGGL((k_gather<8>), dim3(blocks), dim3(256), 0, s, (const int*) idx, (const float*) val, (const float*) x, n, (float*) out);
    return (int) hipGetLastError();
}
}

// ===== stream issue shapes =====

template <int MODE>
__global__ __launch_bounds__(256) void k_stream(const float4v* __restrict__ a, size_t n4, float* out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) {
        float4v v;
        if (MODE == 0) v = a[i];
        else if (MODE == 1) v = __builtin_nontemporal_load(&a[i]);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// two interleaved streams per block-tile like the SpMV staging: block b reads [b*T, (b+1)*T) of each array
template <int MODE>
__global__ __launch_bounds__(256) void k_tiles(const float4v* __restrict__ a, const float4v* __restrict__ b, size_t n4, int per_block4, float* out)
{
    float acc = 0.f;
    for (size_t t = blockIdx.x; t * per_block4 < n4; t += gridDim.x) {
        size_t base = t * per_block4;
        for (int k = threadIdx.x; k < per_block4; k += 256) {
            size_t i = base + k; if (i >= n4) break;
            float4v v, w;
            if (MODE == 0) { v = a[i]; w = b[i]; } else { v = __builtin_nontemporal_load(&a[i]); w = __builtin_nontemporal_load(&b[i]); }
            acc += v.x + v.y + v.z + v.w + w.x + w.y + w.z + w.w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
extern "C" int ub3_stream(const void* a, size_t bytes, void* out, int blocks, int mode, void* s)
{
    if (mode == 0) hipLaunchKernelGGL((k_stream<0>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float4v*) a, bytes / 16, (float*) out);
    else hipLaunchKernelGGL((k_stream<1>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float4v*) a, bytes / 16, (float*) out);
    return (int) hipGetLastError();
}
extern "C" int ub3_tiles(const void* a, const void* b, size_t bytes_each, int per_block4, void* out, int blocks, int mode, void* s)
{
    if (mode == 0) hipLaunchKernelGGL((k_tiles<0>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float4v*) a, (const float4v*) b, bytes_each / 16, per_block4, (float*) out);
    else hipLaunchKernelGGL((k_tiles<1>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float4v*) a, (const float4v*) b, bytes_each / 16, per_block4, (float*) out);
    return (int) hipGetLastError();
}

// ===== gather cache policy =====
// gather cache-policy probe: random dword gathers over a table, with the cache-policy bits of
// global_load_dword set explicitly (gfx950: sc0, sc1, nt).  POL: 0 plain, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt
template <int POL>
__device__ __forceinline__ float ld(const float* p)
{
    float v;
    if (POL == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POL == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (POL == 2) asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (POL == 3) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == 5) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL>
__global__ __launch_bounds__(256) void k_gather_policy(const float* __restrict__ x, unsigned mask, int iters, float* out)
{
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { h = h * 1664525u + 1013904223u; v[u] = ld<POL>(x + ((h >> 7) & mask)); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 12345.678f) out[0] = acc;
}
extern "C" int ub5(const void* x, unsigned mask, int iters, void* out, int blocks, int pol, void* s)
{
#define C(M) case M: hipLaunchKernelGGL((k_gather_policy<M>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float*) x, mask, iters, (float*) out); break;
    switch (pol) { C(0) C(1) C(2) C(3) C(4) C(5) }
    return (int) hipGetLastError();
}

// ===== gather address pattern =====
// gather-rate vs address pattern for 4-byte and 8-byte elements (L1-resident table)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_pat(const T* __restrict__ x, unsigned mask, int iters, T* out)
{
    const unsigned lane = threadIdx.x & 63;
    T acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned k = it * 8 + u;
            if (MODE == 0) idx[u] = k & mask;                                        // all lanes same address
            else if (MODE == 1) idx[u] = (lane + 64 * k) & mask;                     // consecutive across lanes
            else if (MODE == 2) idx[u] = ((4 * lane + (k & 3)) + 256 * (k >> 2)) & mask;   // stride-4 (chunk layout, dense row)
            else if (MODE == 3) idx[u] = ((16 * (lane >> 2) + 4 * (k & 3) + (lane & 3)) + 256 * (k >> 2)) & mask;  // quad-transposed chunk layout: runs of 4, quads 16 apart
            else if (MODE == 4) idx[u] = ((64 * (lane >> 4) + 16 * (k & 3) + (lane & 15)) + 256 * (k >> 2)) & mask;  // 16-lane transposed: runs of 16
            else if (MODE == 5) idx[u] = (16 * (lane >> 2) + 2 * (lane & 3) + (k & 1) + 256 * (k >> 2)) & mask;       // quad: stride 2 inside 32 B
            else if (MODE == 6) idx[u] = (16 * (lane >> 2) + ((0x1302 >> (4 * (lane & 3))) & 3) + 4 * (k & 3) + 256 * (k >> 2)) & mask;  // quad: run of 4, permuted order
            else if (MODE == 7) idx[u] = (16 * (lane >> 2) + ((lane & 3) == 3 ? 2 : (lane & 3)) + 4 * (k & 3) + 256 * (k >> 2)) & mask;   // quad: 0,1,2,2
            else if (MODE == 8) idx[u] = (16 * (lane >> 2) + (lane & 3) + 1 + 4 * (k & 3) + 256 * (k >> 2)) & mask;       // quad: run of 4 starting at 4n+1
            else if (MODE == 9) idx[u] = (16 * (lane >> 2) + (lane & 3) + 2 + 4 * (k & 3) + 256 * (k >> 2)) & mask;       // quad: run of 4 starting at 4n+2
            else if (MODE == 10) idx[u] = (16 * (lane >> 2) + ((lane & 3) == 3 ? 9 : (lane & 3)) + 4 * (k & 3) + 256 * (k >> 2)) & mask;  // quad: 0,1,2,9 (run of 3 + far one)
            else if (MODE == 11) idx[u] = (16 * (lane >> 2) + ((lane & 1) + 8 * ((lane >> 1) & 1)) + 256 * (k >> 2) + 2 * (k & 3)) & mask;  // quad: two runs of 2 (0,1,8,9)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[idx[u]];
    }
    if (acc == (T) 12345.678) out[0] = acc;
}
extern "C" int ub6(const void* x, unsigned mask, int iters, void* out, int blocks, int mode, int dbl, void* s)
{
#define C(M) case M: if (dbl) hipLaunchKernelGGL((k_pat<double, M>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const double*) x, mask, iters, (double*) out); \
                     else hipLaunchKernelGGL((k_pat<float, M>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float*) x, mask, iters, (float*) out); break;
    switch (mode) { C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) }
    return (int) hipGetLastError();
}

// ===== column-banded traversal probe =====
// feasibility probe: column-banded traversal of a CSR matrix whose x does not fit the XCD's L2.
// block b runs on XCD b % 8 (observed round-robin); XCD k handles column band k % NB, so each L2
// only ever gathers from a 1/NB slice of x.  splits[(band) * rows + r] = first nonzero of row r
// whose column is >= band * band_width (splits for band NB = row end).
template <int NB>
__global__ __launch_bounds__(256) void k_banded(const float* __restrict__ val, const int* __restrict__ col,
                                                const int* __restrict__ splits, const float* __restrict__ x,
                                                float* __restrict__ ypart, int rows, int rows_per_block)
{
    const int xcd = blockIdx.x & 7;
    const int band = xcd % NB;
    const int rep = xcd / NB;                      // 8 / NB blocks share a band
    const int chunk = (blockIdx.x >> 3) * (8 / NB) + rep;
    const int r0 = chunk * rows_per_block;
    const int lane8 = threadIdx.x & 7;             // 8 lanes per row piece
    const int* __restrict__ lo = splits + (size_t) band * rows;
    const int* __restrict__ hi = splits + (size_t) (band + 1) * rows;
    for (int r = r0 + (threadIdx.x >> 3); r < r0 + rows_per_block && r < rows; r += 32) {
        const int a = lo[r], b = hi[r];
        float sum = 0.f;
        for (int j = a + lane8; j < b; j += 8) sum += __builtin_nontemporal_load(val + j) * x[__builtin_nontemporal_load(col + j)];
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        if (lane8 == 0) ypart[(size_t) band * rows + r] = sum;
    }
}
__global__ void k_combine(const float* __restrict__ ypart, float* __restrict__ y, int rows, int nb)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) { float s = ypart[r]; for (int b = 1; b < nb; ++b) s += ypart[(size_t) b * rows + r]; y[r] = s; }
}
// split finder: one thread per (row), binary search of NB-1 boundaries in the row's sorted columns
__global__ void k_splits(const int* __restrict__ off, const int* __restrict__ col, int* __restrict__ splits, int rows, int nb, int band_width)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int a = off[r], b = off[r + 1];
    splits[r] = a; splits[(size_t) nb * rows + r] = b;
    int lo = a;
    for (int k = 1; k < nb; ++k) {
        int l = lo, h = b; const int key = k * band_width;
        while (l < h) { const int m = (l + h) >> 1; if (col[m] < key) l = m + 1; else h = m; }
        splits[(size_t) k * rows + r] = l; lo = l;
    }
}
extern "C" int ub7_banded(const void* val, const void* col, const void* splits, const void* x, void* ypart, void* y, int rows, int nb, int rows_per_block, void* s)
{
    const int chunks = (rows + rows_per_block - 1) / rows_per_block;
    const int per = 8 / nb;
    const int blocks = ((chunks + per - 1) / per) * 8;
    if (nb == 4) hipLaunchKernelGGL((k_banded<4>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float*) val, (const int*) col, (const int*) splits, (const float*) x, (float*) ypart, rows, rows_per_block);
    else if (nb == 8) hipLaunchKernelGGL((k_banded<8>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float*) val, (const int*) col, (const int*) splits, (const float*) x, (float*) ypart, rows, rows_per_block);
    else if (nb == 2) hipLaunchKernelGGL((k_banded<2>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float*) val, (const int*) col, (const int*) splits, (const float*) x, (float*) ypart, rows, rows_per_block);
    else hipLaunchKernelGGL((k_banded<1>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float*) val, (const int*) col, (const int*) splits, (const float*) x, (float*) ypart, rows, rows_per_block);
    hipLaunchKernelGGL(k_combine, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t) s, (const float*) ypart, (float*) y, rows, nb);
    return (int) hipGetLastError();
}
extern "C" int ub7_splits(const void* off, const void* col, void* splits, int rows, int nb, int band_width, void* s)
{
    hipLaunchKernelGGL(k_splits, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t) s, (const int*) off, (const int*) col, (int*) splits, rows, nb, band_width);
    return (int) hipGetLastError();
}

// ===== scalar-load gather probe =====
// The vector L1 (TCP) asks L2 for whole 128-byte lines; the scalar data cache works on 64-byte lines.  Does a gather issued
// as 64 scalar loads per wave (v_readlane -> s_load_dword, address made wave-uniform) move fewer bytes per miss, and how
// fast can a CU issue it?  Same index stream as k_gather_policy.
__global__ __launch_bounds__(256) void k_gather_scalar(const float* __restrict__ x, unsigned mask, int iters, float* out)
{
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            const int idx = (int) ((h >> 7) & mask);
#pragma unroll
            for (int l = 0; l < 64; ++l) {
                const int s = __builtin_amdgcn_readlane(idx, l);
                acc += x[s];                                   // uniform address: s_load_dword
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
extern "C" int ub_gather_scalar(const void* x, unsigned mask, int iters, void* out, int blocks, void* s)
{
    hipLaunchKernelGGL(k_gather_scalar, dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float*) x, mask, iters, (float*) out);
    return (int) hipGetLastError();
}
