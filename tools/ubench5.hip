// gather cache-policy probe: random dword gathers over a table, with the cache-policy bits of
// global_load_dword set explicitly (gfx950: sc0, sc1, nt).  POL: 0 plain, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt
#include <hip/hip_runtime.h>
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
