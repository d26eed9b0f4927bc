// tools/ubench.hip -- hardware-ceiling microbenchmarks (development aid):
// what a pure stream and a pure random gather reach on this chip, to price
// the tile kernel against.  Built to tools/libubench.so, driven by tools/ubench.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
