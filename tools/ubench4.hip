// gather-rate vs address pattern (no index stream: indices computed from lane/iteration)
#include <hip/hip_runtime.h>
template <int MODE>
__global__ __launch_bounds__(256) void k_pat(const float* __restrict__ x, unsigned mask, int iters, float* out)
{
    const unsigned lane = threadIdx.x & 63;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned k = it * 8 + u;
            if (MODE == 0) idx[u] = k & mask;                              // all lanes same address
            else if (MODE == 1) idx[u] = (lane + 64 * k) & mask;           // consecutive dwords across lanes
            else if (MODE == 2) idx[u] = ((4 * lane + (k & 3)) + 256 * (k >> 2)) & mask;  // stride-4 (vector-chunk layout)
            else if (MODE == 3) { h = h * 1664525u + 1013904223u; idx[u] = (h >> 8) & mask; }   // random per lane
            else if (MODE == 4) { unsigned g = (lane >> 2) * 2654435761u + k * 40503u; idx[u] = (((g >> 7) << 2) + (lane & 3)) & mask; } // runs of 4 consecutive, random run start
            else if (MODE == 5) { unsigned g = (lane >> 3) * 2654435761u + k * 40503u; idx[u] = (((g >> 7) << 3) + (lane & 7)) & mask; } // runs of 8
            else if (MODE == 6) { unsigned g = (lane >> 4) * 2654435761u + k * 40503u; idx[u] = (((g >> 7) << 4) + (lane & 15)) & mask; } // runs of 16
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[idx[u]];
    }
    if (acc == 12345.678f) out[0] = acc;
}
extern "C" int ub4(const void* x, unsigned mask, int iters, void* out, int blocks, int mode, void* s)
{
#define C(M) case M: hipLaunchKernelGGL((k_pat<M>), dim3(blocks), dim3(256), 0, (hipStream_t) s, (const float*) x, mask, iters, (float*) out); break;
    switch (mode) { C(0) C(1) C(2) C(3) C(4) C(5) C(6) }
    return (int) hipGetLastError();
}
