// gather-rate vs address pattern for 4-byte and 8-byte elements (L1-resident table)
#include <hip/hip_runtime.h>
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
