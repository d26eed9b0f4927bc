// feasibility probe: column-banded traversal of a CSR matrix whose x does not fit the XCD's L2.
// block b runs on XCD b % 8 (observed round-robin); XCD k handles column band k % NB, so each L2
// only ever gathers from a 1/NB slice of x.  splits[(band) * rows + r] = first nonzero of row r
// whose column is >= band * band_width (splits for band NB = row end).
#include <hip/hip_runtime.h>
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
