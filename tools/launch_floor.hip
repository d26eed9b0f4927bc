// tools/launch_floor.hip -- what does ONE call cost in the reference's timing loop (N back-to-back launches between two events,
// gpu_spmv.cu:418-434) before the kernel does any SpMV work?  Times loops of 2000 launches of synthetic kernels that have the
// footprint of the small tile shape (256 threads, 20.5 KB of static LDS) and a chain of K dependent memory round trips
// (K = 0: empty; each trip = one load whose address depends on the previous one's value, then one store), over grids of
// 1 ... 1400 blocks.  The SpMV's own figures (3.4-3.6 us per call below 300 tiles, 5.0 at 697) are to be read against these:
// a lone block of the one-launch kernel makes three dependent trips (hints -> streams -> x) and one store.
// build: make -C tools launch_floor ; run on the GPU box: tools/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int TRIPS, int LDS_BYTES, bool SCALAR_FIRST>
__global__ __launch_bounds__(256) void chain_kernel(const int *__restrict__ table, int *__restrict__ out, int mask)
{
    __shared__ int s_pad[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    if (TRIPS == 0) return;
    int idx = (blockIdx.x * 256 + threadIdx.x) & mask;
    int first = 0;
    if (SCALAR_FIRST) {                      // trip 1 as a scalar load on a block-uniform address (the hint load's shape)
        const int *p = table + (blockIdx.x & mask);
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(first) : "s"(p) : "memory");
        idx = (idx + first) & mask;
    }
#pragma unroll
    for (int k = SCALAR_FIRST ? 1 : 0; k < TRIPS; ++k) idx = (idx + table[idx]) & mask;      // dependent loads
    if (LDS_BYTES > 0) { s_pad[threadIdx.x] = idx; __syncthreads(); idx = s_pad[threadIdx.x ^ 1]; }
    out[blockIdx.x * 256 + threadIdx.x] = idx;
}

template <int TRIPS, int LDS_BYTES, bool SCALAR_FIRST>
float loop_us(int blocks, const int *table, int *out, int mask, int iters)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) chain_kernel<TRIPS, LDS_BYTES, SCALAR_FIRST><<<blocks, 256>>>(table, out, mask);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) chain_kernel<TRIPS, LDS_BYTES, SCALAR_FIRST><<<blocks, 256>>>(table, out, mask);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return best * 1000.0f / iters;
}

int main()
{
    const int n = 1 << 22, mask = n - 1;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (int) ((i * 2654435761u) >> 9) & 1023;      // small hops: stays in the caches after warm-up
    int *table, *out;
    CK(hipMalloc(&table, n * sizeof(int))); CK(hipMalloc(&out, 1400 * 256 * sizeof(int)));
    CK(hipMemcpy(table, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
    const int grids[] = {1, 3, 28, 251, 697, 1003, 1366};
    printf("us per call, loop of 2000 back-to-back launches (best of 3 loops), 256-thread blocks\n");
    printf("%-58s", "kernel \\ blocks");
    for (int g : grids) printf("%8d", g);
    printf("\n");
#define ROW(name, T, L, S) do { printf("%-58s", name); for (int g : grids) printf("%8.2f", loop_us<T, L, S>(g, table, out, mask, 2000)); printf("\n"); } while (0)
    ROW("empty, no LDS", 0, 0, false);
    ROW("empty, 20.5 KB LDS declared", 0, 20992, false);
    ROW("1 vector trip + store", 1, 0, false);
    ROW("2 dependent vector trips + store", 2, 0, false);
    ROW("3 dependent vector trips + store", 3, 0, false);
    ROW("3 trips (first scalar) + store", 3, 0, true);
    ROW("3 trips (first scalar) + LDS exchange/barrier + store, 20.5 KB", 3, 20992, true);
    ROW("4 dependent vector trips + store", 4, 0, false);
    return 0;
}
