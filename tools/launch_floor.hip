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
#include <chrono>
#include <cstring>
#include <algorithm>
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

// ---- the HOST side of a launch: the reference's loop is bound by the enqueueing thread below ~300 tiles (tools/ab_driver: host
// enqueue time per call == loop time per call there), so what a call costs on the host is what the loop measures.  An empty kernel
// with the one-launch kernel's argument block (176 bytes), enqueued 2000 times through (a) hipLaunchKernelGGL, (b) hipModuleLaunchKernel
// on the hipFunction_t of the same kernel (hipGetFuncBySymbol) with the arguments as ONE pre-packed buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER)
struct BigArgs { void *a[9]; int i[6]; double d[2]; void *b[6]; int j[6]; };
__global__ __launch_bounds__(256) void big_args_kernel(BigArgs g) { if (g.i[0] == 0x7fffffff) *(int *) g.a[0] = g.j[5]; }

static void host_side()
{
    BigArgs g; memset(&g, 0, sizeof(g));
    printf("\nhost enqueue / loop time (us per call, 2000 calls, best of 5 loops), empty kernel with a %zu-byte argument block, 28 blocks\n", sizeof(g));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipFunction_t fn = nullptr;
    const hipError_t got = hipGetFuncBySymbol(&fn, (const void *) big_args_kernel);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 1 && (got != hipSuccess || !fn)) { printf("hipGetFuncBySymbol unavailable (%d)\n", (int) got); break; }
        double best_host = 1e30; float best_loop = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(a, 0));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 2000; ++i) {
                if (mode == 0) big_args_kernel<<<28, 256>>>(g);
                else {
                    size_t sz = sizeof(g);
                    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &g, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
                    CK(hipModuleLaunchKernel(fn, 28, 1, 1, 256, 1, 1, 0, 0, nullptr, extra));
                }
            }
            const double host = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0) { best_host = std::min(best_host, host); best_loop = std::min(best_loop, ms * 1000.0f / 2000); }
        }
        printf("%-72s host %.2f  loop %.2f\n", mode == 0 ? "hipLaunchKernelGGL (<<<>>>)" : "hipModuleLaunchKernel(hipGetFuncBySymbol(kernel), packed argument buffer)", best_host, best_loop);
    }
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
    host_side();
    return 0;
}
