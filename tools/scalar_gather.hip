// tools/scalar_gather.hip -- does the SCALAR memory path (s_load through the scalar data cache) add gather bandwidth to the
// vector path's?  A gather that hits L2 moves a whole 128-byte line from L2 to the CU's vector L1: 262 G gathers/s on this
// chip whatever the load's cache bits (profiles/r02_hw_ceilings.txt), which is what bounds BASELINE config 2.  The scalar
// cache is a different client of L2 with 64-byte lines; this probe measures random 4-byte gathers from an L2-sized table
//   v   all through vector loads (the known figure)
//   s   all through scalar loads (v_readlane -> s_load_dword, 16 in flight)
//   m   64 vector + S scalar gathers per round, S = 8, 16, 32
// Development aid, not part of the product.  Build: make -C tools scalar_gather ; run: tools/scalar_gather [table MB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned) (z ^ (z >> 31));
}
__global__ void k_idx(int* idx, size_t n, unsigned mask) { for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) idx[i] = (int) (mix(i) & mask); }
__global__ void k_tab(unsigned* t, size_t n) { for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) t[i] = (unsigned) i * 2654435761u; }

// one scalar load of table[i] for a wave-uniform i (inline asm: the choice is not left to the compiler)
__device__ __forceinline__ unsigned sload(const unsigned* table, int i)
{
    unsigned r;
    const unsigned off = (unsigned) i * 4u;
    asm volatile("s_load_dword %0, %1, %2" : "=s"(r) : "s"(table), "s"(off) : "memory");
    return r;
}
// (the loaded registers pass through the wait as operands: nothing that uses them can be scheduled before it)
__device__ __forceinline__ void swait8(unsigned (&t)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t[0]), "+s"(t[1]), "+s"(t[2]), "+s"(t[3]), "+s"(t[4]), "+s"(t[5]), "+s"(t[6]), "+s"(t[7]) :: "memory");
}

// VEC: vector gathers per lane and round (each 64 lanes wide); SC: scalar gathers per round (indices taken lane by lane from further index words)
template <int VEC, int SC>
__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ idx, const unsigned* __restrict__ table, size_t rounds, unsigned* __restrict__ out)
{
    constexpr int SW = (SC + 63) / 64;                 // index words of the scalar part
    constexpr int STRIDE = 64 * 6;                      // index words per round (4 vector + 2 scalar at most)
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((size_t) gridDim.x * blockDim.x) >> 6;
    unsigned acc = 0, sacc = 0;
    for (size_t r = wave; r < rounds; r += waves) {
        int iv[VEC > 0 ? VEC : 1], is[SW > 0 ? SW : 1];
#pragma unroll
        for (int k = 0; k < VEC; ++k) iv[k] = __builtin_nontemporal_load(idx + r * STRIDE + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < SW; ++k) is[k] = __builtin_nontemporal_load(idx + r * STRIDE + (4 + k) * 64 + lane);
        unsigned got[VEC > 0 ? VEC : 1];
#pragma unroll
        for (int k = 0; k < VEC; ++k) got[k] = table[iv[k]];
#pragma unroll
        for (int b = 0; b < SC; b += 8) {
            unsigned t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = sload(table, __builtin_amdgcn_readlane(is[b / 64], (b + j) % 64));
            swait8(t);
#pragma unroll
            for (int j = 0; j < 8; ++j) sacc += t[j];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc += got[k];
    }
    acc += sacc;
    if (acc == 0x12345678u) out[0] = acc;
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

int main(int argc, char** argv)
{
    const int mb = argc > 1 ? atoi(argv[1]) : 2;
    const size_t entries = (size_t) mb << 18;                   // 4-byte entries
    const size_t rounds = 1 << 19;                               // 6 x 64 index words per round
    int* idx; unsigned *table, *out;
    CK(hipMalloc(&idx, rounds * 384 * 4)); CK(hipMalloc(&table, entries * 4)); CK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(k_idx, dim3(4096), dim3(256), 0, 0, idx, rounds * 384, (unsigned) (entries - 1));
    hipLaunchKernelGGL(k_tab, dim3(1024), dim3(256), 0, 0, table, entries);
    CK(hipDeviceSynchronize());
    int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("# scalar_gather: table %d MB, %zu rounds, %d CUs; G gathers/s (vector part | scalar part | both)\n", mb, rounds, cus);
    for (int per_cu : {4, 8}) {
        const int grid = per_cu * cus;
        auto rep = [&](const char* name, float ms, int vec, int sc) {
            printf("blocks/CU %d  %-26s: %.4f ms   %7.1f | %7.1f | %7.1f\n", per_cu, name, ms, vec * rounds / ms * 1e-6, sc * rounds / ms * 1e-6, (vec + sc) * rounds / ms * 1e-6);
        };
#define RUN(V, S) rep("vector " #V " x 64 + scalar " #S, time_ms([&] { hipLaunchKernelGGL((k_gather<V, S>), dim3(grid), dim3(256), 0, 0, idx, table, rounds, out); }), V * 64, S)
        RUN(1, 0); RUN(2, 0); RUN(4, 0); RUN(0, 64); RUN(0, 128);
        RUN(4, 16); RUN(4, 32); RUN(4, 64); RUN(4, 96); RUN(4, 128); RUN(2, 32); RUN(2, 64);
    }
    return 0;
}
