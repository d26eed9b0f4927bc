// tools/gather_granularity.hip -- how many bytes does one random 8-byte gather cost when the table is far beyond every cache?
// MI355X fetches whole 128-byte lines into L2 on a miss (TCC_EA0_RDREQ_128B), so a DRAM-resident gather moves 16 x the bytes it
// uses.  This microbenchmark times random gathers over a table of `MB` megabytes for every cache-policy flavour of the load
// instruction (sc0 / sc1 / nt bits) and for three kinds of allocation (ordinary, fine-grained, uncached), to see whether any of
// them makes the memory system fetch less than a line.  build: make -C tools gather_granularity
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ void ld_issue(double &v, const double *p)       // the load only; ld_wait() below before v is used
{
    if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 6) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld_wait(double (&v)[8])
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
}

// every thread: PER independent random gathers per round (issued back to back, one wait), ROUNDS rounds
template <int MODE, int PER>
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ table, unsigned long long mask, int rounds, double *out)
{
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    double acc = 0;
    for (int r = 0; r < rounds; ++r) {
        unsigned long long idx[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) { s = s * 6364136223846793005ull + 1442695040888963407ull; idx[k] = (s >> 20) & mask; }
        double v[PER];
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < PER; ++k) v[k] = table[idx[k]];
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) ld_issue<MODE>(v[k], table + idx[k]);
            ld_wait(v);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) acc += v[k];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

template <int MODE>
static double run(const double *t, unsigned long long mask, double *out, int blocks, int rounds)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<MODE, 8>), dim3(blocks), dim3(256), 0, 0, t, mask, 2, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<MODE, 8>), dim3(blocks), dim3(256), 0, 0, t, mask, rounds, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return (double) blocks * 256 * 8 * rounds / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t) atoll(argv[1]) : 512;
    size_t n = 1; while (n * 2 * 8 <= mb << 20) n *= 2;              // power of two entries
    const unsigned long long mask = n - 1;
    const int blocks = 256 * 8, rounds = 64;
    double *out; CK(hipMalloc(&out, 64));
    const char *kinds[3] = {"hipMalloc", "fine-grained", "uncached"};
    printf("# random 8-byte gathers over a %zu MB table (%zu entries), %d blocks x 256 threads x 8 x %d; G gathers/s\n", n * 8 >> 20, n, blocks, rounds);
    printf("# %-13s %9s %9s %9s %9s %9s %9s %9s %9s\n", "allocation", "plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0sc1nt", "sc1 nt", "sc0 nt");
    for (int kind = 0; kind < 3; ++kind) {
        double *t = nullptr;
        hipError_t e = kind == 0 ? hipMalloc(&t, n * 8) : hipExtMallocWithFlags((void **) &t, n * 8, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("  %-13s allocation failed: %s\n", kinds[kind], hipGetErrorString(e)); (void) hipGetLastError(); continue; }
        CK(hipMemset(t, 0, n * 8));
        CK(hipDeviceSynchronize());
        printf("  %-13s %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f\n", kinds[kind],
               run<0>(t, mask, out, blocks, rounds), run<1>(t, mask, out, blocks, rounds), run<2>(t, mask, out, blocks, rounds), run<3>(t, mask, out, blocks, rounds),
               run<4>(t, mask, out, blocks, rounds), run<5>(t, mask, out, blocks, rounds), run<6>(t, mask, out, blocks, rounds), run<7>(t, mask, out, blocks, rounds));
        fflush(stdout);
        CK(hipFree(t));
    }
    return 0;
}
