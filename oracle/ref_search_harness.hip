// TEST INFRASTRUCTURE: builds the REFERENCE's own cub::MergePathSearch
// (cub/thread/thread_search.cuh:53-84) and ReduceByKeyOp
// (cub/thread/thread_operators.cuh:278-302) as host code, straight from the
// read-only reference tree, so the oracle restatement can be pinned against
// it.  hipcc understands __host__ __device__ natively; nothing is stubbed.
// Output goes to oracle/_ref/ only (git-ignored).  Never shipped as product.
//
// usage: ref_search <offsets-file>
//   file: "rows nnz" then rows+1 row offsets.  Prints, for every diagonal
//   d in [0, rows+nnz+3], "d x y" as computed by the reference.
// usage: ref_search rbk <file>
//   file: n then n "(key value)" pairs; prints the inclusive left-to-right
//   fold of cub::ReduceByKeyOp<cub::Sum> over the pairs.
#include <hip/hip_runtime.h>
#include <iterator>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "cub/util_macro.cuh"
#include "cub/thread/thread_search.cuh"
#include "cub/thread/thread_operators.cuh"

struct Counting {
    typedef int value_type; typedef int difference_type; typedef int* pointer;
    typedef int reference; typedef std::random_access_iterator_tag iterator_category;
    int base;
    __host__ __device__ int operator[](int i) const { return base + i; }
};
struct Coord { int x, y; };

int main(int argc, char** argv)
{
    if (argc == 3 && !strcmp(argv[1], "rbk")) {
        FILE* f = fopen(argv[2], "r"); if (!f) return 2;
        int n; if (fscanf(f, "%d", &n) != 1) return 2;
        typedef cub::KeyValuePair<int, double> Pair;
        cub::ReduceByKeyOp<cub::Sum> op;
        Pair acc; 
        for (int i = 0; i < n; ++i) {
            Pair p; if (fscanf(f, "%d %lf", &p.key, &p.value) != 2) return 2;
            acc = (i == 0) ? p : op(acc, p);
            printf("%d %.17g\n", acc.key, acc.value);
        }
        return 0;
    }
    if (argc != 2) { fprintf(stderr, "usage\n"); return 1; }
    FILE* f = fopen(argv[1], "r"); if (!f) return 2;
    int rows, nnz; if (fscanf(f, "%d %d", &rows, &nnz) != 2) return 2;
    std::vector<int> off(rows + 1);
    for (int i = 0; i <= rows; ++i) if (fscanf(f, "%d", &off[i]) != 1) return 2;
    Counting b; b.base = 0;
    for (int d = 0; d <= rows + nnz + 3; ++d) {
        Coord c;
        // row END offsets = row_offsets + 1 (device_spmv.cuh:148)
        cub::MergePathSearch(d, off.data() + 1, b, rows, nnz, c);
        printf("%d %d %d\n", d, c.x, c.y);
    }
    return 0;
}
