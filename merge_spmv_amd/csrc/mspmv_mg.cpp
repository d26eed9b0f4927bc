// mspmv_mg.cpp -- host side of the multi-GPU merge partitioning declared in
// include/mspmv.h.  New design: the reference has no
// multi-device code, only the claim that the merge decomposition "is suitable
// for recursively partitioning CSR datasets" (reference README.md:5) and the
// thread-level scheme of cpu_spmv.cpp:305-352, which this lifts one level:
// a GPU plays the role of an OpenMP thread, its swath being
// ceil((rows+nnz)/parts) merge items.  64-bit throughout (rows+nnz of the
// global problem may exceed 2^31; each part must fit int32).
#include <cstdint>
#include "../../include/mspmv.h"

namespace {

constexpr int kInvalidValue = 1;   // hipErrorInvalidValue

// thread_search.cuh:53-84 in 64 bits; row_end = row_offsets + 1.
void search64(int64_t diagonal, const int64_t *row_end, int64_t rows, int64_t nnz, int64_t &x, int64_t &y)
{
    int64_t lo = diagonal - nnz; if (lo < 0) lo = 0;
    int64_t hi = diagonal < rows ? diagonal : rows;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (row_end[mid] <= diagonal - mid - 1) lo = mid + 1; else hi = mid;
    }
    x = lo < rows ? lo : rows;
    y = diagonal - lo;
}

}  // namespace

extern "C" {

int mspmv_mg_partition(const int64_t *h_row_offsets, int64_t rows, int64_t nnz, int32_t parts, int64_t *row_split,
                       int64_t *nz_split)
{
    if (!h_row_offsets || !row_split || !nz_split || rows < 0 || nnz < 0 || parts < 1) return kInvalidValue;
    const int64_t total = rows + nnz;
    const int64_t per_part = (total + parts - 1) / parts;     // cpu_spmv.cpp:312
    for (int32_t g = 0; g <= parts; ++g) {
        int64_t d = per_part * g; if (d > total) d = total;    // cpu_spmv.cpp:317-318
        search64(d, h_row_offsets + 1, rows, nnz, row_split[g], nz_split[g]);
    }
    return 0;
}

int mspmv_mg_local_offsets(const int64_t *h_row_offsets, int64_t rows, int64_t row_begin, int64_t row_end_,
                           int64_t nz_begin, int64_t nz_end_, int32_t *h_local_offsets)
{
    if (!h_row_offsets || !h_local_offsets || row_begin < 0 || row_end_ < row_begin || row_end_ > rows ||
        nz_end_ < nz_begin)
        return kInvalidValue;
    const int64_t local_rows = row_end_ - row_begin + 1;       // + the open row
    if (local_rows + (nz_end_ - nz_begin) > 0x7fffffffLL) return kInvalidValue;
    // local row r (< local_rows-1) is global row row_begin + r: it ENDS at
    // row_offsets[row_begin + r + 1]; its local start is clamped to the part
    // (the first row may have begun on an earlier part).
    h_local_offsets[0] = 0;
    for (int64_t r = 0; r + 1 < local_rows; ++r)
        h_local_offsets[r + 1] = (int32_t) (h_row_offsets[row_begin + r + 1] - nz_begin);
    h_local_offsets[local_rows] = (int32_t) (nz_end_ - nz_begin);   // open row ends with the part
    return 0;
}

}  // extern "C"
