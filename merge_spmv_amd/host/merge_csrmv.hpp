// merge_csrmv.hpp -- the PRODUCT's OpenMP merge-path CsrMV (the CPU path timed beside the GPU one,
// BASELINE.md 3).  Follows the algorithm of the reference's OmpMergeCsrmv (cpu_spmv.cpp:292-353):
// the merge path of (row-end offsets) x (nonzero indices) is cut into `segments` equal pieces, one per
// OpenMP thread; a thread stores the rows that end inside its piece and leaves one (row, partial sum)
// carry; a serial pass adds the carries.  It shares no code with the oracle (oracle/merge_oracle.c);
// tests/test_cpu_product_parity.py checks it bit for bit against it for equal segment counts.
//
// Differences from the reference, all deliberate: the carry arrays are sized by the segment count
// (the reference's fixed [256] stack arrays overflow beyond 256 threads, cpu_spmv.cpp:302-303); the
// thread team is `threads`, independent of the segment count, so a result computed with S segments is
// bitwise the same on any machine; the diagonals are 64-bit until clamped.
#pragma once
#include <omp.h>

#include <algorithm>
#include <vector>

namespace mspmv_host {

struct PathPoint { int row, nz; };

// Diagonal search over row END offsets vs the natural numbers
// (cpu_spmv.cpp:223-245): first row whose end offset exceeds diagonal - row - 1.
inline PathPoint SearchDiagonal(int diagonal, const int *row_end, int rows, int nnz)
{
    int lo = std::max(diagonal - nnz, 0), hi = std::min(diagonal, rows);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row_end[mid] <= diagonal - mid - 1) lo = mid + 1; else hi = mid;
    }
    return PathPoint{std::min(lo, rows), diagonal - lo};
}

// the piece of the merge path segment s of `segments` walks: [begin, end)
inline void SegmentBounds(int s, int segments, int rows, int nnz, const int *row_end, PathPoint &begin, PathPoint &end)
{
    const long long total = (long long) rows + nnz;
    const long long per_segment = (total + segments - 1) / segments;     // cpu_spmv.cpp:312
    const int d0 = (int) std::min(per_segment * s, total);               // cpu_spmv.cpp:317-318
    const int d1 = (int) std::min((long long) d0 + per_segment, total);
    begin = SearchDiagonal(d0, row_end, rows, nnz);
    end = SearchDiagonal(d1, row_end, rows, nnz);
}

// y = A*x.  The association order depends on `segments` only; `threads` <= 0 means one thread per segment.
template <typename V>
void MergeCsrmv(int segments, int rows, int nnz, const int *row_end, const int *cols, const V *vals, const V *x, V *y,
                std::vector<int> &carry_row, std::vector<V> &carry_val, int threads = 0)
{
    carry_row.resize(segments); carry_val.resize(segments);
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : segments)
    for (int s = 0; s < segments; ++s) {
        PathPoint p, end;
        SegmentBounds(s, segments, rows, nnz, row_end, p, end);
        for (; p.row < end.row; ++p.row) {           // rows that end inside the segment
            V sum = 0;
            for (; p.nz < row_end[p.row]; ++p.nz) sum += vals[p.nz] * x[cols[p.nz]];
            y[p.row] = sum;
        }
        V sum = 0;                                    // the row left open at the segment end
        for (; p.nz < end.nz; ++p.nz) sum += vals[p.nz] * x[cols[p.nz]];
        carry_row[s] = end.row; carry_val[s] = sum;
    }
    for (int s = 0; s + 1 < segments; ++s)            // cpu_spmv.cpp:348-352
        if (carry_row[s] < rows) y[carry_row[s]] += carry_val[s];
}

}  // namespace mspmv_host
