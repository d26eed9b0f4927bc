#!/usr/bin/env python3
"""tools/hot_columns.py [scale edges] -- does renumbering the columns by how often they are referenced help a matrix whose x
is far larger than the caches (BASELINE config 5: R-MAT scale 26, x = 512 MB)?  The stateless CsrMV is timed on the matrix
as generated and on the same matrix with its columns relabelled in order of descending reference count (x permuted to
match; the result y is the same vector).  Hot columns then share cache lines instead of being spread over the whole of x:
R-MAT's 314 000 hottest columns (56 % of the references at scale 26) shrink from 110 000 lines (14 MB) to 2.5 MB.
Experiment behind DESIGN.md's section on config 5; output kept under profiles/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
edges = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000_000
seed = G.SEED_C5


def timeit(A, x, iters=10):
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    for _ in range(2):
        M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, y


t0 = time.time()
A = G.rmat_csr(scale, edges, dtype=torch.float64, seed=seed)
x = G.uniform_pm1(seed + 2, A.cols, torch.float64, "cuda")
print(f"# R-MAT scale {scale}, {A.nnz} nonzeros, x = {A.cols * 8 / 1e6:.0f} MB; generated in {time.time() - t0:.1f} s", flush=True)
ms0, y0 = timeit(A, x)
print(f"as generated            : {ms0:8.3f} ms = {2 * A.nnz / ms0 / 1e6:7.1f} GFLOP/s = {A.nnz / ms0 / 1e6:6.1f} G gathers/s", flush=True)

t0 = time.time()
chunk = 1 << 28
counts = torch.zeros(A.cols, dtype=torch.int64, device="cuda")
for a in range(0, A.nnz, chunk):
    counts += torch.bincount(A.column_indices[a:a + chunk].long(), minlength=A.cols)
order = torch.argsort(counts, descending=True, stable=True)          # order[k] = the k-th most referenced column
rank = torch.empty(A.cols, dtype=torch.int32, device="cuda")
rank[order] = torch.arange(A.cols, dtype=torch.int32, device="cuda")
for a in range(0, A.nnz, chunk):
    A.column_indices[a:a + chunk] = rank[A.column_indices[a:a + chunk].long()]
torch.cuda.synchronize()
setup = time.time() - t0
tot = float(counts.sum())
csum = torch.cumsum(counts[order].double(), 0) / tot
for k in (1 << 16, 1 << 19, 1 << 22, 1 << 25):
    if k < A.cols:
        print(f"#   the {k:>9d} most referenced columns ({k * 8 / 1e6:7.1f} MB of x) take {float(csum[k - 1]) * 100:5.1f} % of the references")
tp0 = time.perf_counter()
xp = x[order]                                                       # what a caller with x in the original order pays per SpMV
torch.cuda.synchronize()
perm_ms = (time.perf_counter() - tp0) * 1e3
ms1, y1 = timeit(A, xp)
print(f"columns by hotness      : {ms1:8.3f} ms = {2 * A.nnz / ms1 / 1e6:7.1f} GFLOP/s = {A.nnz / ms1 / 1e6:6.1f} G gathers/s   "
      f"(relabelling set-up with torch ops {setup:.1f} s; permuting x {perm_ms:.2f} ms per SpMV if the caller keeps x in the original order)", flush=True)
err = float((y1 - y0).abs().max()); ref = float(y0.abs().max())
print(f"max |y_hot - y_orig| = {err:.3g} (max |y| = {ref:.3g}): the same sums in another order")
