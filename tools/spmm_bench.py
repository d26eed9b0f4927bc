#!/usr/bin/env python3
"""tools/spmm_bench.py -- SpMM (mspmv_csrmm_*) vs k CsrMV calls on the sweep workloads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import merge_spmv_amd as M
import sweep
def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
for label, A, x in sweep.workloads(sys.argv[1:] or ["c2", "dense32"]):
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    t1 = timeit(lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws), 20)
    print(f"== {label}: CsrMV {t1:.4f} ms = {2*A.nnz/t1/1e6:.1f} GFLOP/s", flush=True)
    tmp = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for k in (1, 2, 4, 8, 16):
        X = torch.empty(A.cols, k, dtype=A.values.dtype, device="cuda").uniform_(-1, 1)
        Y = torch.empty(A.rows, k, dtype=A.values.dtype, device="cuda")
        t = timeit(lambda: M.csrmm(A.values, A.row_offsets, A.column_indices, X, Y=Y, temp=tmp), 10)
        # spot check of column 0 against CsrMV
        y0 = M.csrmv(A.values, A.row_offsets, A.column_indices, X[:, 0].contiguous(), num_cols=A.cols)
        err = float((Y[:, 0].double() - y0.double()).abs().max())
        try:
            import rocsparse_ref
            tr, Cr = rocsparse_ref.time_csrmm(A, X)
            vend = f"rocSPARSE csrmm {tr:.4f} ms ({tr/t:.2f}x ours, max diff {float((Cr.double() - Y.double()).abs().max()):.1e})"
        except Exception as e:
            vend = f"rocSPARSE csrmm unavailable: {e}"
        print(f"   k={k:2d}: SpMM {t:.4f} ms = {2*A.nnz*k/t/1e6:8.1f} GFLOP/s   ({k} CsrMV calls: {k*t1:.4f} ms, {k*t1/t:.2f}x)   {vend}", flush=True)
    del A, x
    torch.cuda.empty_cache()
