#!/usr/bin/env python3
"""tools/long_rows_probe.py -- rows of a few hundred to a few thousand nonzeros (longer than the 192 a tile boundary can snap over, shorter than
a few tiles): the one-launch kernel's tagged records against the classic three launches (development library: MSPMV_TUNE_TWO_LAUNCH), and
rocSPARSE's csrmv.  ms per SpMV, steady state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G
import rocsparse_ref


def timed(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def run(name, A, x):
    vb = A.values.element_size()
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    n = max(20, min(2000, int(4e9 / max(A.nnz, 1))))
    out = []
    for label, flags in (("one launch", 0), ("classic 3 launches", 0x40000000)):
        M.set_tuning(vb, 0, 0, flags) if flags else M.use_library("product")
        ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
        out.append(timed(lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws), n))
        M.set_tuning(vb)
    M.use_library("product")
    try:
        _, ven, _ = rocsparse_ref.time_csrmv(A, x, iters=min(n, 50))
    except Exception as e:  # noqa: BLE001
        ven = float("nan")
    b_alg = A.nnz * (vb + 4) + (A.rows + 1) * 4 + A.rows * vb + A.cols * vb
    print(f"{name:34s} fp{vb * 8} rows {A.rows:9d} nnz {A.nnz:10d}: one launch {out[0]:.4f} ms ({b_alg / out[0] / 1e6 / 8000:.3f} of 8 TB/s) | classic {out[1]:.4f} | rocSPARSE {ven:.4f}", flush=True)


for dt in (torch.float64, torch.float32):
    for cols, nnz in ((512, 3_000_000), (512, 30_000_000), (512, 100_000_000), (256, 30_000_000), (1024, 30_000_000), (4096, 30_000_000)):
        A = G.dense_csr(nnz // cols, cols, dtype=dt, device="cuda", ones=False)
        run(f"dense rows x {cols}", A, G.uniform_pm1(7, A.cols, dt, "cuda"))
        del A
    for k, nnz in ((300, 30_000_000), (512, 30_000_000), (1000, 30_000_000), (3000, 30_000_000)):
        rows = nnz // k
        A = G.uniform_csr(rows, max(rows, 100_000), k, dtype=dt, device="cuda")
        run(f"uniform {k} per row, x {A.cols} entries", A, G.uniform_pm1(7, A.cols, dt, "cuda"))
        del A
