#!/usr/bin/env python3
"""prototype A/B: MSPMV_BAND_PASSES=B python tools/scratch/band_ab.py [case ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G


def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def cases():
    f64, f32 = torch.float64, torch.float32
    for mb in (4, 6, 8, 12, 16, 24, 32):
        for dt, nm in ((f32, "f32"), (f64, "f64")):
            n = mb * 2**20 // (4 if dt == f32 else 8)
            yield f"u{mb}MB_{nm}", (lambda n=n, dt=dt: G.uniform_csr(96_000_000 // 32, n, 32, dtype=dt))
    yield "rmat21_64M_f32", lambda: G.rmat_csr(21, 64_000_000, dtype=f32, seed=G.SEED_C3)
    yield "dense32_f32", lambda: G.dense_csr(3_125_000, 32, dtype=f32, ones=False)


def main():
    only = set(sys.argv[1:])
    print(f"MSPMV_BAND_PASSES={os.environ.get('MSPMV_BAND_PASSES', '(off)')}", flush=True)
    for name, make in cases():
        if only and name not in only: continue
        A = make()
        dt = A.values.dtype; vb = A.values.element_size()
        x = G.uniform_pm1(1, A.cols, dt, "cuda")
        ws = M.CsrMVWorkspace(A.rows, A.nnz, dt); y = torch.empty(A.rows, dtype=dt, device="cuda")
        call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, workspace=ws)
        y.fill_(float("nan")); call(); torch.cuda.synchronize()
        lens = (A.row_offsets[1:] - A.row_offsets[:-1]).long()
        prod = A.values.double() * x.double()[A.column_indices.long()]
        g = torch.segment_reduce(prod, "sum", lengths=lens, unsafe=True); s = torch.segment_reduce(prod.abs(), "sum", lengths=lens, unsafe=True)
        eps = 2.0 ** -24 if vb == 4 else 2.0 ** -53
        tol = 2.0 * (torch.ceil(torch.log2(lens.double() + 1)) + 32) * eps * s
        bad = int(((y.double() - g).abs() > tol).sum())
        del prod, g, s
        ms = timeit(call)
        print(f"{name:18s} rows {A.rows:8d} nnz {A.nnz:10d} x {A.cols * vb / 2**20:6.1f} MiB: {ms:8.4f} ms  {2 * A.nnz / ms / 1e6:7.1f} GFLOP/s  bad={bad}", flush=True)
        del A, x, y, ws


if __name__ == "__main__":
    main()
