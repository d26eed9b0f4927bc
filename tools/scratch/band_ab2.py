#!/usr/bin/env python3
"""A/B of the column-band passes through the product API: python tools/scratch/band_ab2.py [case ...]"""
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
    yield "c2_f32", lambda: G.uniform_csr(3_125_000, 3_125_000, 32, dtype=f32)
    yield "c2_f64", lambda: G.uniform_csr(3_125_000, 3_125_000, 32, dtype=f64)
    for mb in (8, 16, 24, 32):
        n = mb * 2**20 // 4
        yield f"u{mb}MB_f32", (lambda n=n: G.uniform_csr(96_000_000 // 32, n, 32, dtype=f32))
    yield "u12MB_f64", lambda: G.uniform_csr(3_000_000, 12 * 2**20 // 8, 32, dtype=f64)
    yield "rmat21_64M_f32", lambda: G.rmat_csr(21, 64_000_000, dtype=f32, seed=G.SEED_C3)
    yield "rmat22_100M_f32", lambda: G.rmat_csr(22, 100_000_000, dtype=f32, seed=G.SEED_C3)
    yield "band_3M_x33_f32", lambda: banded(3_000_000, 33, f32)
    yield "dense32_f32", lambda: G.dense_csr(3_125_000, 32, dtype=f32, ones=False)


def banded(n, w, dt):
    """row i: w consecutive columns around i (clipped): a local matrix with x = 12 MB"""
    rows = torch.arange(n, device="cuda", dtype=torch.int64)
    start = (rows - w // 2).clamp(0, n - w)
    cols = (start[:, None] + torch.arange(w, device="cuda")[None, :]).reshape(-1).to(torch.int32)
    off = (torch.arange(n + 1, device="cuda", dtype=torch.int64) * w).to(torch.int32)
    vals = (torch.rand(n * w, device="cuda", dtype=torch.float64) * 2 - 1).to(dt)
    return G.DeviceCsr(n, n, off, cols, vals)


def main():
    only = set(a for a in sys.argv[1:] if not a.startswith("-"))
    for name, make in cases():
        if only and name not in only: continue
        A = make()
        dt = A.values.dtype; vb = A.values.element_size()
        x = G.uniform_pm1(1, A.cols, dt, "cuda")
        ws = M.CsrMVWorkspace(A.rows, A.nnz, dt); y = torch.empty(A.rows, dtype=dt, device="cuda")
        call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
        lens = (A.row_offsets[1:] - A.row_offsets[:-1]).long()
        prod = A.values.double() * x.double()[A.column_indices.long()]
        g = torch.segment_reduce(prod, "sum", lengths=lens, unsafe=True); s = torch.segment_reduce(prod.abs(), "sum", lengths=lens, unsafe=True)
        del prod
        eps = 2.0 ** -24 if vb == 4 else 2.0 ** -53
        tol = 2.0 * (torch.ceil(torch.log2(lens.double() + 1)) + 32) * eps * s
        out = []
        for label, passes in (("never", -1), ("auto", 0), ("force2", 2), ("force3", 3), ("force4", 4)):
            M.set_band_passes(vb, passes)
            y.fill_(float("nan")); call(); torch.cuda.synchronize()
            bad = int(((y.double() - g).abs() > tol).sum()) + int(torch.isnan(y).sum())
            y2 = y.clone(); call(); torch.cuda.synchronize()
            rep = bool(torch.equal(y, y2))
            ms = timeit(call)
            extra = ""
            if passes == 0:
                try: extra = f" windows={int(M.debug_band_windows(ws, A.rows, A.nnz, vb).sum())}/64"
                except Exception as e: extra = f" ({e})"
                M.profile_begin(20)
                for _ in range(20): call()
                torch.cuda.synchronize(); pr = M.profile_end()
                extra += f" [search {pr['search_ms']*1e3:.1f} tile {pr['tile_ms']*1e3:.1f} fix {pr['fixup_ms']*1e3:.1f} us]"
            out.append(f"{label} {ms:7.4f}{'' if bad == 0 else ' BAD=' + str(bad)}{'' if rep else ' NONREPRO'}{extra}")
        M.set_band_passes(vb, 0)
        print(f"{name:18s} x {A.cols * vb / 2**20:5.1f} MiB nnz {A.nnz/1e6:5.1f}M: " + "  ".join(out), flush=True)
        del A, x, y, ws, g, s


if __name__ == "__main__":
    main()
