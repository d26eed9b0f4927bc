#!/usr/bin/env python3
"""cost of the column-band machinery on matrices it does not help: never vs auto, interleaved"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G
from band_ab2 import banded, timeit

f32, f64 = torch.float32, torch.float64
cases = [("band_3M_x33_f32", lambda: banded(3_000_000, 33, f32)),
         ("band_2M_x40_f64", lambda: banded(2_000_000, 40, f64)),
         ("rmat21_64M_f32", lambda: G.rmat_csr(21, 64_000_000, dtype=f32, seed=G.SEED_C3)),
         ("rmat22_100M_f32", lambda: G.rmat_csr(22, 100_000_000, dtype=f32, seed=G.SEED_C3))]
for name, make in cases:
    A = make(); dt = A.values.dtype; vb = A.values.element_size()
    x = G.uniform_pm1(1, A.cols, dt, "cuda")
    ws = M.CsrMVWorkspace(A.rows, A.nnz, dt); y = torch.empty(A.rows, dtype=dt, device="cuda")
    call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    res = {-1: [], 0: []}
    prof = {}
    for rep in range(5):
        for mode in (-1, 0):
            M.set_band_passes(vb, mode)
            res[mode].append(timeit(call, 300))
    for mode in (-1, 0):
        M.set_band_passes(vb, mode)
        M.profile_begin(100)
        for _ in range(100): call()
        torch.cuda.synchronize(); prof[mode] = M.profile_end()
    M.set_band_passes(vb, 0)
    fmt = lambda pr: f"[search {pr['search_ms']*1e3:.1f} tile {pr['tile_ms']*1e3:.1f} fix {pr['fixup_ms']*1e3:.1f}]"
    print(f"{name:18s} never {min(res[-1])*1e3:7.1f} us (median {sorted(res[-1])[2]*1e3:7.1f}) {fmt(prof[-1])}   auto {min(res[0])*1e3:7.1f} us (median {sorted(res[0])[2]*1e3:7.1f}) {fmt(prof[0])}", flush=True)
    del A, x, y, ws
