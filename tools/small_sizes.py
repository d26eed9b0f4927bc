#!/usr/bin/env python3
"""tools/small_sizes.py -- latency of one SpMV call on small matrices (the regime the second dependent
launch dominates, paper p.10) vs rocSPARSE csrmv after analysis."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G
import rocsparse_ref
def t(fn, iters=300):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
for w in (30, 100, 200, 300, 400, 500, 600, 700, 1200):
    A = G.grid2d_csr(w, torch.float64)
    x = torch.ones(A.cols, dtype=torch.float64, device="cuda")
    ws = M.CsrMVWorkspace(A.rows, A.nnz, torch.float64); y = torch.empty(A.rows, dtype=torch.float64, device="cuda")
    us = t(lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, workspace=ws))
    M.set_tuning(8, 0, 0, 0x40000000)          # the same kernel followed by the separate fix-up launch
    us2 = t(lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, workspace=ws))
    M.set_tuning(8)
    info = M.launch_info(A.rows, A.nnz, 8)
    _, avg, _ = rocsparse_ref.time_csrmv(A, x, iters=300)
    print(f"grid2d_{w}: rows {A.rows:8d} nnz {A.nnz:9d} tiles {info['num_tiles']:5d} ({info['block_threads']}x{info['items_per_thread']}): ours {us:7.1f} us (two launches {us2:7.1f} us)   rocSPARSE {avg*1e3:7.1f} us", flush=True)
