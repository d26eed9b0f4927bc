#!/usr/bin/env python3
"""tools/small_shapes.py [width ...] -- per-call time of small fp64 5-point grids for every compiled tile shape, free of host
launch overhead: 50 calls captured into one hipGraph, replayed 20 times (the calls are dependent through y, as in a solver)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G

f32 = os.environ.get("SMALL_F32") == "1"
npdt, tdt, vb = (np.float32, torch.float32, 4) if f32 else (np.float64, torch.float64, 8)
widths = [int(a) for a in sys.argv[1:]] or [100, 300, 500, 600, 700, 800, 1000, 1400]
for w in widths:
    A = G.grid2d_csr(w, tdt)
    x = torch.ones(A.cols, dtype=tdt, device="cuda"); y = torch.empty(A.rows, dtype=tdt, device="cuda")
    line = f"grid2d_{w} {'fp32' if f32 else 'fp64'} ({A.nnz} nnz):"
    dev = "exp" in os.environ.get("MSPMV_LIB", "")        # MSPMV_LIB=merge_spmv_amd/libmspmv_exp.so: every sweep shape
    for shape in [(0, 0)] + (([(256, 7), (256, 9), (256, 11), (256, 15)] if f32 else [(256, 5), (256, 7), (256, 9), (256, 11)]) if dev else [(256, 7), (256, 11)]):
        M.set_tuning(vb, shape[0], shape[1], 0)
        info = M.launch_info(A.rows, A.nnz, vb)
        ws = M.CsrMVWorkspace(A.rows, A.nnz, tdt)
        call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, workspace=ws)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): call()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(50): call()
            g.replay(); s.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): g.replay()
            s.synchronize()
        us = (time.perf_counter() - t0) / 1000 * 1e6
        line += f"  {'default ' if shape[0] == 0 else ''}{info['block_threads']}x{info['items_per_thread']} ({info['num_tiles']} tiles) {us:5.1f} us |"
    M.set_tuning(vb)
    print(line, flush=True)
