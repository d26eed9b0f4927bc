#!/usr/bin/env python3
"""tools/compact_other_matrices.py -- the compact launches (csrc/mspmv_kernels.hpp: compact_front, one contiguous tile range per XCD)
on matrices that are NOT 5-point grids: small R-MAT graphs, circuit-shaped matrices, rows of 32 uniform columns, 7-point 3-D grids,
--dense=5 -- most of whose tiles do not qualify for the fast lane and run the general body behind it.  Per matrix and precision: µs per
call, back-to-back calls captured in one graph and replayed (the reference's loop, gpu_spmv.cu:418-434, without the host side), compact launches on (the library default)
against off (mspmv_set_compact_tiles(-1): the general kernel), loops interleaved, median of 7.  What it is for: the tile limits of
the compact launches (mspmv_api.hip: compact_max_tiles) were set on grids; this shows what they do elsewhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G

CALLS = int(os.environ.get('COMPACT_OTHER_CALLS', '500'))


def captured(fn):
    """CALLS back-to-back calls captured in one graph (the Python binding's own ~10 us per call would hide everything otherwise)"""
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(CALLS): fn()
    torch.cuda.synchronize()
    return g


def loop(g):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1000.0 / CALLS


def cases(dt):
    if os.environ.get("COMPACT_OTHER_FULL"):            # fp64 workloads of the sweep, 256x7 behind the compact front end against the default (256x11)
        if dt == torch.float64:
            yield "dense5 fp64 (16.8M)", lambda: G.dense_csr((1 << 24) // 5, 5, dtype=dt, ones=False)
            yield "grid2d 2000 (16M)", lambda: G.grid2d_csr(2000, dtype=dt)
            yield "grid3d 200 (47.8M)", lambda: G.grid3d_csr(200, dtype=dt)
        return
    if os.environ.get("COMPACT_OTHER_BIG"):             # the same families at 3400-4400 tiles of the small shape
        yield "rmat20 5M", lambda: G.rmat_csr(20, 5_000_000, dtype=dt)
        yield "circuit 600k rows 6.4M", lambda: G.circuit_csr(rows=600_000, nnz=6_400_000, dtype=dt)
        yield "uniform32 200k rows 6.4M", lambda: G.uniform_csr(200_000, 200_000, 32, dtype=dt)
        yield "grid3d 100 (6.9M)", lambda: G.grid3d_csr(100, dtype=dt)
        yield "dense5 1.2M rows (6.0M)", lambda: G.dense_csr(1_200_000, 5, dtype=dt)
        yield "grid2d 1200 (5.8M)", lambda: G.grid2d_csr(1200, dtype=dt)
        return
    yield "rmat18 1.5M", lambda: G.rmat_csr(18, 1_500_000, dtype=dt)
    yield "rmat19 2.5M", lambda: G.rmat_csr(19, 2_500_000, dtype=dt)
    yield "circuit 300k rows 3.2M", lambda: G.circuit_csr(rows=300_000, nnz=3_200_000, dtype=dt)
    yield "uniform32 60k rows 1.9M", lambda: G.uniform_csr(60_000, 60_000, 32, dtype=dt)
    yield "grid3d 70 (2.4M)", lambda: G.grid3d_csr(70, dtype=dt)
    yield "dense5 400k rows (2.0M)", lambda: G.dense_csr(400_000, 5, dtype=dt)
    yield "grid2d 600 (1.4M)", lambda: G.grid2d_csr(600, dtype=dt)


for dt in (torch.float64, torch.float32):
    vb = 8 if dt == torch.float64 else 4
    print(f"# {'fp64' if vb == 8 else 'fp32'}: us per call, median of 7 interleaved loops of {CALLS} calls; tiles = tiles of the call's shape")
    for name, make in cases(dt):
        A = make()
        x = G.uniform_pm1(7, A.cols, dt, "cuda")
        y = torch.empty(A.rows, dtype=dt, device="cuda")
        if os.environ.get("COMPACT_OTHER_FULL"): M.set_tuning(vb, 256, 7)          # (the larger temp storage of the two shapes)
        ws = M.CsrMVWorkspace(A.rows, A.nnz, dt)
        M.set_tuning(vb)
        info = M.launch_info(A.rows, A.nnz, vb)
        call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
        res = {}
        force7 = bool(os.environ.get("COMPACT_OTHER_FULL"))
        def setmode(mode):
            M.set_compact_tiles(mode)
            if force7: M.set_tuning(vb, 256, 7) if mode == 0 else M.set_tuning(vb)
        for mode in (0, -1):
            setmode(mode); call(); torch.cuda.synchronize(); res[mode] = [y.clone()]
        t = {0: [], -1: []}; graphs = {}
        for mode in (0, -1):
            setmode(mode); graphs[mode] = captured(call)
        setmode(-1); M.set_compact_tiles(0)
        for _ in range(7):
            for mode in (0, -1): loop(graphs[mode]); t[mode].append(loop(graphs[mode]))
        same = bool(torch.equal(res[0][0], res[-1][0]))
        print(f"{name:28s} nnz {A.nnz:9d} tiles {info['num_tiles']:5d} (256x{info['items_per_thread']}): compact launches {np.median(t[0]):6.2f}   general kernel {np.median(t[-1]):6.2f}   bitwise equal: {same}", flush=True)
        del graphs, A, x, y, ws
