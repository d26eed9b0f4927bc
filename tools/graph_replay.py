#!/usr/bin/env python3
"""tools/graph_replay.py -- one CsrMV call (2-4 kernel launches) issued directly vs replayed from a hipGraph:
what a solver loop gains by capturing the call (the C ABI allocates nothing and makes no other runtime call, so it is capturable)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import merge_spmv_amd as M
import sweep
def timeit(fn, iters):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for label, A, x in sweep.workloads(sys.argv[1:] or ["web", "grid2d"]):
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    for _ in range(5): call()
    t_direct = timeit(call, 300)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): call()
    for _ in range(3): g.replay()
    t_graph = timeit(g.replay, 100) / 10
    print(f"{label}: direct {t_direct*1e3:.1f} us per SpMV, replayed from a graph of 10 calls {t_graph*1e3:.1f} us per SpMV", flush=True)
