#!/usr/bin/env python3
"""tools/first_call.py -- what the verified coordinate hints of tile_kernel_snap cost and save: per-call time of the stateless
CsrMV (a) on temp storage whose hints are garbage EVERY call (the buffer is refilled with 0xFF between calls: every tile
searches its boundaries), (b) in steady state (hints left by the previous call), (c) with the classic three launches
(MSPMV_TUNE_TWO_LAUNCH), plus rocSPARSE csrmv after analysis.  hipEvents around single calls, median of 30."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import merge_spmv_amd as M
import sweep


def one_call_ms(fn, before=None, n=30):
    ts = []
    for _ in range(n):
        if before is not None:
            before()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


names = sys.argv[1:] or ["dense5d", "grid2d", "grid3d", "band", "web", "rmat", "c4", "dense32"]
print("# ms per single call (hipEvents; median of 30): hints garbage every call | hints from the previous call | classic three launches | rocSPARSE")
for label, A, x in sweep.workloads(names):
    vb = A.values.element_size()
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype)
    y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    call(); torch.cuda.synchronize()
    cold = one_call_ms(call, before=lambda: ws.buffer.fill_(255))
    call()
    warm = one_call_ms(call)
    M.set_tuning(vb, 0, 0, 0x40000000)
    classic = one_call_ms(call)
    M.set_tuning(vb)
    roc = float("nan")
    try:
        import rocsparse_ref
        _, roc, _ = rocsparse_ref.time_csrmv(A, x)
    except Exception:
        pass
    print(f"{label:28s} {cold:8.4f} | {warm:8.4f} | {classic:8.4f} | {roc:8.4f}", flush=True)
    del A, x, ws, y
    torch.cuda.empty_cache()
