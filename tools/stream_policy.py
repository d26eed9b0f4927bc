#!/usr/bin/env python3
"""tools/stream_policy.py [workload ...] -- non-temporal vs ordinary loads for the CSR streams of the one-launch kernel
(MSPMV_TUNE_FORCE_NT = 32 / MSPMV_TUNE_FORCE_TEMPORAL = 64), interleaved 5 x N calls each, min and median per setting:
what the dispatcher's size threshold (csrc/mspmv_api.hip: 256 MB of stream bytes) is read from."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import merge_spmv_amd as M
import sweep


def timeit(call, iters):
    for _ in range(5): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


extra = int(os.environ.get("POLICY_EXTRA_FLAGS", "0"), 0)
for label, A, x in sweep.workloads(sys.argv[1:] or ["dense32", "dense32d", "band", "grid2d4096", "grid3d", "c4", "rmat"]):
    vb = A.values.element_size()
    ws = M.CsrMVWorkspace(A.rows, A.nnz, A.values.dtype); y = torch.empty(A.rows, dtype=A.values.dtype, device="cuda")
    call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
    stream_mb = (A.nnz * (vb + 4) + 4 * A.rows) / 2**20
    iters = max(20, min(300, int(30 / max(stream_mb / 5000, 0.02))))
    res = {32: [], 64: []}
    for _ in range(5):
        for fl in (32, 64):
            M.set_tuning(vb, 0, 0, fl | extra)
            res[fl].append(timeit(call, iters))
    M.set_tuning(vb)
    nt, tp = sorted(res[32]), sorted(res[64])
    print(f"{label:24s} stream {stream_mb:7.0f} MiB x {A.cols * vb / 2**20:6.1f} MiB: non-temporal min {nt[0]:.4f} med {nt[2]:.4f}   ordinary min {tp[0]:.4f} med {tp[2]:.4f}   "
          f"ordinary/non-temporal {tp[2] / nt[2]:.3f}", flush=True)
    del A, x, ws, y
    torch.cuda.empty_cache()
