#!/usr/bin/env python3
"""tools/ubench.py -- ceilings on this chip: stream read, random gather vs x
size, and rocSPARSE (via torch.sparse CSR mv) on the bench matrices."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G
M.load_library()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libubench.so"))
vp = ctypes.c_void_p
lib.ub_stream_f4.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, vp]
lib.ub_stream_f1.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, vp]
lib.ub_gather.argtypes = [vp, vp, vp, ctypes.c_size_t, vp, ctypes.c_int, ctypes.c_int, vp]


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


out = torch.zeros(4, device="cuda")
s = vp(torch.cuda.current_stream().cuda_stream)
buf = torch.empty(800_000_000 // 4, dtype=torch.float32, device="cuda").uniform_()
for blocks in (2048, 4096, 8192, 16384):
    t4 = timeit(lambda: lib.ub_stream_f4(vp(buf.data_ptr()), buf.numel() * 4, vp(out.data_ptr()), blocks, s))
    t1 = timeit(lambda: lib.ub_stream_f1(vp(buf.data_ptr()), buf.numel() * 4, vp(out.data_ptr()), blocks, s))
    print(f"stream 800MB blocks {blocks}: float4 {t4:.4f} ms = {0.8/t4:.2f} TB/s | dword {t1:.4f} ms = {0.8/t1:.2f} TB/s", flush=True)
n = 100_000_000
val = buf[:n]
for xbytes in (1 << 20, 3 << 20, 12_500_000, 100_000_000, 1_000_000_000):
    xn = xbytes // 4
    x = torch.empty(xn, dtype=torch.float32, device="cuda").uniform_()
    idx = torch.randint(0, xn, (n,), dtype=torch.int32, device="cuda")
    for unroll in (1, 4, 8):
        for blocks in (4096, 16384):
            t = timeit(lambda: lib.ub_gather(vp(idx.data_ptr()), vp(val.data_ptr()), vp(x.data_ptr()), n, vp(out.data_ptr()), blocks, unroll, s), 10)
            print(f"gather 100M idx+val, x {xbytes/1e6:8.1f} MB, unroll {unroll}, blocks {blocks}: {t:.4f} ms = {n/t/1e6:.1f} Ggather/s", flush=True)
    del x, idx
# rocSPARSE via torch
for name in ("c2", "dense32"):
    A = G.uniform_csr(3_125_000, 3_125_000, 32) if name == "c2" else G.dense_csr(3_125_000, 32, ones=False)
    x = G.uniform_pm1(5, A.cols, torch.float32, "cuda")
    T = torch.sparse_csr_tensor(A.row_offsets, A.column_indices, A.values, size=(A.rows, A.cols))
    t = timeit(lambda: torch.mv(T, x), 10)
    print(f"torch.mv (hipSPARSE/rocSPARSE) {name}: {t:.4f} ms = {2*A.nnz/t/1e6:.1f} GFLOP/s", flush=True)
    ws = M.CsrMVWorkspace(A.rows, A.nnz, torch.float32); y = torch.empty(A.rows, device="cuda")
    t = timeit(lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws), 20)
    print(f"mspmv {name}: {t:.4f} ms = {2*A.nnz/t/1e6:.1f} GFLOP/s", flush=True)
