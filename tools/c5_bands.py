#!/usr/bin/env python3
"""tools/c5_bands.py [scale edges] -- BASELINE config 5 on one GPU (fp64 R-MAT scale 26, 2e9 edges, x = 512 MB) through the stateless call
with the column-band passes forced to 2 ... 8 (the device-side detector refuses R-MAT; this asks what bands the size of the Infinity
Cache rather than of an L2 would buy): ms per call, and the result against the never-banded one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
edges = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000_000
dev = torch.device("cuda", 0)
A = G.rmat_csr(scale, edges, dtype=torch.float64, device=dev, seed=G.SEED_C5)
x = G.uniform_pm1(G.SEED_C5 + 2, A.cols, torch.float64, dev)
ws = M.CsrMVWorkspace(A.rows, A.nnz, torch.float64, device=dev)
y = torch.empty(A.rows, dtype=torch.float64, device=dev)
call = lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=y, num_cols=A.cols, workspace=ws)
print(f"R-MAT scale {scale}: rows {A.rows} nnz {A.nnz}, x {A.cols * 8 / 2**20:.0f} MB", flush=True)
ref = None
for passes in [int(p) for p in os.environ.get("C5_BANDS", "-1,2,3,4,6,8").split(",")]:
    M.set_band_passes(8, passes)
    for _ in range(2): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): call()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    if ref is None: ref = y.clone()
    err = float((y - ref).abs().max())
    print(f"  band passes {passes:3d}: {ms:8.3f} ms  {2.0 * A.nnz / ms / 1e6:7.1f} GFLOP/s   max |y - y_never| = {err:.3g}", flush=True)
M.set_band_passes(8, 0)
