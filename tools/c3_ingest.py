#!/usr/bin/env python3
"""tools/c3_ingest.py [edges] [scale] -- BASELINE config 3 through the REAL ingest path at corpus scale (run on the GPU box).

Writes a com-Orkut-sized graph (R-MAT scale 22, 117 185 083 stored entries, the SNAP file's count) as a
`coordinate pattern symmetric` Matrix Market file -- the format SuiteSparse ships such graphs in -- then runs
`gpu_spmv --mtx=<file> --cache --timing` twice: the first run reads, parses (all host threads), mirrors the
off-diagonal entries (InitMarket, sparse_matrix.h:362-368), converts COO -> CSR and leaves the binary CSR image; the
second is served from the image.  Both print the ingest phases; the CSR the driver built is compared, array for
array, with the one the generator gives for the symmetrised graph."""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merge_spmv_amd import generators as G

edges = int(sys.argv[1]) if len(sys.argv) > 1 else 117_185_083
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 22
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = ctypes.CDLL(os.path.join(root, "merge_spmv_amd", "libmspmv_host.so"))
H.mspmv_host_write_pattern_mtx.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
n = 1 << scale
t0 = time.time()
r, c = G.rmat_edges(scale, 0, edges, "cuda", G.SEED_C3)
# expected CSR of the symmetrised pattern matrix: (r,c) plus (c,r) for r != c, sorted by (row, col), duplicates kept
off_diag = r != c
rr = torch.cat([r, c[off_diag]]); cc = torch.cat([c, r[off_diag]])
order = torch.sort(rr * n + cc, stable=True).indices
exp_cols = cc[order].to(torch.int32).cpu().numpy()
exp_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); torch.cumsum(torch.bincount(rr, minlength=n), 0, out=exp_off[1:])
exp_off = exp_off.to(torch.int32).cpu().numpy()
rh = r.to(torch.int32).cpu().numpy(); ch = c.to(torch.int32).cpu().numpy()
del r, c, rr, cc, order
path = "/tmp/c3_orkut_like.mtx"
t1 = time.time()
assert H.mspmv_host_write_pattern_mtx(path.encode(), n, n, edges, rh.ctypes.data, ch.ctypes.data, 1) == 0
print(f"# generated {edges} edges in {t1 - t0:.1f} s, wrote {os.path.getsize(path) / 1e9:.2f} GB Matrix Market file in {time.time() - t1:.1f} s; "
      f"symmetrised nnz = {exp_cols.size}", flush=True)
for f in (path + ".fp64.csrbin",):
    if os.path.exists(f): os.remove(f)
exe = os.path.join(root, "merge_spmv_amd", "gpu_spmv")
for run in ("first run (parse + sort + save image)", "second run (binary CSR image)"):
    t0 = time.time()
    out = subprocess.run([exe, f"--mtx={path}", "--cache", "--timing", "--i=50", "--no-vendor"], capture_output=True, text=True, timeout=1800)
    print(f"## {run}: {time.time() - t0:.1f} s wall, exit {out.returncode}")
    keep = [l for l in out.stdout.splitlines() if any(k in l for k in ("ingest seconds", "num_rows", "num_nonzeros", "PASS", "FAIL", "fp64:", "Reading", "compulsory"))]
    print("\n".join(keep), flush=True)
    if out.returncode != 0: print(out.stderr[-2000:])
# the image the driver left IS its CSR: compare with the generator's
img = path + ".fp64.csrbin"
with open(img, "rb") as f:
    head = f.read(28)
    rows, cols, nnz = np.frombuffer(head[16:28], np.int32)
    off = np.fromfile(f, np.int32, rows + 1); col = np.fromfile(f, np.int32, nnz)
assert (rows, cols, nnz) == (n, n, exp_cols.size), (rows, cols, nnz)
assert np.array_equal(off, exp_off) and np.array_equal(col, exp_cols)
print(f"# CSR built by the driver == the generator's symmetrised CSR ({rows} rows, {nnz} nonzeros): row_offsets and column_indices identical")
