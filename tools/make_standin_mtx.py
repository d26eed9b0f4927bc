#!/usr/bin/env python3
"""tools/make_standin_mtx.py --dir DIR [--which web,orkut,circuit] -- write the size-matched STAND-INS of the SuiteSparse matrices the
reference's evaluation names (eval_csrmv.sh:8-17 sweeps a directory of .mtx files; README.md:116,137-138 publishes circuit5M;
BASELINE config 3 names webbase-1M and com-Orkut) as Matrix Market files UNDER THE REAL FILES' NAMES, so that

    python bench.py --mtx-dir DIR          (or MSPMV_C3_DIR=DIR)
    tools/eval_csrmv.sh DIR ...

exercise the product's Matrix Market ingest (CooMatrix::InitMarket, sparse_matrix.h:217-380) at corpus scale with the same command
that runs on the real files: drop webbase-1M.mtx / com-Orkut.mtx / circuit5M.mtx from the SuiteSparse collection into DIR and the
very same command reads those instead.  Every stand-in carries the comment line `%STAND-IN written by tools/make_standin_mtx.py ...`
right after its banner; bench.py labels a record `data: "suitesparse ..."` only for a file without it.  (No network here: the real
files cannot be fetched.)  Needs a GPU (the generators run there); writing is parallel on the host (mspmv_host_write_mtx).

    web      webbase-1M.mtx   R-MAT scale 20, 3 105 536 entries (webbase-1M's count), `coordinate pattern general`
    orkut    com-Orkut.mtx    R-MAT scale 22, 117 185 083 stored entries, `coordinate pattern symmetric` (SNAP graphs ship like that)
    circuit  circuit5M.mtx    5 558 326 rows, 59 524 291 entries with a circuit matrix's row-length spread, `coordinate real general`
"""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from merge_spmv_amd import generators as G

MARK = "STAND-IN written by tools/make_standin_mtx.py"          # bench.py: STANDIN_MARK

ap = argparse.ArgumentParser()
ap.add_argument("--dir", required=True)
ap.add_argument("--which", default="web,orkut,circuit")
ap.add_argument("--small", action="store_true", help="1/64 of the sizes (a quick functional check of the same path)")
args = ap.parse_args()
os.makedirs(args.dir, exist_ok=True)
H = ctypes.CDLL(os.path.join(ROOT, "merge_spmv_amd", "libmspmv_host.so"))
H.mspmv_host_write_mtx.restype = ctypes.c_int
H.mspmv_host_write_mtx.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int, ctypes.c_char_p]


def write(name, rows, cols, r, c, v, symmetric, what):
    path = os.path.join(args.dir, name)
    t0 = time.time()
    rh = r.to(torch.int32).cpu().numpy(); ch = c.to(torch.int32).cpu().numpy()
    vh = None if v is None else np.ascontiguousarray(v.double().cpu().numpy())
    st = H.mspmv_host_write_mtx(path.encode(), rows, cols, int(rh.size), rh.ctypes.data, ch.ctypes.data, None if vh is None else vh.ctypes.data,
                                1 if symmetric else 0, f"{MARK}: {what}".encode())
    if st != 0:
        raise SystemExit(f"{path}: write failed ({st})")
    print(f"{path}: {rows} x {cols}, {rh.size} entries{' (symmetric storage)' if symmetric else ''}, {os.path.getsize(path) / 1e6:.0f} MB in {time.time() - t0:.1f} s", flush=True)


div = 64 if args.small else 1
for which in args.which.split(","):
    if which == "web":
        scale = G.C3_WEBBASE_SCALE - (3 if args.small else 0); edges = G.C3_WEBBASE_EDGES // div
        r, c = G.rmat_edges(scale, 0, edges, "cuda", G.SEED_C3)
        write("webbase-1M.mtx", 1 << scale, 1 << scale, r, c, None, False, f"R-MAT scale {scale}, {edges} entries; NOT the SuiteSparse matrix")
    elif which == "orkut":
        scale = G.C3_ORKUT_SCALE - (3 if args.small else 0); edges = G.C3_ORKUT_EDGES // div
        r, c = G.rmat_edges(scale, 0, edges, "cuda", G.SEED_C3)
        write("com-Orkut.mtx", 1 << scale, 1 << scale, r, c, None, True, f"R-MAT scale {scale}, {edges} stored entries mirrored on reading; NOT the SuiteSparse matrix")
    elif which == "circuit":
        A = G.circuit_csr(rows=G.CIRCUIT5M_ROWS // div, nnz=G.CIRCUIT5M_NNZ // div, dtype=torch.float64, device="cuda")
        lens = (A.row_offsets[1:] - A.row_offsets[:-1]).to(torch.int64)
        r = torch.repeat_interleave(torch.arange(A.rows, device="cuda"), lens)
        write("circuit5M.mtx", A.rows, A.cols, r, A.column_indices, A.values, False,
              f"{A.rows} rows, {A.nnz} entries with a circuit matrix's row-length spread (generators.circuit_csr); NOT the SuiteSparse matrix")
        del A, r
    else:
        raise SystemExit(f"unknown stand-in {which}")
    torch.cuda.empty_cache()
