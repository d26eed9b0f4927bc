#!/usr/bin/env python3
"""tools/skew_probe.py [workload ...] -- what mspmv_csrmv_hotcols_skew says about the sweep's matrices (and `c5`: BASELINE config 5 itself, `rmat18`:
the test's small scale-free matrix): the figure the multi-GPU plan's automatic hot-column decision rests on (150 <= x < 800, >= 256 wide windows)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G
import sweep
lib = M.load_library()


def probe(label, A):
    d, w = ctypes.c_int32(), ctypes.c_int32()
    vb = A.values.element_size()
    st = lib.mspmv_csrmv_hotcols_skew(ctypes.c_void_p(A.column_indices.data_ptr()), A.cols, A.nnz, vb, None, ctypes.byref(d), ctypes.byref(w))
    x_mb = A.cols * vb / 2**20
    want = 150 <= d.value < 800 and w.value >= 256
    print(f"{label:28s} x {x_mb:8.1f} MiB  distinct lines {d.value:5d} per mille of a uniform draw, wide windows {w.value:3d} / 512  -> "
          f"{'hot-column plan' if want else 'no plan'}{'' if x_mb > 256 else ' (x fits the Infinity Cache: no plan either way)'}  [status {st}]", flush=True)


for n in sys.argv[1:] or ["c2", "dense32", "band", "grid2d", "rmat", "orkut", "circuit", "web", "rmat24", "rmat18", "c5"]:
    if n == "c5":
        probe("c5_rmat26_f64", G.rmat_csr(26, 2_000_000_000, dtype=torch.float64, device="cuda", seed=G.SEED_C5))
    elif n == "rmat18":
        probe("rmat18_6M_f64", G.rmat_csr(18, 6_000_000, dtype=torch.float64, device="cuda", seed=G.SEED_C5))
    else:
        for label, A, x in sweep.workloads([n]):
            probe(label, A)
    torch.cuda.empty_cache()
