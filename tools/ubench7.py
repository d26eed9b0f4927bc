import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import merge_spmv_amd as M
from merge_spmv_amd import generators as G
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libubench7.so"))
vp = ctypes.c_void_p
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
A = G.uniform_csr(3_125_000, 3_125_000, 32, dtype=torch.float32)
x = G.uniform_pm1(1, A.cols, torch.float32, "cuda")
s = vp(torch.cuda.current_stream().cuda_stream)
yref = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
t_ref = timeit(lambda: M.csrmv(A.values, A.row_offsets, A.column_indices, x, y=yref, num_cols=A.cols))
print(f"merge CsrMV: {t_ref:.4f} ms")
for nb in (1, 2, 4, 8):
    bw = (A.cols + nb - 1) // nb
    splits = torch.empty((nb + 1) * A.rows, dtype=torch.int32, device="cuda")
    ypart = torch.empty(nb * A.rows, dtype=torch.float32, device="cuda")
    y = torch.empty(A.rows, dtype=torch.float32, device="cuda")
    f_s = lambda: lib.ub7_splits(vp(A.row_offsets.data_ptr()), vp(A.column_indices.data_ptr()), vp(splits.data_ptr()), A.rows, nb, bw, s)
    t_s = timeit(f_s)
    for rpb in (256, 1024, 4096):
        f_b = lambda: lib.ub7_banded(vp(A.values.data_ptr()), vp(A.column_indices.data_ptr()), vp(splits.data_ptr()), vp(x.data_ptr()), vp(ypart.data_ptr()), vp(y.data_ptr()), A.rows, nb, rpb, s)
        t_b = timeit(f_b)
        err = float((y.double() - yref.double()).abs().max())
        print(f"bands {nb}: splits {t_s:.4f} ms, banded pass + combine ({rpb} rows/block) {t_b:.4f} ms, total {t_s + t_b:.4f} ms  (max diff vs merge {err:.2e})", flush=True)
