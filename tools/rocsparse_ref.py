"""tools/rocsparse_ref.py -- time rocSPARSE csrmv (with analysis) on torch CSR
tensors through ctypes, as the vendor-library reference point the reference's
driver had in cuSPARSE (gpu_spmv.cu:262-364).  Development / reporting aid."""
import ctypes, os
import torch

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librocsparse.so")
        _lib = ctypes.CDLL(path if os.path.exists(path) else "librocsparse.so")
    return _lib


def time_csrmv(A, x, iters=30):
    """returns (analysis_ms, avg_ms, y)"""
    L = lib(); vp = ctypes.c_void_p
    handle, descr, info = vp(), vp(), vp()
    assert L.rocsparse_create_handle(ctypes.byref(handle)) == 0
    assert L.rocsparse_set_stream(handle, vp(torch.cuda.current_stream().cuda_stream)) == 0
    assert L.rocsparse_create_mat_descr(ctypes.byref(descr)) == 0
    assert L.rocsparse_create_mat_info(ctypes.byref(info)) == 0
    f32 = A.values.dtype == torch.float32
    ana = L.rocsparse_scsrmv_analysis if f32 else L.rocsparse_dcsrmv_analysis
    mv = L.rocsparse_scsrmv if f32 else L.rocsparse_dcsrmv
    ct = ctypes.c_float if f32 else ctypes.c_double
    y = torch.empty(A.rows, dtype=A.values.dtype, device=x.device)
    alpha, beta = ct(1.0), ct(0.0)
    i32 = ctypes.c_int
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    st = ana(handle, i32(111), i32(A.rows), i32(A.cols), i32(A.nnz), descr, vp(A.values.data_ptr()),
             vp(A.row_offsets.data_ptr()), vp(A.column_indices.data_ptr()), info)
    e1.record(); torch.cuda.synchronize()
    assert st == 0, st
    analysis_ms = e0.elapsed_time(e1)

    def call():
        st = mv(handle, i32(111), i32(A.rows), i32(A.cols), i32(A.nnz), ctypes.byref(alpha), descr, vp(A.values.data_ptr()),
                vp(A.row_offsets.data_ptr()), vp(A.column_indices.data_ptr()), info, vp(x.data_ptr()), ctypes.byref(beta),
                vp(y.data_ptr()))
        assert st == 0, st
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    avg = e0.elapsed_time(e1) / iters
    L.rocsparse_destroy_mat_info(info); L.rocsparse_destroy_mat_descr(descr); L.rocsparse_destroy_handle(handle)
    return analysis_ms, avg, y


def time_csrmm(A, X, iters=10):
    """rocSPARSE csrmm (legacy entry point) on a row-major X [cols, k] (passed as op(B) = B^T of the k x cols
    column-major matrix with the same memory); C is written column-major [rows, k].  returns (avg_ms, C as [rows, k])."""
    L = lib(); vp = ctypes.c_void_p; i32 = ctypes.c_int
    handle, descr = vp(), vp()
    assert L.rocsparse_create_handle(ctypes.byref(handle)) == 0
    assert L.rocsparse_set_stream(handle, vp(torch.cuda.current_stream().cuda_stream)) == 0
    assert L.rocsparse_create_mat_descr(ctypes.byref(descr)) == 0
    f32 = A.values.dtype == torch.float32
    mm = L.rocsparse_scsrmm if f32 else L.rocsparse_dcsrmm
    ct = ctypes.c_float if f32 else ctypes.c_double
    k = X.shape[1]
    C = torch.empty(k, A.rows, dtype=A.values.dtype, device=X.device)         # column-major rows x k
    alpha, beta = ct(1.0), ct(0.0)

    def call():
        st = mm(handle, i32(111), i32(112), i32(A.rows), i32(k), i32(A.cols), i32(A.nnz), ctypes.byref(alpha), descr,
                vp(A.values.data_ptr()), vp(A.row_offsets.data_ptr()), vp(A.column_indices.data_ptr()), vp(X.data_ptr()), i32(k),
                ctypes.byref(beta), vp(C.data_ptr()), i32(A.rows))
        assert st == 0, st
    for _ in range(2): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    avg = e0.elapsed_time(e1) / iters
    L.rocsparse_destroy_mat_descr(descr); L.rocsparse_destroy_handle(handle)
    return avg, C.t()
