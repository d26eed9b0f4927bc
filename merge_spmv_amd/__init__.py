"""merge_spmv_amd -- MI355X-native merge-based CsrMV.

Python face of the C ABI in include/mspmv.h (libmspmv.so: hand-written HIP
kernels for gfx950).  It mirrors the reference's device API
``cub::DeviceSpmv::CsrMV`` (reference cub/device/device_spmv.cuh:129-164) --
same argument order, same two-phase temp-storage convention -- so parity
tests read like the reference's own call sites (gpu_spmv.cu:390-409).

PyTorch is used only as plumbing: device memory (tensors), streams and, for
the multi-GPU path, torch.distributed.  There is NO CPU or eager fallback: if
libmspmv.so is missing or fails to load, importing the compute entry points
raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

__all__ = ["DeviceSpmv", "csrmv", "csrmm", "CsrMVWorkspace", "library_path", "load_library", "launch_info",
           "set_tuning", "debug_read_tiles", "profile_begin", "profile_end", "MspmvError",
           "TUNE_XCD_REMAP", "TUNE_ATOMIC_FIX", "TUNE_NO_VEC"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libmspmv.so"
_lib: Optional[ctypes.CDLL] = None

TUNE_XCD_REMAP = 1
TUNE_ATOMIC_FIX = 2
TUNE_NO_VEC = 4


class MspmvError(RuntimeError):
    pass


def library_path() -> str:
    return os.path.join(_HERE, _LIB_NAME)


class _LaunchInfo(ctypes.Structure):
    _fields_ = [("block_threads", ctypes.c_int32), ("items_per_thread", ctypes.c_int32),
                ("tile_items", ctypes.c_int32), ("num_tiles", ctypes.c_int32),
                ("fixup_chunk", ctypes.c_int32), ("fixup_levels", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("temp_bytes", ctypes.c_uint64), ("coords_offset", ctypes.c_uint64),
                ("carries_offset", ctypes.c_uint64)]


def load_library() -> ctypes.CDLL:
    """dlopen libmspmv.so (built in-tree by `make -C merge_spmv_amd` or
    __graft_entry__.build()).  Fails loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise MspmvError(f"{path} not found: build the HIP extension first "
                         f"(python -c 'import __graft_entry__ as g; g.build()' or make -C merge_spmv_amd)")
    # torch bundles its own libamdhip64.so.7; importing it first makes the
    # dynamic linker bind libmspmv.so to that same HIP runtime (same SONAME),
    # so tensors' device pointers and streams are valid inside the library.
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    vp, i32, sz_p = ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t)
    for name in ("mspmv_csrmv_f32", "mspmv_csrmv_f64"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, sz_p, vp, vp, vp, vp, vp, i32, i32, i32, vp, ctypes.c_int]
    lib.mspmv_csrmv_axpby_f32.restype = ctypes.c_int
    lib.mspmv_csrmv_axpby_f32.argtypes = [vp, sz_p, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.c_float,
                                          ctypes.c_float, vp, ctypes.c_int]
    lib.mspmv_csrmv_axpby_f64.restype = ctypes.c_int
    lib.mspmv_csrmv_axpby_f64.argtypes = [vp, sz_p, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.c_double,
                                          ctypes.c_double, vp, ctypes.c_int]
    lib.mspmv_csrmv_prepare.restype = ctypes.c_int
    lib.mspmv_csrmv_prepare.argtypes = [vp, sz_p, vp, i32, i32, i32, vp, ctypes.c_int]
    lib.mspmv_csrmv_prepared_f32.restype = ctypes.c_int
    lib.mspmv_csrmv_prepared_f32.argtypes = [vp, sz_p, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.c_float,
                                             ctypes.c_float, vp, ctypes.c_int]
    lib.mspmv_csrmv_prepared_f64.restype = ctypes.c_int
    lib.mspmv_csrmv_prepared_f64.argtypes = [vp, sz_p, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.c_double,
                                             ctypes.c_double, vp, ctypes.c_int]
    lib.mspmv_csrmm_f32.restype = ctypes.c_int
    lib.mspmv_csrmm_f32.argtypes = [vp, sz_p, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, ctypes.c_float, ctypes.c_float,
                                    vp, ctypes.c_int]
    lib.mspmv_csrmm_f64.restype = ctypes.c_int
    lib.mspmv_csrmm_f64.argtypes = [vp, sz_p, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, ctypes.c_double, ctypes.c_double,
                                    vp, ctypes.c_int]
    lib.mspmv_error_string.restype = ctypes.c_char_p
    lib.mspmv_error_string.argtypes = [ctypes.c_int]
    lib.mspmv_version.restype = ctypes.c_int
    lib.mspmv_get_launch_info.restype = ctypes.c_int
    lib.mspmv_get_launch_info.argtypes = [i32, i32, i32, ctypes.POINTER(_LaunchInfo)]
    lib.mspmv_debug_read_tiles.restype = ctypes.c_int
    lib.mspmv_debug_read_tiles.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    lib.mspmv_set_tuning.restype = ctypes.c_int
    lib.mspmv_set_tuning.argtypes = [i32, i32, i32, i32]
    lib.mspmv_profile_begin.restype = ctypes.c_int
    lib.mspmv_profile_begin.argtypes = [i32]
    lib.mspmv_profile_end.restype = ctypes.c_int
    lib.mspmv_profile_end.argtypes = [ctypes.POINTER(ctypes.c_int32)] + [ctypes.POINTER(ctypes.c_float)] * 3
    lib.mspmv_mg_partition.restype = ctypes.c_int
    lib.mspmv_mg_partition.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, i32, vp, vp]
    lib.mspmv_mg_local_offsets.restype = ctypes.c_int
    lib.mspmv_mg_local_offsets.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int64, vp]
    lib.mspmv_mg_apply_carries.restype = ctypes.c_int
    lib.mspmv_mg_apply_carries.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    _lib = lib
    return lib


def _check(status: int, what: str) -> None:
    if status != 0:
        msg = load_library().mspmv_error_string(status)
        raise MspmvError(f"{what} failed: hipError {status} ({msg.decode() if msg else '?'})")


def _value_bytes(t) -> int:
    import torch
    if t.dtype == torch.float32:
        return 4
    if t.dtype == torch.float64:
        return 8
    raise TypeError(f"CsrMV is instantiated for float32 and float64 only (gpu_spmv.cu:730,734), got {t.dtype}")


def _ptr(t) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else 0)


def _stream_handle(stream) -> ctypes.c_void_p:
    import torch
    if stream is None:
        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))


class DeviceSpmv:
    """Mirror of ``cub::DeviceSpmv`` (reference cub/device/device_spmv.cuh:70-170)."""

    @staticmethod
    def CsrMV(d_temp_storage, temp_storage_bytes: int, d_values, d_row_offsets, d_column_indices,
              d_vector_x, d_vector_y, num_rows: int, num_cols: int, num_nonzeros: int,
              stream=None, debug_synchronous: bool = False, alpha: Optional[float] = None,
              beta: Optional[float] = None) -> Tuple[int, int]:
        """y = A*x.  Returns ``(status, temp_storage_bytes)``.

        ``d_temp_storage is None`` -> size query only (no work), exactly like the
        reference (dispatch_spmv_orig.cuh:651-655); otherwise a uint8 CUDA tensor of
        at least the queried size.  All d_* arguments are CUDA tensors (int32
        offsets/indices, float32|float64 values/x/y).  ``alpha``/``beta`` select the
        y = alpha*A*x + beta*y extension (mspmv_csrmv_axpby_*); leave None for the
        reference semantics.
        """
        lib = load_library()
        vb = _value_bytes(d_vector_y)
        size = ctypes.c_size_t(int(temp_storage_bytes))
        if d_temp_storage is None:
            temp_ptr = ctypes.c_void_p(0)
        else:
            size = ctypes.c_size_t(min(int(temp_storage_bytes), d_temp_storage.numel() * d_temp_storage.element_size()))
            temp_ptr = ctypes.c_void_p(d_temp_storage.data_ptr())
        args = [temp_ptr, ctypes.byref(size), _ptr(d_values), _ptr(d_row_offsets), _ptr(d_column_indices),
                _ptr(d_vector_x), _ptr(d_vector_y), int(num_rows), int(num_cols), int(num_nonzeros)]
        if alpha is None and beta is None:
            fn = lib.mspmv_csrmv_f32 if vb == 4 else lib.mspmv_csrmv_f64
        else:
            fn = lib.mspmv_csrmv_axpby_f32 if vb == 4 else lib.mspmv_csrmv_axpby_f64
            ct = ctypes.c_float if vb == 4 else ctypes.c_double
            args += [ct(1.0 if alpha is None else alpha), ct(0.0 if beta is None else beta)]
        args += [_stream_handle(stream) if d_temp_storage is not None else ctypes.c_void_p(0),
                 int(bool(debug_synchronous))]
        status = fn(*args)
        return int(status), int(size.value)


class CsrMVWorkspace:
    """Caller-owned temp storage for repeated CsrMV calls on one matrix shape
    (what TestGpuMergeCsrmv does by hand, gpu_spmv.cu:385-398)."""

    def __init__(self, num_rows: int, num_nonzeros: int, dtype, device="cuda"):
        import torch
        self.rows, self.nnz, self.dtype = int(num_rows), int(num_nonzeros), dtype
        probe = torch.empty(0, dtype=dtype)
        info = launch_info(self.rows, self.nnz, _value_bytes(probe))
        self.bytes = int(info["temp_bytes"])
        self.buffer = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        self.prepared_for = None

    def prepare(self, row_offsets, stream=None):
        """Run the tile-coordinate pass once for this matrix (mspmv_csrmv_prepare); later
        ``csrmv(..., workspace=ws)`` calls with the same row_offsets tensor skip it."""
        size = ctypes.c_size_t(self.bytes)
        _check(load_library().mspmv_csrmv_prepare(ctypes.c_void_p(self.buffer.data_ptr()), ctypes.byref(size), _ptr(row_offsets),
                                                  self.rows, self.nnz, _value_bytes(self.buffer.new_empty(0, dtype=self.dtype)),
                                                  _stream_handle(stream), 0), "mspmv_csrmv_prepare")
        self.prepared_for = (row_offsets.data_ptr(), self.rows, self.nnz)
        return self


def csrmv(values, row_offsets, column_indices, x, y=None, num_cols: Optional[int] = None,
          workspace: Optional[CsrMVWorkspace] = None, stream=None, alpha=None, beta=None,
          debug_synchronous: bool = False):
    """Convenience wrapper: size query + temp allocation + CsrMV.  Tensors must
    be contiguous CUDA tensors.  Returns y."""
    import torch
    if not values.is_cuda or not row_offsets.is_cuda or not x.is_cuda:
        raise MspmvError("csrmv needs CUDA (HIP) tensors: the merge-path kernels only run on the GPU")
    for t, name in ((values, "values"), (row_offsets, "row_offsets"), (column_indices, "column_indices"), (x, "x")):
        if not t.is_contiguous():
            raise MspmvError(f"{name} must be contiguous")
    if row_offsets.dtype != torch.int32 or column_indices.dtype != torch.int32:
        raise TypeError("row_offsets/column_indices must be int32 (OffsetT=int, gpu_spmv.cu:730,734)")
    rows = row_offsets.numel() - 1
    nnz = values.numel()
    cols = int(num_cols) if num_cols is not None else x.numel()
    if y is None:
        y = torch.empty(rows, dtype=values.dtype, device=values.device)
    if workspace is None:
        workspace = CsrMVWorkspace(rows, nnz, values.dtype, device=values.device)
    if workspace.prepared_for == (row_offsets.data_ptr(), rows, nnz):
        # coordinates already in the workspace (CsrMVWorkspace.prepare): mspmv_csrmv_prepared_*
        vb = _value_bytes(values)
        fn = load_library().mspmv_csrmv_prepared_f32 if vb == 4 else load_library().mspmv_csrmv_prepared_f64
        ct = ctypes.c_float if vb == 4 else ctypes.c_double
        size = ctypes.c_size_t(workspace.bytes)
        status = fn(ctypes.c_void_p(workspace.buffer.data_ptr()), ctypes.byref(size), _ptr(values), _ptr(row_offsets),
                    _ptr(column_indices), _ptr(x), _ptr(y), rows, cols, nnz, ct(1.0 if alpha is None else alpha),
                    ct(0.0 if beta is None else beta), _stream_handle(stream), int(bool(debug_synchronous)))
        _check(int(status), "mspmv_csrmv_prepared")
        return y
    status, _ = DeviceSpmv.CsrMV(workspace.buffer, workspace.bytes, values, row_offsets, column_indices, x, y,
                                 rows, cols, nnz, stream=stream, debug_synchronous=debug_synchronous,
                                 alpha=alpha, beta=beta)
    _check(status, "mspmv_csrmv")
    return y


def csrmm(values, row_offsets, column_indices, X, Y=None, alpha: float = 1.0, beta: float = 0.0, temp=None, stream=None,
          debug_synchronous: bool = False):
    """Y = alpha*A*X + beta*Y (mspmv_csrmm_*).  X: [cols, k] CUDA tensor, row-major (stride (ldx, 1));
    Y likewise [rows, k].  Returns Y."""
    import torch
    if not values.is_cuda or not X.is_cuda:
        raise MspmvError("csrmm needs CUDA (HIP) tensors: the merge-path kernels only run on the GPU")
    if X.dim() != 2 or X.stride(1) != 1:
        raise MspmvError("X must be 2-D with unit stride along the right-hand-side index (row-major)")
    rows, nnz, k = row_offsets.numel() - 1, values.numel(), X.shape[1]
    if Y is None:
        Y = torch.empty(rows, k, dtype=values.dtype, device=values.device)
    if Y.dim() != 2 or Y.stride(1) != 1 or Y.shape != (rows, k):
        raise MspmvError("Y must be a row-major [rows, k] tensor")
    vb = _value_bytes(values)
    fn = load_library().mspmv_csrmm_f32 if vb == 4 else load_library().mspmv_csrmm_f64
    ct = ctypes.c_float if vb == 4 else ctypes.c_double
    ldx = X.stride(0) if X.shape[0] > 1 else max(k, 1)
    ldy = Y.stride(0) if Y.shape[0] > 1 else max(k, 1)
    def call(tmp_ptr, size):
        return int(fn(tmp_ptr, ctypes.byref(size), _ptr(values), _ptr(row_offsets), _ptr(column_indices), _ptr(X), int(ldx),
                      _ptr(Y), int(ldy), rows, X.shape[0], nnz, k, ct(alpha), ct(beta), _stream_handle(stream),
                      int(bool(debug_synchronous))))
    size = ctypes.c_size_t(0)
    _check(call(ctypes.c_void_p(0), size), "mspmv_csrmm (size query)")
    if temp is None or temp.numel() < size.value:
        temp = torch.empty(max(int(size.value), 1), dtype=torch.uint8, device=values.device)
    size = ctypes.c_size_t(temp.numel())
    _check(call(ctypes.c_void_p(temp.data_ptr()), size), "mspmv_csrmm")
    return Y


def launch_info(num_rows: int, num_nonzeros: int, value_bytes: int) -> dict:
    info = _LaunchInfo()
    _check(load_library().mspmv_get_launch_info(int(num_rows), int(num_nonzeros), int(value_bytes),
                                                ctypes.byref(info)), "mspmv_get_launch_info")
    return {name: getattr(info, name) for name, _ in _LaunchInfo._fields_}


def set_tuning(value_bytes: int, block_threads: int = 0, items_per_thread: int = 0, flags: int = 0) -> None:
    _check(load_library().mspmv_set_tuning(int(value_bytes), int(block_threads), int(items_per_thread), int(flags)),
           "mspmv_set_tuning")


def profile_begin(max_calls: int) -> None:
    """Record hipEvents around the three kernels of the next `max_calls` CsrMV calls."""
    _check(load_library().mspmv_profile_begin(int(max_calls)), "mspmv_profile_begin")


def profile_end() -> dict:
    """Average per-call milliseconds of each pass since profile_begin()."""
    calls = ctypes.c_int32()
    ms = [ctypes.c_float() for _ in range(3)]
    _check(load_library().mspmv_profile_end(ctypes.byref(calls), *[ctypes.byref(m) for m in ms]), "mspmv_profile_end")
    return {"calls": calls.value, "search_ms": ms[0].value, "tile_ms": ms[1].value, "fixup_ms": ms[2].value}


def debug_read_tiles(workspace_buffer, num_rows: int, num_nonzeros: int, value_bytes: int, stream=None):
    """(coords[num_tiles+1, 2], carry_keys[num_tiles], carry_values[num_tiles])
    left in temp storage by the last CsrMV call, as numpy arrays."""
    import numpy as np
    info = launch_info(num_rows, num_nonzeros, value_bytes)
    nt = info["num_tiles"]
    coords = np.zeros((nt + 1, 2), dtype=np.int32)
    keys = np.zeros(max(nt, 1), dtype=np.int32)
    vals = np.zeros(max(nt, 1), dtype=np.float32 if value_bytes == 4 else np.float64)
    _check(load_library().mspmv_debug_read_tiles(
        ctypes.c_void_p(workspace_buffer.data_ptr()), int(num_rows), int(num_nonzeros), int(value_bytes),
        coords.ctypes.data_as(ctypes.c_void_p), keys.ctypes.data_as(ctypes.c_void_p),
        vals.ctypes.data_as(ctypes.c_void_p), _stream_handle(stream)), "mspmv_debug_read_tiles")
    return coords, keys[:nt], vals[:nt]
