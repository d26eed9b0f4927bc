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

__all__ = ["DeviceSpmv", "csrmv", "csrmm", "CsrMVWorkspace", "CsrMVPlan", "plan_bench_record", "library_path", "load_library", "launch_info",
           "set_tuning", "set_tdm", "clocked_bands", "debug_read_tiles", "profile_begin", "profile_end", "MspmvError",
           "TUNE_ATOMIC_FIX", "TUNE_NO_VEC"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAMES = {"product": "libmspmv.so", "dev": "libmspmv_dev.so"}
_libs = {}                 # kind -> ctypes.CDLL
_active = "product"

TUNE_ATOMIC_FIX = 2
TUNE_NO_VEC = 4


class MspmvError(RuntimeError):
    pass


def library_path(kind: Optional[str] = None) -> str:
    """The active library's file: libmspmv.so (the product: no setters, nothing read from the environment) or libmspmv_dev.so (the same
    sources built with -DMSPMV_TUNING: include/mspmv_dev.h) next to this file.  MSPMV_LIB=<path> replaces the PRODUCT library by another
    build of the same ABI (A/B tools)."""
    kind = kind or _active
    if kind == "product" and os.environ.get("MSPMV_LIB"):
        return os.environ["MSPMV_LIB"]
    return os.path.join(_HERE, _LIB_NAMES[kind])


def use_library(kind: str = "product") -> str:
    """Make `kind` ("product" | "dev") the library every wrapper of this module calls from now on; returns the previous kind.  The
    setters (set_tuning, set_band_passes, set_record_polls, set_compact_tiles) exist in the development library only and switch to
    it themselves when asked for a non-default value -- tests/conftest.py switches back to the product after every test."""
    global _active
    if kind not in _LIB_NAMES:
        raise ValueError(f"use_library: {kind!r} is not one of {sorted(_LIB_NAMES)}")
    prev, _active = _active, kind
    return prev


def active_library() -> str:
    return _active


class _LaunchInfo(ctypes.Structure):
    _fields_ = [("block_threads", ctypes.c_int32), ("items_per_thread", ctypes.c_int32),
                ("tile_items", ctypes.c_int32), ("num_tiles", ctypes.c_int32),
                ("fixup_chunk", ctypes.c_int32), ("fixup_levels", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("snap_head_max", ctypes.c_int32),
                ("temp_bytes", ctypes.c_uint64), ("coords_offset", ctypes.c_uint64),
                ("carries_offset", ctypes.c_uint64), ("diag_offset", ctypes.c_uint64), ("records_offset", ctypes.c_uint64)]


def load_library() -> ctypes.CDLL:
    """dlopen libmspmv.so (built in-tree by `make -C merge_spmv_amd` or
    __graft_entry__.build()).  Fails loudly when absent."""
    if _active in _libs:
        return _libs[_active]
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
    lib.mspmv_get_launch_info_cols.restype = ctypes.c_int
    lib.mspmv_get_launch_info_cols.argtypes = [i32, i32, i32, i32, ctypes.POINTER(_LaunchInfo)]
    lib.mspmv_debug_read_tiles.restype = ctypes.c_int
    lib.mspmv_debug_read_tiles.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    lib.mspmv_profile_begin.restype = ctypes.c_int
    lib.mspmv_profile_begin.argtypes = [i32]
    lib.mspmv_profile_end.restype = ctypes.c_int
    lib.mspmv_profile_end.argtypes = [ctypes.POINTER(ctypes.c_int32)] + [ctypes.POINTER(ctypes.c_float)] * 3
    if hasattr(lib, "mspmv_set_tuning"):          # (the development library, or an MSPMV_LIB build that has the setters)
        lib.mspmv_set_tuning.restype = ctypes.c_int
        lib.mspmv_set_tuning.argtypes = [i32, i32, i32, i32]
        lib.mspmv_set_band_passes.restype = ctypes.c_int
        lib.mspmv_set_band_passes.argtypes = [ctypes.c_int32, ctypes.c_int32]
        lib.mspmv_set_record_polls.restype = ctypes.c_int
        lib.mspmv_set_record_polls.argtypes = [ctypes.c_int32]
        lib.mspmv_set_compact_tiles.restype = ctypes.c_int
        lib.mspmv_set_compact_tiles.argtypes = [ctypes.c_int32]
    lib.mspmv_get_band_passes.restype = ctypes.c_int
    lib.mspmv_get_band_passes.argtypes = [ctypes.c_int32] * 4 + [ctypes.POINTER(ctypes.c_int32)]
    lib.mspmv_debug_band_windows.restype = ctypes.c_int
    lib.mspmv_debug_band_windows.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    lib.mspmv_mg_partition.restype = ctypes.c_int
    lib.mspmv_mg_partition.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, i32, vp, vp]
    lib.mspmv_mg_local_offsets.restype = ctypes.c_int
    lib.mspmv_mg_local_offsets.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int64, vp]
    lib.mspmv_mg_apply_carries.restype = ctypes.c_int
    lib.mspmv_mg_apply_carries.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    i64, u64 = ctypes.c_int64, ctypes.c_uint64
    lib.mspmv_csrmv_plan_size.restype = ctypes.c_int
    lib.mspmv_csrmv_plan_size.argtypes = [i32, i32, i32, i32, i32, sz_p, ctypes.POINTER(i32)]
    for name, ct in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
        fn = getattr(lib, "mspmv_csrmv_plan_build_" + name)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, i32, i32, i32, i32, vp, ctypes.c_int]
        fn = getattr(lib, "mspmv_csrmv_plan_apply_" + name)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, ctypes.c_size_t, vp, vp, i32, i32, i32, i32, ct, ct, vp, ctypes.c_int]
    lib.mspmv_csrmv_hotcols_size.restype = ctypes.c_int
    lib.mspmv_csrmv_hotcols_size.argtypes = [i32, i32, i32, i32, sz_p]
    lib.mspmv_csrmv_hotcols_build.restype = ctypes.c_int
    lib.mspmv_csrmv_hotcols_build.argtypes = [vp, ctypes.c_size_t, vp, vp, i32, i32, i32, i32, vp, ctypes.c_int]
    for name, ct in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
        for stem in ("mspmv_csrmv_hotcols_apply_", "mspmv_csrmv_hotcols_apply_permuted_"):
            fn = getattr(lib, stem + name)
            fn.restype = ctypes.c_int
            fn.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp, i32, i32, i32, ct, ct, vp, ctypes.c_int]
        fn = getattr(lib, "mspmv_csrmv_hotcols_permute_" + name)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp, ctypes.c_size_t, vp, vp, i32, i32, i32, vp, ctypes.c_int]
    for name in ("mspmv_csrmv_hotcols_order", "mspmv_csrmv_hotcols_columns"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp, i32, i32, i32, i32]
    lib.mspmv_mg_unique_id.restype = ctypes.c_int
    lib.mspmv_mg_unique_id.argtypes = [vp]
    lib.mspmv_mg_plan_create.restype = ctypes.c_int
    lib.mspmv_mg_plan_create.argtypes = [ctypes.POINTER(vp), i32, i32, vp, vp, vp, vp, i64, i32, i32, vp]
    lib.mspmv_mg_plan_set_part.restype = ctypes.c_int
    lib.mspmv_mg_plan_set_part.argtypes = [vp, i32, vp, vp, vp]
    lib.mspmv_mg_plan_hot_columns.restype = ctypes.c_int
    lib.mspmv_mg_plan_hot_columns.argtypes = [vp, i32]
    lib.mspmv_mg_plan_ipc_export.restype = ctypes.c_int
    lib.mspmv_mg_plan_ipc_export.argtypes = [vp, vp, sz_p]
    lib.mspmv_mg_plan_ipc_import.restype = ctypes.c_int
    lib.mspmv_mg_plan_ipc_import.argtypes = [vp, vp, i32, ctypes.c_size_t]
    for name in ("mspmv_mg_plan_x", "mspmv_mg_plan_y", "mspmv_mg_plan_stream"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp, i32]
    lib.mspmv_mg_plan_info.restype = ctypes.c_int
    lib.mspmv_mg_plan_info.argtypes = [vp, vp]
    for name in ("mspmv_mg_csrmv", "mspmv_mg_allgather_rows", "mspmv_mg_synchronize", "mspmv_mg_plan_destroy"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = [vp]
    _libs[_active] = lib
    return lib


def _setter(name: str, default: bool, *args) -> None:
    """The setters of include/mspmv_dev.h.  The product library has none: asking it for the defaults is a no-op; asking for anything else
    switches this module to the development library (use_library("dev")) -- the same kernels with the per-thread overrides compiled in."""
    lib = load_library()
    if not hasattr(lib, name):
        if default:
            if "dev" in _libs and _libs["dev"] is not lib:        # (a test may have left an override in the development library's thread state)
                _check(getattr(_libs["dev"], name)(*args), name)
            return
        use_library("dev")
        lib = load_library()
    _check(getattr(lib, name)(*args), name)


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


def _validate(values, row_offsets, column_indices, x, y, rows: int, cols: int, nnz: int, what: str,
              x_rows: Optional[int] = None) -> None:
    """The checks the C ABI cannot make (it sees raw pointers): every csrmv / csrmm / DeviceSpmv.CsrMV
    call goes through here, so a wrong dtype, device, stride or length is an MspmvError instead of
    a reinterpretation of memory.  x / y may be 1-D (CsrMV) or 2-D row-major (SpMM)."""
    import torch
    dev = y.device if y is not None else values.device
    if dev.type != "cuda":
        raise MspmvError(f"{what} needs CUDA (HIP) tensors: the merge-path kernels only run on the GPU")
    if y is None or y.dtype not in (torch.float32, torch.float64):
        raise MspmvError(f"{what}: y must be a float32 or float64 tensor (gpu_spmv.cu:730,734)")
    if rows < 0 or cols < 0 or nnz < 0:
        raise MspmvError(f"{what}: negative size")
    for t, name, need in ((row_offsets, "row_offsets", rows + 1), (column_indices, "column_indices", nnz)):
        if need == 0 and (t is None or t.numel() == 0):
            continue
        if t is None or t.dtype != torch.int32:
            raise MspmvError(f"{what}: {name} must be int32 (OffsetT=int, gpu_spmv.cu:730,734), got {None if t is None else t.dtype}")
        if t.device != dev or not t.is_contiguous() or t.numel() < need:
            raise MspmvError(f"{what}: {name} must be a contiguous tensor on {dev} with at least {need} entries")
    if nnz > 0:
        if values is None or values.dtype != y.dtype or values.device != dev or not values.is_contiguous() or values.numel() < nnz:
            raise MspmvError(f"{what}: values must be a contiguous {y.dtype} tensor on {dev} with at least {nnz} entries")
    for t, name, need in ((x, "x", cols if x_rows is None else x_rows), (y, "y", rows)):
        if t is None:
            if need == 0 or (name == "x" and nnz == 0):
                continue
            raise MspmvError(f"{what}: {name} is missing")
        if t.dtype != y.dtype or t.device != dev:
            raise MspmvError(f"{what}: {name} must be {y.dtype} on {dev}, got {t.dtype} on {t.device}")
        if t.dim() == 1:
            if t.numel() > 1 and t.stride(0) != 1:
                raise MspmvError(f"{what}: {name} must have unit stride")
            if t.numel() < need and not (name == "x" and nnz == 0):
                raise MspmvError(f"{what}: {name} has {t.numel()} entries, needs {need}")
        elif t.dim() == 2:
            if t.shape[1] > 1 and t.stride(1) != 1:
                raise MspmvError(f"{what}: {name} must be row-major (unit stride along the right-hand-side index)")
            if t.shape[0] < need and not (name == "x" and nnz == 0):
                raise MspmvError(f"{what}: {name} has {t.shape[0]} rows, needs {need}")
        else:
            raise MspmvError(f"{what}: {name} must be 1-D or 2-D")


class DeviceSpmv:
    """Mirror of ``cub::DeviceSpmv`` (reference cub/device/device_spmv.cuh:70-170)."""

    @staticmethod
    def CsrMV(d_temp_storage, temp_storage_bytes: int, d_values, d_row_offsets, d_column_indices,
              d_vector_x, d_vector_y, num_rows: int, num_cols: int, num_nonzeros: int,
              stream=None, debug_synchronous: bool = False, alpha: Optional[float] = None,
              beta: Optional[float] = None, _checked: bool = False) -> Tuple[int, int]:
        """y = A*x.  Returns ``(status, temp_storage_bytes)``.

        ``d_temp_storage is None`` -> size query only (no work), exactly like the
        reference (dispatch_spmv_orig.cuh:651-655); otherwise a uint8 CUDA tensor of
        at least the queried size.  All d_* arguments are CUDA tensors (int32
        offsets/indices, float32|float64 values/x/y).  ``alpha``/``beta`` select the
        y = alpha*A*x + beta*y extension (mspmv_csrmv_axpby_*); leave None for the
        reference semantics.  Wrong dtypes / devices / strides / lengths raise MspmvError
        before anything reaches the library.
        """
        lib = load_library()
        vb = _value_bytes(d_vector_y)
        size = ctypes.c_size_t(int(temp_storage_bytes))
        if d_temp_storage is None:
            temp_ptr = ctypes.c_void_p(0)
        else:
            if not _checked:
                _validate(d_values, d_row_offsets, d_column_indices, d_vector_x, d_vector_y, int(num_rows), int(num_cols),
                          int(num_nonzeros), "DeviceSpmv.CsrMV")
            if not d_temp_storage.is_cuda or not d_temp_storage.is_contiguous():
                raise MspmvError("DeviceSpmv.CsrMV: d_temp_storage must be a contiguous CUDA tensor")
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
        self.value_bytes = _value_bytes(probe)
        info = launch_info(self.rows, self.nnz, self.value_bytes)
        self.bytes = int(info["temp_bytes"])
        self.buffer = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        self._checked = None            # the tensors of the last validated csrmv call through this workspace
        self.prepared_for = None        # the row_offsets TENSOR the coordinates in `buffer` belong to
        self.prepared_info = None       # launch_info at prepare time (tile shape / flags / tile count)

    def prepare(self, row_offsets, stream=None):
        """Run the tile-coordinate pass once for this matrix (mspmv_csrmv_prepare); later
        ``csrmv(..., workspace=ws)`` calls with the SAME row_offsets tensor (identity, not address)
        and unchanged tuning skip it."""
        import torch
        if row_offsets.dtype != torch.int32 or not row_offsets.is_cuda or not row_offsets.is_contiguous() or \
                row_offsets.numel() < self.rows + 1:
            raise MspmvError("prepare: row_offsets must be a contiguous int32 CUDA tensor with rows + 1 entries")
        size = ctypes.c_size_t(self.bytes)
        _check(load_library().mspmv_csrmv_prepare(ctypes.c_void_p(self.buffer.data_ptr()), ctypes.byref(size), _ptr(row_offsets),
                                                  self.rows, self.nnz, self.value_bytes,
                                                  _stream_handle(stream), 0), "mspmv_csrmv_prepare")
        self.prepared_for = row_offsets           # keeps the tensor alive: its address cannot be recycled
        self.prepared_info = launch_info(self.rows, self.nnz, self.value_bytes)
        return self

    def is_prepared_for(self, row_offsets, rows: int, nnz: int, dtype) -> bool:
        return (self.prepared_for is row_offsets and rows == self.rows and nnz == self.nnz and dtype == self.dtype and
                self.prepared_info == launch_info(rows, nnz, self.value_bytes))


def csrmv(values, row_offsets, column_indices, x, y=None, num_cols: Optional[int] = None,
          workspace: Optional[CsrMVWorkspace] = None, stream=None, alpha=None, beta=None,
          debug_synchronous: bool = False):
    """Convenience wrapper: size query + temp allocation + CsrMV.  Tensors must
    be contiguous CUDA tensors.  Returns y."""
    import torch
    if not values.is_cuda or not row_offsets.is_cuda or not x.is_cuda:
        raise MspmvError("csrmv needs CUDA (HIP) tensors: the merge-path kernels only run on the GPU")
    if row_offsets.dtype != torch.int32 or column_indices.dtype != torch.int32:
        raise TypeError("row_offsets/column_indices must be int32 (OffsetT=int, gpu_spmv.cu:730,734)")
    rows = row_offsets.numel() - 1
    nnz = values.numel()
    cols = int(num_cols) if num_cols is not None else x.numel()
    if y is None:
        y = torch.empty(rows, dtype=values.dtype, device=values.device)
    if workspace is None:
        _validate(values, row_offsets, column_indices, x, y, rows, cols, nnz, "csrmv")
        workspace = CsrMVWorkspace(rows, nnz, values.dtype, device=values.device)
    else:
        if workspace.rows != rows or workspace.nnz != nnz or workspace.dtype != values.dtype:
            raise MspmvError("csrmv: the workspace was sized for another matrix shape or precision")
        # repeated calls with the very same tensor objects (a solver loop, a timing loop) are checked once: the
        # workspace keeps the checked tensors alive, so their identities cannot be recycled
        checked = workspace._checked
        if not (checked is not None and checked[0] is values and checked[1] is row_offsets and checked[2] is column_indices
                and checked[3] is x and checked[4] is y and checked[5] == cols):
            _validate(values, row_offsets, column_indices, x, y, rows, cols, nnz, "csrmv")
            workspace._checked = (values, row_offsets, column_indices, x, y, cols)
    if workspace.is_prepared_for(row_offsets, rows, nnz, values.dtype):
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
                                 alpha=alpha, beta=beta, _checked=True)
    _check(status, "mspmv_csrmv")
    return y


def csrmm(values, row_offsets, column_indices, X, Y=None, alpha: float = 1.0, beta: float = 0.0, temp=None, stream=None,
          debug_synchronous: bool = False):
    """Y = alpha*A*X + beta*Y (mspmv_csrmm_*).  X: [cols, k] CUDA tensor, row-major (stride (ldx, 1));
    Y likewise [rows, k].  Returns Y."""
    import torch
    if not values.is_cuda or not X.is_cuda:
        raise MspmvError("csrmm needs CUDA (HIP) tensors: the merge-path kernels only run on the GPU")
    if X.dim() != 2 or (X.shape[1] > 1 and X.stride(1) != 1):
        raise MspmvError("X must be 2-D with unit stride along the right-hand-side index (row-major)")
    rows, nnz, k = row_offsets.numel() - 1, values.numel(), X.shape[1]
    if Y is None:
        Y = torch.empty(rows, k, dtype=values.dtype, device=values.device)
    if Y.dim() != 2 or (k > 1 and Y.stride(1) != 1) or Y.shape != (rows, k):
        raise MspmvError("Y must be a row-major [rows, k] tensor")
    _validate(values, row_offsets, column_indices, X, Y, rows, X.shape[0], nnz, "csrmm")
    vb = _value_bytes(values)
    fn = load_library().mspmv_csrmm_f32 if vb == 4 else load_library().mspmv_csrmm_f64
    ct = ctypes.c_float if vb == 4 else ctypes.c_double
    ldx = X.stride(0) if X.shape[0] > 1 else max(k, 1)
    ldy = Y.stride(0) if Y.shape[0] > 1 else max(k, 1)
    if ldx < k or ldy < k:
        raise MspmvError("csrmm: leading dimensions must be at least k (no overlapping / broadcast rows)")
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


class CsrMVPlan:
    """The opt-in prepared plan (mspmv_csrmv_plan_*): a band-major copy of the matrix made once, so that every
    XCD gathers from an L2-sized slice of x.  For a matrix that is multiplied many times and whose x does not
    fit an XCD's 4 MiB L2; the stateless `csrmv` never uses it.  `plan(x, y)` computes y = alpha*A*x + beta*y."""

    def __init__(self, values, row_offsets, column_indices, num_cols: int, bands: int = 0, stream=None):
        import torch
        self.rows, self.cols, self.nnz = row_offsets.numel() - 1, int(num_cols), values.numel()
        y_probe = torch.empty(0, dtype=values.dtype, device=values.device)
        _validate(values, row_offsets, column_indices, None, torch.empty(self.rows, dtype=values.dtype, device=values.device)
                  if self.rows == 0 else y_probe.new_empty(self.rows), self.rows, 0, self.nnz, "CsrMVPlan")
        self.dtype = values.dtype
        self.vb = _value_bytes(values)
        size = ctypes.c_size_t(0); used = ctypes.c_int32(0)
        _check(load_library().mspmv_csrmv_plan_size(self.rows, self.cols, self.nnz, self.vb, int(bands), ctypes.byref(size),
                                                    ctypes.byref(used)), "mspmv_csrmv_plan_size")
        self.bytes, self.bands = int(size.value), int(used.value)
        self.storage = torch.empty(max(self.bytes, 1), dtype=torch.uint8, device=values.device)
        fn = load_library().mspmv_csrmv_plan_build_f32 if self.vb == 4 else load_library().mspmv_csrmv_plan_build_f64
        _check(fn(ctypes.c_void_p(self.storage.data_ptr()), self.bytes, _ptr(values), _ptr(row_offsets), _ptr(column_indices),
                  self.rows, self.cols, self.nnz, self.bands, _stream_handle(stream), 0), "mspmv_csrmv_plan_build")

    def __call__(self, x, y=None, alpha: float = 1.0, beta: float = 0.0, stream=None, debug_synchronous: bool = False):
        import torch
        if y is None:
            y = torch.empty(self.rows, dtype=self.dtype, device=self.storage.device)
        if x.dtype != self.dtype or y.dtype != self.dtype or x.device != self.storage.device or y.device != self.storage.device \
                or not x.is_contiguous() or not y.is_contiguous() or x.dim() != 1 or y.dim() != 1 \
                or x.numel() < self.cols or y.numel() < self.rows:
            raise MspmvError("CsrMVPlan: x / y must be contiguous 1-D tensors of the plan's dtype on its device, with at least cols / rows entries")
        fn = load_library().mspmv_csrmv_plan_apply_f32 if self.vb == 4 else load_library().mspmv_csrmv_plan_apply_f64
        _check(fn(ctypes.c_void_p(self.storage.data_ptr()), self.bytes, _ptr(x), _ptr(y), self.rows, self.cols, self.nnz, self.bands,
                  float(alpha), float(beta), _stream_handle(stream), int(bool(debug_synchronous))), "mspmv_csrmv_plan_apply")
        return y


class CsrMVHotColumns:
    """The opt-in hot-column plan (mspmv_csrmv_hotcols_*): the columns renumbered once by how often the matrix references
    them, so that the hot part of a huge x is contiguous and stays in the caches (scale-free graphs; BASELINE config 5).
    Values and row offsets are used from the caller's tensors (kept alive here); y is bit for bit the stateless result."""

    def __init__(self, values, row_offsets, column_indices, num_cols: int, stream=None):
        import torch
        self.rows, self.cols, self.nnz = row_offsets.numel() - 1, int(num_cols), values.numel()
        _validate(values, row_offsets, column_indices, None, torch.empty(self.rows, dtype=values.dtype, device=values.device),
                  self.rows, 0, self.nnz, "CsrMVHotColumns")
        self.dtype, self.vb = values.dtype, _value_bytes(values)
        self.values, self.row_offsets = values, row_offsets
        size = ctypes.c_size_t(0)
        _check(load_library().mspmv_csrmv_hotcols_size(self.rows, self.cols, self.nnz, self.vb, ctypes.byref(size)), "mspmv_csrmv_hotcols_size")
        self.bytes = int(size.value)
        self.storage = torch.empty(max(self.bytes, 1), dtype=torch.uint8, device=values.device)
        _check(load_library().mspmv_csrmv_hotcols_build(ctypes.c_void_p(self.storage.data_ptr()), self.bytes, _ptr(row_offsets), _ptr(column_indices),
                                                        self.rows, self.cols, self.nnz, self.vb, _stream_handle(stream), 0), "mspmv_csrmv_hotcols_build")

    def _view(self, fn, count):
        import torch
        ptr = fn(ctypes.c_void_p(self.storage.data_ptr()), self.rows, self.cols, self.nnz, self.vb)
        if not ptr or count == 0:
            return torch.empty(0, dtype=torch.int32, device=self.storage.device)
        off = int(ptr) - self.storage.data_ptr()
        return self.storage[off: off + 4 * count].view(torch.int32)

    def order(self):
        """order[k] = the original column that became column k"""
        return self._view(load_library().mspmv_csrmv_hotcols_order, self.cols)

    def columns(self):
        """the renumbered column indices"""
        return self._view(load_library().mspmv_csrmv_hotcols_columns, self.nnz)

    def permute(self, x, out=None, stream=None):
        """x in the plan's numbering (mspmv_csrmv_hotcols_permute_*): out[k] = x[order[k]]; pass the result to __call__(..., x_is_permuted=True)"""
        import torch
        if out is None:
            out = torch.empty(max(self.cols, 1), dtype=self.dtype, device=self.storage.device)[:self.cols]
        if x.dtype != self.dtype or out.dtype != self.dtype or not x.is_contiguous() or not out.is_contiguous() or x.numel() < self.cols or out.numel() < self.cols \
                or x.data_ptr() == out.data_ptr():
            raise MspmvError("CsrMVHotColumns.permute: x / out must be distinct contiguous tensors of the plan's dtype with at least cols entries")
        lib = load_library()
        fn = lib.mspmv_csrmv_hotcols_permute_f32 if self.vb == 4 else lib.mspmv_csrmv_hotcols_permute_f64
        _check(fn(ctypes.c_void_p(self.storage.data_ptr()), self.bytes, _ptr(x), _ptr(out), self.rows, self.cols, self.nnz, _stream_handle(stream), 0),
               "mspmv_csrmv_hotcols_permute")
        return out

    def __call__(self, x, y=None, alpha: float = 1.0, beta: float = 0.0, stream=None, debug_synchronous: bool = False, x_is_permuted: bool = False):
        import torch
        if y is None:
            y = torch.empty(self.rows, dtype=self.dtype, device=self.storage.device)
        if x.dtype != self.dtype or y.dtype != self.dtype or x.device != self.storage.device or y.device != self.storage.device \
                or not x.is_contiguous() or not y.is_contiguous() or x.dim() != 1 or y.dim() != 1 \
                or x.numel() < self.cols or y.numel() < self.rows:
            raise MspmvError("CsrMVHotColumns: x / y must be contiguous 1-D tensors of the plan's dtype on its device, with at least cols / rows entries")
        lib = load_library()
        if x_is_permuted:
            fn = lib.mspmv_csrmv_hotcols_apply_permuted_f32 if self.vb == 4 else lib.mspmv_csrmv_hotcols_apply_permuted_f64
        else:
            fn = lib.mspmv_csrmv_hotcols_apply_f32 if self.vb == 4 else lib.mspmv_csrmv_hotcols_apply_f64
        _check(fn(ctypes.c_void_p(self.storage.data_ptr()), self.bytes, _ptr(self.values), _ptr(self.row_offsets), _ptr(x), _ptr(y), self.rows, self.cols,
                  self.nnz, float(alpha), float(beta), _stream_handle(stream), int(bool(debug_synchronous))), "mspmv_csrmv_hotcols_apply")
        return y


def hotcols_bench_record(A, x, y_stateless, steps: int = 5, warmup: int = 2, peak_gbs: float = 8000.0) -> dict:
    """bench.py's `hot_column_plan` sub-record: set-up time, SpMV time (x permutation included), agreement with the stateless y."""
    import time
    import torch
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = CsrMVHotColumns(A.values, A.row_offsets, A.column_indices, A.cols)
    torch.cuda.synchronize(); setup_ms = (time.perf_counter() - t0) * 1e3
    y = torch.empty_like(y_stateless)
    for _ in range(max(warmup, 1)):
        plan(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        plan(x, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    vb = A.values.element_size()
    b_alg = A.nnz * (vb + 4) + (A.rows + 1) * 4 + A.rows * vb + A.cols * vb
    xp = plan.permute(x)
    yp = torch.empty_like(y_stateless)
    for _ in range(max(warmup, 1)):
        plan(xp, yp, x_is_permuted=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        plan(xp, yp, x_is_permuted=True)
    torch.cuda.synchronize()
    ms_p = (time.perf_counter() - t0) * 1e3 / steps
    permuted = {"api": "mspmv_csrmv_hotcols_permute_* once, mspmv_csrmv_hotcols_apply_permuted_* per SpMV (the caller keeps x in the plan's numbering)",
                "ms_per_step": round(ms_p, 5), "value": round(2.0 * A.nnz / (ms_p * 1e-3) / 1e9, 3), "frac": round(b_alg / (ms_p * 1e-3) / 1e9 / peak_gbs, 4),
                "bitwise_equal_to_apply": bool(torch.equal(yp, y))}
    return {"api": "mspmv_csrmv_hotcols_build once, then mspmv_csrmv_hotcols_apply_* per SpMV (opt-in; not the drop-in call)", "x_kept_permuted": permuted,
            "what": "columns renumbered by reference count (hot columns contiguous); x permuted once per SpMV, inside the timed call",
            "setup_ms": round(setup_ms, 3), "storage_bytes": plan.bytes, "ms_per_step": round(ms, 5),
            "value": round(2.0 * A.nnz / (ms * 1e-3) / 1e9, 3), "unit": "GFLOP/s",
            "roofline": {"bound": "hbm", "achieved": round(b_alg / (ms * 1e-3) / 1e9, 2), "peak": peak_gbs, "unit": "GB/s",
                         "frac": round(b_alg / (ms * 1e-3) / 1e9 / peak_gbs, 4),
                         "note": "algorithmic bytes of the ORIGINAL matrix / whole plan SpMV (x permutation + tile kernel)"},
            "bitwise_equal_to_stateless": bool(torch.equal(y, y_stateless)),
            "bitwise_equal_to_stateless_one_launch": _equal_to_one_launch(A, x, y),
            "bitwise_note": "the plan's y is bit for bit the stateless call's in its one-launch form; a stateless call that is a CANDIDATE for the "
                            "column-band passes (mspmv_get_band_passes > 1) runs the classic three launches, another association: compared "
                            "with mspmv_set_band_passes(vb, -1) in `bitwise_equal_to_stateless_one_launch`"}


def _equal_to_one_launch(A, x, y_plan) -> bool:
    """y_plan == the stateless call's y with the column-band candidacy switched off (mspmv_set_band_passes(vb, -1))"""
    import torch
    vb = A.values.element_size()
    prev = active_library()
    try:
        set_band_passes(vb, -1)               # (development library: the same kernels + the override)
        y = csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=A.cols)
        torch.cuda.synchronize()
    finally:
        set_band_passes(vb, 0)
        use_library(prev)
    return bool(torch.equal(y, y_plan))


def plan_bench_record(A, x, y_stateless, steps: int = 50, warmup: int = 5, peak_gbs: float = 8000.0) -> dict:
    """bench.py's `prepared_plan` sub-record: set-up time of the plan, its SpMV time on the same matrix,
    agreement with the stateless call's y."""
    import time
    import torch
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = CsrMVPlan(A.values, A.row_offsets, A.column_indices, A.cols)
    torch.cuda.synchronize(); setup_ms = (time.perf_counter() - t0) * 1e3
    y = torch.empty_like(y_stateless)
    for _ in range(max(warmup, 1)):
        plan(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        plan(x, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    profile_begin(steps)
    for _ in range(steps):
        plan(x, y)
    torch.cuda.synchronize()
    prof = profile_end()
    vb = A.values.element_size()
    b_alg = A.nnz * (vb + 4) + (A.rows + 1) * 4 + A.rows * vb + A.cols * vb
    diff = float((y.double() - y_stateless.double()).abs().max())
    scale = float(y_stateless.double().abs().max())
    return {"api": "mspmv_csrmv_plan_build_* once, then mspmv_csrmv_plan_apply_* per SpMV (opt-in; not the drop-in call)",
            "bands": plan.bands, "setup_ms": round(setup_ms, 3), "storage_bytes": plan.bytes,
            "ms_per_step": round(ms, 5), "value": round(2.0 * A.nnz / (ms * 1e-3) / 1e9, 3), "unit": "GFLOP/s",
            "tile_kernel_ms": round(prof["tile_ms"], 5),
            "roofline": {"bound": "hbm", "achieved": round(b_alg / (ms * 1e-3) / 1e9, 2), "peak": peak_gbs, "unit": "GB/s",
                         "frac": round(b_alg / (ms * 1e-3) / 1e9 / peak_gbs, 4),
                         "note": "algorithmic bytes of the ORIGINAL matrix / whole plan SpMV (tile kernel + fix-up + band fold)"},
            "max_abs_diff_vs_stateless": diff, "max_abs_y": scale}


def launch_info(num_rows: int, num_nonzeros: int, value_bytes: int, num_cols: Optional[int] = None) -> dict:
    """mspmv_get_launch_info; with `num_cols` mspmv_get_launch_info_cols: exactly the layout a stateless call of these sizes runs (one
    family of calls -- large fp64 matrices of short rows over a tiny x -- picks its tile shape by the column count too)."""
    info = _LaunchInfo()
    if num_cols is None:
        _check(load_library().mspmv_get_launch_info(int(num_rows), int(num_nonzeros), int(value_bytes),
                                                    ctypes.byref(info)), "mspmv_get_launch_info")
    else:
        _check(load_library().mspmv_get_launch_info_cols(int(num_rows), int(num_cols), int(num_nonzeros), int(value_bytes),
                                                         ctypes.byref(info)), "mspmv_get_launch_info_cols")
    return {name: getattr(info, name) for name, _ in _LaunchInfo._fields_}


def serial_sum_depth(num_rows: int, num_cols: int, num_nonzeros: int, value_bytes: int, extra: int = 0) -> int:
    """How many products one thread of the tile kernel adds up serially before the scan tree takes over, for a call of these
    sizes -- the compiled tile's items per thread rounded up to whole 4-element chunks -- plus one re-association per
    column-band pass the call may run, plus `extra` (parts of a multi-GPU split, bands of a prepared plan).  It is the
    `items_per_thread` term of the stated error bound |y - g| <= 2 (ceil(log2(len + 1)) + items_per_thread + 8) eps s
    (DESIGN.md 3): derived from the compiled shape, so the bound follows the library when a shape changes."""
    ipt = launch_info(num_rows, num_nonzeros, value_bytes)["items_per_thread"]
    return 4 * (ipt // 4 + 1) + max(band_passes(num_rows, num_cols, num_nonzeros, value_bytes), 0) + int(extra)


def set_tuning(value_bytes: int, block_threads: int = 0, items_per_thread: int = 0, flags: int = 0) -> None:
    """mspmv_set_tuning (development library: a non-default value switches to it, see _setter)."""
    _setter("mspmv_set_tuning", block_threads == 0 and items_per_thread == 0 and flags == 0,
            int(value_bytes), int(block_threads), int(items_per_thread), int(flags))


def profile_begin(max_calls: int) -> None:
    """Record hipEvents around the three kernels of the next `max_calls` CsrMV calls."""
    _check(load_library().mspmv_profile_begin(int(max_calls)), "mspmv_profile_begin")


def profile_end() -> dict:
    """Average per-call milliseconds of each pass since profile_begin()."""
    calls = ctypes.c_int32()
    ms = [ctypes.c_float() for _ in range(3)]
    _check(load_library().mspmv_profile_end(ctypes.byref(calls), *[ctypes.byref(m) for m in ms]), "mspmv_profile_end")
    return {"calls": calls.value, "search_ms": ms[0].value, "tile_ms": ms[1].value, "fixup_ms": ms[2].value}


def set_band_passes(value_bytes: int, passes: int = 0) -> None:
    """Column-band passes (mspmv_set_band_passes): 0 automatic, < 0 never, >= 2 always that many."""
    _setter("mspmv_set_band_passes", passes == 0, int(value_bytes), int(passes))


def set_tdm(value_bytes: int, policy: int = 0, slot_permille: int = 0, lookahead_plus_1: int = 0, band_shift: int = 0) -> None:
    """Clock-scheduled column bands (mspmv_set_tdm): policy 0 the library's rule, < 0 never (the passes), > 0 always where the passes are offered."""
    _setter("mspmv_set_tdm", policy == 0 and slot_permille == 0 and lookahead_plus_1 == 0 and band_shift == 0,
            int(value_bytes), int(policy), int(slot_permille), int(lookahead_plus_1), int(band_shift))


def device_caches() -> dict:
    """What the column-band policy is derived from (mspmv_get_device_caches): one XCD's L2 bytes, XCD count, CU count."""
    l2 = ctypes.c_int64(0); x = ctypes.c_int32(0); c = ctypes.c_int32(0)
    lib = load_library()
    lib.mspmv_get_device_caches.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    _check(lib.mspmv_get_device_caches(ctypes.byref(l2), ctypes.byref(x), ctypes.byref(c)), "mspmv_get_device_caches")
    return {"l2_bytes_per_xcd": l2.value, "xcds": x.value, "cus": c.value}


def set_record_polls(polls: int = 0) -> None:
    """Testing aid (mspmv_set_record_polls): 0 = library default, 1 = one look, -1 = never look: tiles in which a long row ends
    compute the pieces held by other workgroups themselves instead of taking the published records."""
    _setter("mspmv_set_record_polls", polls == 0, int(polls))


def cache_stream_rate(nbytes: int, reps: int = 20) -> float:
    """GB/s of a bare 16-byte-per-lane read stream (mspmv_probe_read_stream, ordinary loads) over a buffer of `nbytes` that has been
    read before -- for nbytes within the 256 MB Infinity Cache this is the rate out of that cache (and the L2s), the bound a
    cache-resident SpMV's algorithmic bytes are to be read against (bench.py)."""
    import time
    import torch
    lib = load_library()
    lib.mspmv_probe_read_stream.restype = ctypes.c_int
    lib.mspmv_probe_read_stream.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_void_p]
    n = max(int(nbytes) // 16 * 16, 16)
    buf = torch.zeros(n, dtype=torch.uint8, device="cuda")
    st = _stream_handle(None)
    for _ in range(3):
        _check(lib.mspmv_probe_read_stream(ctypes.c_void_p(buf.data_ptr()), n, 0, st), "mspmv_probe_read_stream")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        lib.mspmv_probe_read_stream(ctypes.c_void_p(buf.data_ptr()), n, 0, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    del buf
    return n / dt / 1e9


def set_compact_tiles(max_tiles: int = 0) -> None:
    """Testing / tuning aid (mspmv_set_compact_tiles): up to how many tiles a call of the small tile shape runs the one-launch kernel
    behind its compact front end (0 = library default, > 0 = that many, < 0 = never).  y is bit for bit the same either way."""
    _setter("mspmv_set_compact_tiles", max_tiles == 0, int(max_tiles))


def clocked_bands(rows: int, cols: int, nnz: int, value_bytes: int):
    """(bands, columns per band) of the clock-scheduled form for a call of these sizes (mspmv_get_clocked_bands); (0, 0): not a candidate."""
    b = ctypes.c_int32(0); c = ctypes.c_int32(0)
    lib = load_library()
    lib.mspmv_get_clocked_bands.restype = ctypes.c_int
    lib.mspmv_get_clocked_bands.argtypes = [ctypes.c_int32] * 4 + [ctypes.POINTER(ctypes.c_int32)] * 2
    _check(lib.mspmv_get_clocked_bands(int(rows), int(cols), int(nnz), int(value_bytes), ctypes.byref(b), ctypes.byref(c)), "mspmv_get_clocked_bands")
    return b.value, c.value


def band_passes(rows: int, cols: int, nnz: int, value_bytes: int) -> int:
    """Passes a call of these sizes is offered (0: none); automatic setting: subject to the device-side verdicts."""
    n = ctypes.c_int32(0)
    _check(load_library().mspmv_get_band_passes(int(rows), int(cols), int(nnz), int(value_bytes), ctypes.byref(n)), "mspmv_get_band_passes")
    return n.value


def debug_band_windows(workspace, rows: int, nnz: int, value_bytes: int):
    """The 64 window verdicts the last automatic large-problem call left in the workspace (numpy int32[64])."""
    import numpy as np
    out = np.zeros(64, np.int32)
    tmp = workspace.buffer if hasattr(workspace, "buffer") else workspace
    _check(load_library().mspmv_debug_band_windows(ctypes.c_void_p(tmp.data_ptr()), int(rows), int(nnz), int(value_bytes),
                                                   out.ctypes.data_as(ctypes.c_void_p), None), "mspmv_debug_band_windows")
    return out


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


def sampled_check(A, x, y, samples: int = 1 << 16, seed: int = 0x5A3D, depth: Optional[int] = None) -> dict:
    """An untimed correctness witness for a benchmark record (NOT the parity tests: those are tests/ -m gpu against the oracle):
    `samples` seeded rows -- plus the first, the last and the longest row -- recomputed on the device in fp64 with torch gathers
    (val.double() * x.double()[col], rows up to 4096 nonzeros by index_add_, longer ones by torch.sum's tree) and compared with
    y under the stated bound of SURVEY 8d / DESIGN 3: |y - g| <= 2 (ceil(log2(len + 1)) + depth + 8) eps s, s = sum |val x|,
    eps = 2^-24 / 2^-53, empty rows exactly zero.  depth = serial_sum_depth of the call's shape unless given.
    Returns {"rows_checked", "worst_ratio" (max |y - g| / bound; < 1 passes), "violations", "longest_row"}."""
    import torch
    rows, nnz = int(A.rows), int(A.nnz)
    dev = A.values.device
    if rows == 0:
        return {"rows_checked": 0, "worst_ratio": 0.0, "violations": 0, "longest_row": 0}
    vb = A.values.element_size()
    if depth is None:
        depth = serial_sum_depth(rows, A.cols, nnz, vb)
    eps = 2.0 ** -24 if vb == 4 else 2.0 ** -53
    off = A.row_offsets.to(torch.int64)
    lens_all = off[1:] - off[:-1]
    g = torch.Generator(device="cpu"); g.manual_seed(int(seed))
    pick = torch.randint(0, rows, (min(int(samples), rows),), generator=g, dtype=torch.int64).to(dev)
    longest = int(torch.argmax(lens_all).item())
    pick = torch.unique(torch.cat([pick, torch.tensor([0, rows - 1, longest], dtype=torch.int64, device=dev)]))
    lens = lens_all[pick]
    start = off[pick]
    gold = torch.zeros(pick.numel(), dtype=torch.float64, device=dev)
    mag = torch.zeros_like(gold)
    xd = x.double()
    short = lens <= 4096
    if bool(short.any()):
        sl = lens[short]; st = start[short]
        total = int(sl.sum().item())
        if total > 0:
            seg = torch.repeat_interleave(torch.arange(sl.numel(), device=dev), sl)
            first = torch.cumsum(sl, 0) - sl
            j = st[seg] + (torch.arange(total, device=dev) - first[seg])
            p = A.values[j].double() * xd[A.column_indices[j].to(torch.int64)]
            gs = torch.zeros(sl.numel(), dtype=torch.float64, device=dev); ms = torch.zeros_like(gs)
            gs.index_add_(0, seg, p); ms.index_add_(0, seg, p.abs())
            gold[short] = gs; mag[short] = ms
    for k in torch.nonzero(~short).flatten().tolist():          # a handful of long rows: tree sums
        a, b = int(start[k].item()), int(start[k].item() + lens[k].item())
        gk, mk = 0.0, 0.0
        for c0 in range(a, b, 1 << 26):
            c1 = min(b, c0 + (1 << 26))
            p = A.values[c0:c1].double() * xd[A.column_indices[c0:c1].to(torch.int64)]
            gk += float(p.sum().item()); mk += float(p.abs().sum().item())
        gold[k] = gk; mag[k] = mk
    yy = y[pick].double()
    c = 2.0 * (torch.ceil(torch.log2(lens.double() + 1.0)) + float(depth) + 8.0)
    bound = c * eps * mag
    err = (yy - gold).abs()
    bad_empty = (lens == 0) & (y[pick] != 0)
    ratio = torch.where(bound > 0, err / bound, torch.where(err == 0, torch.zeros_like(err), torch.full_like(err, float("inf"))))
    ratio = torch.where(torch.isfinite(yy), ratio, torch.full_like(ratio, float("inf")))
    ratio = torch.where(bad_empty, torch.full_like(ratio, float("inf")), ratio)
    worst = float(ratio.max().item())
    return {"rows_checked": int(pick.numel()), "worst_ratio": round(worst, 4) if worst != float("inf") else "inf",
            "violations": int((ratio > 1.0).sum().item()), "longest_row": int(lens_all[longest].item()), "depth_term": int(depth)}
