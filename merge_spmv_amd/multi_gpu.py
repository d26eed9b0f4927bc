"""Multi-GPU merge-partitioned CsrMV -- one process per GPU.

New design (the reference is single-device; its README.md:5 only claims the
decomposition partitions recursively): the global merge path is cut at G
equally spaced diagonals exactly as cpu_spmv.cpp:305-321 cuts it per OpenMP
thread; each rank runs the ordinary single-GPU CsrMV (include/mspmv.h) on its
swath, seen as a local CSR matrix with one extra "open" last row whose y IS
the rank's carry; ONE collective (an all-gather of G scalars over RCCL/xGMI)
exchanges the carries; each rank adds the carries addressed to its first row
(cpu_spmv.cpp:348-352 lifted to ranks).  x is replicated, y stays row-sharded.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import load_library, _check, CsrMVWorkspace, DeviceSpmv, csrmv, _stream_handle, _value_bytes, MspmvError

EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_PEER, EXCHANGE_IPC = 0, 1, 2, 3


def partition(row_offsets_i64: np.ndarray, parts: int):
    """(row_split[parts+1], nz_split[parts+1]) -- mspmv_mg_partition."""
    off = np.ascontiguousarray(row_offsets_i64, dtype=np.int64)
    rows = off.size - 1
    nnz = int(off[-1])
    row_split = np.zeros(parts + 1, dtype=np.int64)
    nz_split = np.zeros(parts + 1, dtype=np.int64)
    _check(load_library().mspmv_mg_partition(off.ctypes.data_as(ctypes.c_void_p), rows, nnz, int(parts),
                                             row_split.ctypes.data_as(ctypes.c_void_p),
                                             nz_split.ctypes.data_as(ctypes.c_void_p)), "mspmv_mg_partition")
    return row_split, nz_split


def local_offsets(row_offsets_i64: np.ndarray, row_begin: int, row_end: int, nz_begin: int, nz_end: int) -> np.ndarray:
    """int32 row_offsets of one part's local CSR (local_rows + 1 entries,
    local_rows = row_end - row_begin + 1) -- mspmv_mg_local_offsets."""
    off = np.ascontiguousarray(row_offsets_i64, dtype=np.int64)
    out = np.zeros(int(row_end - row_begin) + 2, dtype=np.int32)
    _check(load_library().mspmv_mg_local_offsets(off.ctypes.data_as(ctypes.c_void_p), off.size - 1, int(row_begin),
                                                 int(row_end), int(nz_begin), int(nz_end),
                                                 out.ctypes.data_as(ctypes.c_void_p)), "mspmv_mg_local_offsets")
    return out


class _MgInfo(ctypes.Structure):
    _fields_ = [("parts", ctypes.c_int32), ("local_parts", ctypes.c_int32), ("exchange", ctypes.c_int32),
                ("value_bytes", ctypes.c_int32), ("replicas", ctypes.c_int32), ("hot_parts", ctypes.c_int32),
                ("rows", ctypes.c_int64), ("cols", ctypes.c_int64), ("carry_bytes_per_step", ctypes.c_uint64),
                ("allgather_bytes_per_step", ctypes.c_uint64), ("steps", ctypes.c_uint64)]


class _DeviceArray:
    """plan-owned device memory seen by torch (torch.as_tensor reads __cuda_array_interface__: no copy)"""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def unique_id() -> bytes:
    """128-byte token (an ncclUniqueId) ONE process makes and every process of a multi-process plan passes."""
    buf = ctypes.create_string_buffer(128)
    _check(load_library().mspmv_mg_unique_id(buf), "mspmv_mg_unique_id")
    return buf.raw


class MgPlan:
    """Thin caller of the C multi-GPU operator (include/mspmv.h: mspmv_mg_plan_*): all the per-step work --
    the parts' CsrMV launches, the one carry exchange (peer reads or one RCCL all-gather) and the owners'
    adds -- happens below the C ABI in ONE call, `csrmv()`.

    part_ids / device_ids: the parts THIS process drives (all of them, or one per process under
    torch.distributed.run; then `id128` = unique_id() of one rank, shipped to the others)."""

    def __init__(self, row_split, nz_split, num_cols: int, dtype, part_ids, device_ids, exchange: int = EXCHANGE_AUTO,
                 id128: Optional[bytes] = None):
        import torch
        self.torch = torch
        self.dtype = dtype
        self.vb = 4 if dtype == torch.float32 else 8
        self.row_split = np.ascontiguousarray(row_split, dtype=np.int64)
        self.nz_split = np.ascontiguousarray(nz_split, dtype=np.int64)
        self.parts = self.row_split.size - 1
        self.part_ids = [int(p) for p in part_ids]
        self.device_ids = [int(d) for d in device_ids]
        self.num_cols = int(num_cols)
        ids = np.asarray(self.part_ids, dtype=np.int32); devs = np.asarray(self.device_ids, dtype=np.int32)
        self._handle = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(id128, 128) if id128 is not None else None
        _check(load_library().mspmv_mg_plan_create(ctypes.byref(self._handle), self.parts, len(self.part_ids),
                                                   ids.ctypes.data_as(ctypes.c_void_p), devs.ctypes.data_as(ctypes.c_void_p),
                                                   self.row_split.ctypes.data_as(ctypes.c_void_p),
                                                   self.nz_split.ctypes.data_as(ctypes.c_void_p), self.num_cols, self.vb,
                                                   int(exchange), idbuf), "mspmv_mg_plan_create")
        self._keep = {}

    def local_rows(self, i: int) -> int:
        g = self.part_ids[i]
        return int(self.row_split[g + 1] - self.row_split[g]) + 1

    def owned_rows(self, i: int) -> int:
        return self.local_rows(i) - 1

    def local_nnz(self, i: int) -> int:
        g = self.part_ids[i]
        return int(self.nz_split[g + 1] - self.nz_split[g])

    def set_part(self, i: int, values, local_row_offsets, column_indices):
        """attach local part i's CSR (tensors on that part's device; kept alive by the plan object)"""
        torch = self.torch
        dev = torch.device("cuda", self.device_ids[i])
        if local_row_offsets.dtype != torch.int32 or local_row_offsets.device != dev or not local_row_offsets.is_contiguous() \
                or local_row_offsets.numel() != self.local_rows(i) + 1:
            raise MspmvError("set_part: local_row_offsets must be a contiguous int32 tensor with local_rows + 1 entries on the part's device")
        n = self.local_nnz(i)
        if n > 0 and (values.dtype != self.dtype or values.device != dev or not values.is_contiguous() or values.numel() != n or
                      column_indices.dtype != torch.int32 or column_indices.device != dev or
                      not column_indices.is_contiguous() or column_indices.numel() != n):
            raise MspmvError("set_part: values / column_indices must be contiguous tensors of the part's nonzero count on its device")
        self._keep[i] = (values, local_row_offsets, column_indices)
        _check(load_library().mspmv_mg_plan_set_part(self._handle, i, ctypes.c_void_p(values.data_ptr() if n else 0),
                                                     ctypes.c_void_p(local_row_offsets.data_ptr()),
                                                     ctypes.c_void_p(column_indices.data_ptr() if n else 0)),
               "mspmv_mg_plan_set_part")

    def _view(self, ptr, count, i):
        torch = self.torch
        if count == 0:
            return torch.empty(0, dtype=self.dtype, device=torch.device("cuda", self.device_ids[i]))
        t = torch.as_tensor(_DeviceArray(ptr, count, "<f4" if self.vb == 4 else "<f8"), device=torch.device("cuda", self.device_ids[i]))
        return t

    def x(self, i: int):
        """the replica of x on local part i's device (num_cols entries), as a tensor aliasing plan memory"""
        return self._view(load_library().mspmv_mg_plan_x(self._handle, i), self.num_cols, i)

    def y(self, i: int, with_open_row: bool = False):
        """local part i's owned rows of y (aliasing plan memory)"""
        n = self.local_rows(i) if with_open_row else self.owned_rows(i)
        return self._view(load_library().mspmv_mg_plan_y(self._handle, i), n, i)

    def stream(self, i: int) -> int:
        return int(load_library().mspmv_mg_plan_stream(self._handle, i) or 0)

    def info(self) -> dict:
        info = _MgInfo()
        _check(load_library().mspmv_mg_plan_info(self._handle, ctypes.byref(info)), "mspmv_mg_plan_info")
        return {name: getattr(info, name) for name, _ in _MgInfo._fields_}

    def ipc_export(self) -> bytes:
        """this process's hipIpc handles (x replicas, mailbox blocks) as one blob (mspmv_mg_plan_ipc_export)"""
        size = ctypes.c_size_t(0)
        _check(load_library().mspmv_mg_plan_ipc_export(self._handle, None, ctypes.byref(size)), "mspmv_mg_plan_ipc_export")
        buf = ctypes.create_string_buffer(size.value)
        _check(load_library().mspmv_mg_plan_ipc_export(self._handle, buf, ctypes.byref(size)), "mspmv_mg_plan_ipc_export")
        return buf.raw[: size.value]

    def ipc_import(self, blobs):
        """every process's blob (its own included, any order): opens the peers' memory (mspmv_mg_plan_ipc_import)"""
        stride = max(len(b) for b in blobs)
        flat = b"".join(b + b"\0" * (stride - len(b)) for b in blobs)
        _check(load_library().mspmv_mg_plan_ipc_import(self._handle, flat, len(blobs), stride), "mspmv_mg_plan_ipc_import")

    def ipc_connect(self, group=None):
        """export, all-gather the blobs over torch.distributed (any backend), import: one call per rank"""
        import torch.distributed as dist
        mine = self.ipc_export()
        blobs = [None] * dist.get_world_size(group)
        dist.all_gather_object(blobs, mine, group=group)
        self.ipc_import(blobs)

    def hot_columns(self, enable=True):
        """the parts' hot-column plans (mspmv_mg_plan_hot_columns): True / 1 always, False / 0 never, -1 automatic (the default of a new
        plan: decided per part when its matrix is attached)"""
        _check(load_library().mspmv_mg_plan_hot_columns(self._handle, -1 if (enable is not True and enable is not False and int(enable) < 0) else int(bool(enable))),
               "mspmv_mg_plan_hot_columns")

    def exchange_ms(self, local_index: int = 0) -> float:
        """milliseconds the exchange of the last step took on a local part (mspmv_mg_plan_exchange_ms; synchronises)"""
        ms = ctypes.c_float(0)
        lib = load_library()
        lib.mspmv_mg_plan_exchange_ms.restype = ctypes.c_int
        lib.mspmv_mg_plan_exchange_ms.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        _check(lib.mspmv_mg_plan_exchange_ms(self._handle, int(local_index), ctypes.byref(ms)), "mspmv_mg_plan_exchange_ms")
        return float(ms.value)

    def csrmv(self):
        _check(load_library().mspmv_mg_csrmv(self._handle), "mspmv_mg_csrmv")

    def allgather_rows(self):
        _check(load_library().mspmv_mg_allgather_rows(self._handle), "mspmv_mg_allgather_rows")

    def synchronize(self):
        _check(load_library().mspmv_mg_synchronize(self._handle), "mspmv_mg_synchronize")

    def close(self):
        if self._handle:
            load_library().mspmv_mg_plan_destroy(self._handle)
            self._handle = ctypes.c_void_p()
            self._keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class Shard:
    """One rank's swath of the global matrix, as an ordinary local CSR."""
    part: int
    parts: int
    row_split: np.ndarray       # int64 [parts+1] (host)
    nz_split: np.ndarray        # int64 [parts+1] (host)
    row_offsets: "object"       # torch int32 [local_rows+1] (device)
    column_indices: "object"    # torch int32 [local_nnz]
    values: "object"            # torch f32/f64 [local_nnz]
    num_cols: int

    @property
    def local_rows(self) -> int:
        return int(self.row_split[self.part + 1] - self.row_split[self.part]) + 1

    @property
    def owned_rows(self) -> int:
        return self.local_rows - 1

    @property
    def local_nnz(self) -> int:
        return int(self.nz_split[self.part + 1] - self.nz_split[self.part])


class ShardedCsrMV:
    """y_owned = (A x)[row_split[p] : row_split[p+1]] on rank p.

    `group` is a torch.distributed process group (backend "nccl" == RCCL on
    ROCm; "gloo" works for the CPU-side protocol tests with `local_spmv`
    overridden).  With parts == 1 no collective is issued.
    """

    def __init__(self, shard: Shard, group=None, local_spmv=None):
        import torch
        self.shard = shard
        self.group = group
        self.torch = torch
        self.local_spmv = local_spmv
        dev = shard.values.device
        self.y_local = torch.empty(shard.local_rows, dtype=shard.values.dtype, device=dev)
        self.carries = torch.zeros(shard.parts, dtype=shard.values.dtype, device=dev)
        self.workspace = None
        if local_spmv is None:
            # the part's matrix is fixed for the life of this operator: its tile coordinates are found once
            self.workspace = CsrMVWorkspace(shard.local_rows, shard.local_nnz, shard.values.dtype, device=dev)
            self.workspace.prepare(shard.row_offsets)

    def __call__(self, x):
        torch, s = self.torch, self.shard
        if self.local_spmv is not None:
            self.local_spmv(s, x, self.y_local)
        else:
            csrmv(s.values, s.row_offsets, s.column_indices, x, y=self.y_local, num_cols=s.num_cols, workspace=self.workspace)
        if s.parts > 1:
            import torch.distributed as dist
            # the ONE exchange: every rank contributes its open-row partial
            dist.all_gather_into_tensor(self.carries, self.y_local[s.local_rows - 1:], group=self.group)
            self._apply_carries()
        return self.y_local[: s.owned_rows]

    def allgather_rows(self, y_owned, out=None):
        """turn the row-sharded result into a replicated vector (the x of
        the next SpMV in an iterative solver): one all-gather of fixed-size (padded) row
        blocks, then each rank's block is copied to its row range.  This is the step whose
        cost is set by xGMI bandwidth (rows * sizeof V bytes per GPU), unlike the 8-byte
        carry exchange."""
        torch, s = self.torch, self.shard
        rows_total = int(s.row_split[-1])
        if out is None:
            out = torch.empty(rows_total, dtype=y_owned.dtype, device=y_owned.device)
        if s.parts == 1:
            out.copy_(y_owned)
            return out
        import torch.distributed as dist
        width = int((s.row_split[1:] - s.row_split[:-1]).max())
        send = torch.zeros(width, dtype=y_owned.dtype, device=y_owned.device)
        send[: s.owned_rows] = y_owned
        recv = torch.empty(s.parts * width, dtype=y_owned.dtype, device=y_owned.device)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        for g in range(s.parts):
            a, b = int(s.row_split[g]), int(s.row_split[g + 1])
            out[a:b] = recv[g * width: g * width + (b - a)]
        return out

    def _apply_carries(self):
        s = self.shard
        if self.y_local.is_cuda:
            _check(load_library().mspmv_mg_apply_carries(
                ctypes.c_void_p(self.y_local.data_ptr()), ctypes.c_void_p(self.carries.data_ptr()),
                s.row_split.ctypes.data_as(ctypes.c_void_p), int(s.parts), int(s.part),
                _value_bytes(self.y_local), _stream_handle(None)), "mspmv_mg_apply_carries")
        else:
            # host mirror of mg_apply_kernel for the gloo protocol tests
            if s.row_split[s.part + 1] > s.row_split[s.part]:
                for j in range(s.part):
                    if s.row_split[j + 1] == s.row_split[s.part]:
                        self.y_local[0] += self.carries[j]


def shard_from_host_csr(row_offsets: np.ndarray, column_indices: np.ndarray, values: np.ndarray, num_cols: int,
                        part: int, parts: int, device="cuda") -> Shard:
    """Cut a host CSR (any integer offset width) and upload rank `part`'s swath."""
    import torch
    off = np.ascontiguousarray(row_offsets, dtype=np.int64)
    row_split, nz_split = partition(off, parts)
    lo = local_offsets(off, row_split[part], row_split[part + 1], nz_split[part], nz_split[part + 1])
    a, b = int(nz_split[part]), int(nz_split[part + 1])
    return Shard(part, parts, row_split, nz_split,
                 torch.from_numpy(lo).to(device),
                 torch.from_numpy(np.ascontiguousarray(column_indices[a:b], dtype=np.int32)).to(device),
                 torch.from_numpy(np.ascontiguousarray(values[a:b])).to(device), int(num_cols))


def uniform_shard(rows: int, cols: int, nnz_per_row: int, part: int, parts: int, dtype, device="cuda") -> Shard:
    """Rank `part`'s swath of generators.uniform_csr(rows, cols, nnz_per_row) built directly
    on `device` without ever materialising the whole matrix: the global row offsets of that
    generator are r * nnz_per_row, the generator's counters are global, so a rank only
    generates the rows its swath touches (plus the row its right boundary cuts)."""
    import torch
    from . import generators as G
    off = np.arange(rows + 1, dtype=np.int64) * nnz_per_row
    row_split, nz_split = partition(off, parts)
    r0, r1 = int(row_split[part]), int(row_split[part + 1])
    r_hi = min(r1 + 1, rows)
    full = G.uniform_csr(rows, cols, nnz_per_row, dtype=dtype, device=device, row_lo=r0, row_hi=r_hi)
    a = int(nz_split[part]) - r0 * nnz_per_row
    b = int(nz_split[part + 1]) - r0 * nnz_per_row
    lo = local_offsets(off, r0, r1, int(nz_split[part]), int(nz_split[part + 1]))
    return Shard(part, parts, row_split, nz_split, torch.from_numpy(lo).to(device),
                 full.column_indices[a:b].contiguous(), full.values[a:b].contiguous(), int(cols))


def rmat_shard(scale: int, edges: int, part: int, parts: int, dtype, device="cuda", seed: Optional[int] = None,
               return_global_offsets: bool = False, dist_group=None, use_dist: bool = False):
    """Rank `part`'s swath of generators.rmat_csr(scale, edges) (C5: scale 26, 2e9 edges, fp64), built on
    `device` without materialising the whole matrix: one pass over the edge ids counts the global row
    lengths (so every rank finds the same partition), a second keeps only the edges of the rows the swath
    touches (plus the row its right boundary cuts).  The matrix is independent of `parts`.
    use_dist: the counting pass is split over the ranks of torch.distributed (each hashes 1/parts of the
    edge ids) and summed with one all-reduce -- set-up work, not part of any timed region."""
    import torch
    from . import generators as G
    seed = G.SEED_C5 if seed is None else seed
    n = 1 << scale
    if use_dist and parts > 1:
        import torch.distributed as dist
        e0 = edges * part // parts; e1 = edges * (part + 1) // parts
        counts = G.rmat_row_counts(scale, e1 - e0, device, seed, edge_begin=e0)
        if dist.get_backend(dist_group) == "gloo":
            c = counts.cpu(); dist.all_reduce(c, group=dist_group); counts = c.to(device)
        else:
            dist.all_reduce(counts, group=dist_group)
    else:
        counts = G.rmat_row_counts(scale, edges, device, seed)
    off_dev = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=off_dev[1:])
    off = off_dev.cpu().numpy()
    del counts, off_dev
    row_split, nz_split = partition(off, parts)
    r0, r1 = int(row_split[part]), int(row_split[part + 1])
    r_hi = min(r1 + 1, n)
    full = G.rmat_csr(scale, edges, dtype=dtype, device=device, seed=seed, row_lo=r0, row_hi=r_hi)
    a = int(nz_split[part]) - int(off[r0])
    b = int(nz_split[part + 1]) - int(off[r0])
    lo = local_offsets(off, r0, r1, int(nz_split[part]), int(nz_split[part + 1]))
    shard = Shard(part, parts, row_split, nz_split, torch.from_numpy(lo).to(device),
                  full.column_indices[a:b].contiguous(), full.values[a:b].contiguous(), n)
    return (shard, off) if return_global_offsets else shard
