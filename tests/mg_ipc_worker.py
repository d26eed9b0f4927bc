"""Worker of tests/test_mg_plan.py::test_ipc_backend_two_processes_sharing_the_device (run under torch.distributed.run, gloo):
every rank drives ONE part of the same matrix on cuda:0 through the C operator's IPC backend -- carries written into the
owner's mailbox through hipIpc-opened memory, step tags instead of collectives -- for several steps with the SAME x (the
producer may run ahead: exercises the two-slot credit), then iterated SpMV with the row all-gather.  Rank 0 checks every
row against the oracle and prints IPC-OK."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
import merge_spmv_amd as M
from merge_spmv_amd import multi_gpu as MG
from oracle import oracle as O

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
M.load_library()
kind, prec = sys.argv[1], sys.argv[2]
dtype, tdt = (np.float32, torch.float32) if prec == "f32" else (np.float64, torch.float64)
rng = np.random.default_rng(1234)                      # the same matrix on every rank
rows = 30000
if kind == "giant":                                     # one row spanning every part: every part but the last sends a carry
    lens = rng.integers(0, 4, rows); lens[rows // 2] = 900000
elif kind == "short":
    lens = rng.integers(0, 12, rows)
else:                                                   # empty parts in the middle of a giant row + ordinary rows
    lens = np.zeros(rows, np.int64); lens[10] = 700000; lens[rows - 5:] = 7
off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
nnz = int(off[-1])
col = rng.integers(0, rows, nnz).astype(np.int32)
val = (rng.uniform(-1, 1, nnz) * 0.05).astype(dtype)
x0 = rng.uniform(-1, 1, rows).astype(dtype)
csr = O.Csr(rows, rows, off.astype(np.int32), col, val)
row_split, nz_split = MG.partition(off, world)
plan = MG.MgPlan(row_split, nz_split, rows, tdt, [rank], [0], exchange=MG.EXCHANGE_IPC)
lo = MG.local_offsets(off, row_split[rank], row_split[rank + 1], nz_split[rank], nz_split[rank + 1])
a, b = int(nz_split[rank]), int(nz_split[rank + 1])
plan.set_part(0, torch.from_numpy(val[a:b].copy()).cuda(), torch.from_numpy(lo).cuda(), torch.from_numpy(col[a:b].copy()).cuda())
plan.ipc_connect()
assert plan.info()["exchange"] == MG.EXCHANGE_IPC
plan.x(0).copy_(torch.from_numpy(x0).cuda()); torch.cuda.synchronize()
dist.barrier()


def gather_y():
    mine = plan.y(0).cpu().numpy()
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    return np.concatenate(parts)


g, s = O.spmv_gold_acc64(csr, x0)
first = None
for step in range(6):                                   # same x: nothing but the credit protocol keeps the ranks within two steps
    plan.csrmv()
    if step % 2 == 1:
        plan.synchronize()
        y = gather_y()
        ok, worst = O.strict_check(csr, y, g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=world))
        assert ok, (kind, prec, step, worst)
        first = y if first is None else first
        assert np.array_equal(y, first)                 # bitwise repeatable
plan.synchronize()
# iterated SpMV: x <- y through the peers' opened replicas
xh = x0.copy()
for step in range(3):
    plan.csrmv(); plan.allgather_rows(); plan.synchronize()
    dist.barrier()                                      # (only so that rank 0 reads x after every rank's pushes of this step; the plan itself needs no barrier)
    got = plan.x(0).cpu().numpy()
    gg, ss = O.spmv_gold_acc64(csr, xh)
    ok, worst = O.strict_check(csr, got, gg, ss, M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=world))
    assert ok, ("iterated", kind, prec, step, worst)
    xh = got
    dist.barrier()
plan.synchronize()
plan.close()
dist.barrier()
if rank == 0:
    print("IPC-OK", kind, prec, world, flush=True)
dist.destroy_process_group()
