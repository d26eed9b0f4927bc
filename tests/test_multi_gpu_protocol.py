"""Multi-GPU merge partitioning on CPU: the partitioner, the
shard-as-local-CSR construction, the one carry exchange and the carry
application, (a) simulated for 1..8 parts in one process and (b) for real
with torch.distributed (gloo, world_size 2, 127.0.0.1).  The per-shard SpMV is
the oracle here (no GPU); on the GPU box the same ShardedCsrMV class calls the
HIP kernels through the C ABI (tests/test_gpu_parity.py::test_sharded_*)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from merge_spmv_amd import multi_gpu as MG


def make_matrix(kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        lens = rng.integers(0, 7, 400)
    elif kind == "giant_middle":          # one row spanning several parts
        lens = rng.integers(0, 4, 300); lens[150] = 5000
    elif kind == "giant_first":
        lens = rng.integers(0, 3, 200); lens[0] = 3000
    elif kind == "giant_last":
        lens = rng.integers(0, 3, 200); lens[-1] = 3000
    elif kind == "empty_runs":
        lens = np.zeros(1000, np.int64); lens[::97] = 40
    elif kind == "all_empty":
        lens = np.zeros(64, np.int64)
    elif kind == "single_row":
        lens = np.array([777])
    rows = lens.size
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1]); cols = 50
    col = rng.integers(0, cols, nnz).astype(np.int32)
    val = rng.integers(-3, 4, nnz).astype(np.float64)     # exact arithmetic: any summation order agrees
    return O.Csr(rows, cols, off.astype(np.int32), col, val), rng.integers(-2, 3, cols).astype(np.float64)


def oracle_local_spmv(shard, x, y_local):
    csr = O.Csr(shard.local_rows, shard.num_cols, shard.row_offsets.numpy(), shard.column_indices.numpy(),
                shard.values.numpy())
    y_local.copy_(torch.from_numpy(O.spmv_gold(csr, x.numpy())))


KINDS = ["uniform", "giant_middle", "giant_first", "giant_last", "empty_runs", "all_empty", "single_row"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("parts", [1, 2, 3, 4, 8])
def test_simulated_exchange(kind, parts):
    csr, x = make_matrix(kind)
    gold = O.spmv_gold(csr, x)
    xt = torch.from_numpy(x)
    ops = [MG.ShardedCsrMV(MG.shard_from_host_csr(csr.row_offsets, csr.column_indices, csr.values, csr.cols, g, parts,
                                                  device="cpu"), local_spmv=oracle_local_spmv) for g in range(parts)]
    # phase 1: every part's local SpMV (its last local row is the carry)
    carries = torch.zeros(parts, dtype=torch.float64)
    for g, op in enumerate(ops):
        s = op.shard
        op.local_spmv(s, xt, op.y_local)
        carries[g] = op.y_local[s.local_rows - 1]
        assert s.local_rows + s.local_nnz <= -(-(csr.rows + csr.nnz) // parts) + 1     # equal swaths (+ the open row)
    # phase 2: "all-gather" + owner adds
    y = np.full(csr.rows, np.nan)
    for g, op in enumerate(ops):
        op.carries.copy_(carries)
        op._apply_carries()
        s = op.shard
        y[int(s.row_split[g]): int(s.row_split[g + 1])] = op.y_local[: s.owned_rows].numpy()
    assert np.array_equal(y, gold)
    # the last part's carry belongs to row `rows` and must be zero / dropped
    assert carries[-1] == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kind, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        csr, x = make_matrix(kind)
        shard = MG.shard_from_host_csr(csr.row_offsets, csr.column_indices, csr.values, csr.cols, rank, world, device="cpu")
        op = MG.ShardedCsrMV(shard, local_spmv=oracle_local_spmv)
        y_owned = op(torch.from_numpy(x))                # local SpMV + ONE all-gather + carry application
        out[rank] = (int(shard.row_split[rank]), y_owned.numpy().copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["uniform", "giant_middle", "giant_first", "single_row"])
def test_two_ranks_gloo(kind):
    world = 2
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, out), nprocs=world, join=True)
    csr, x = make_matrix(kind)
    gold = O.spmv_gold(csr, x)
    y = np.full(csr.rows, np.nan)
    for rank in range(world):
        start, part = out[rank]
        y[start: start + part.size] = part
    assert np.array_equal(y, gold)


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_uniform_shard_matches_sharding_the_whole_matrix(parts):
    """bench.py builds each rank's swath of the C2 matrix directly (uniform_shard); it must
    be the same shard as cutting the fully generated matrix."""
    from merge_spmv_amd import generators as G
    rows, cols, npr = 1000, 777, 5
    A = G.uniform_csr(rows, cols, npr, dtype=torch.float64, device="cpu")
    for g in range(parts):
        want = MG.shard_from_host_csr(A.row_offsets.numpy(), A.column_indices.numpy(), A.values.numpy(), cols, g, parts,
                                      device="cpu")
        got = MG.uniform_shard(rows, cols, npr, g, parts, torch.float64, device="cpu")
        assert torch.equal(got.row_offsets, want.row_offsets)
        assert torch.equal(got.column_indices, want.column_indices)
        assert torch.equal(got.values, want.values)
        assert np.array_equal(got.row_split, want.row_split) and np.array_equal(got.nz_split, want.nz_split)


def _worker_iter(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        csr, x = make_square()
        shard = MG.shard_from_host_csr(csr.row_offsets, csr.column_indices, csr.values, csr.cols, rank, world, device="cpu")
        op = MG.ShardedCsrMV(shard, local_spmv=oracle_local_spmv)
        xt = torch.from_numpy(x)
        for _ in range(3):                                  # x <- A x, three times
            xt = op.allgather_rows(op(xt)).clone()
        out[rank] = xt.numpy().copy()
    finally:
        dist.destroy_process_group()


def make_square():
    rng = np.random.default_rng(5)
    n = 300
    lens = rng.integers(0, 5, n); lens[7] = 900
    off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=off[1:])
    col = rng.integers(0, n, int(off[-1])).astype(np.int32)
    val = rng.integers(-1, 2, int(off[-1])).astype(np.float64)
    return O.Csr(n, n, off.astype(np.int32), col, val), rng.integers(-1, 2, n).astype(np.float64)


def test_iterated_spmv_with_row_allgather():
    """N3: y -> x redistribution; three SpMV iterations on 2 ranks equal the serial result."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker_iter, args=(world, _free_port(), out), nprocs=world, join=True)
    csr, x = make_square()
    want = x
    for _ in range(3):
        want = O.spmv_gold(csr, want)
    for rank in range(world):
        assert np.array_equal(out[rank], want)
