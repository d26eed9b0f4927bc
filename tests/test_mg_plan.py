"""The multi-GPU operator below the C ABI (include/mspmv.h: mspmv_mg_plan_*), SURVEY.md 8e / 8f N3.

CPU (-m "not gpu"): argument checking of the entry points that need no device.
GPU (-m gpu): the single-process form with G parts on ONE device (peer exchange: events + one kernel reading
the carries out of the other parts' memory), the RCCL backend with a one-rank communicator (the whole RCCL
path -- dlopen, communicator, all-gather, broadcast -- on a 1-GPU box), the y -> x redistribution, and
BASELINE config 5 at full size: fp64 R-MAT scale 26, 2e9 edges, cut into 8 parts, EVERY row checked against
the oracle."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT
import merge_spmv_amd as M
from merge_spmv_amd import multi_gpu as MG
from oracle import oracle as O

torch = pytest.importorskip("torch")


def test_plan_entry_points_reject_bad_arguments():
    lib = M.load_library()
    h = ctypes.c_void_p()
    split = (ctypes.c_int64 * 3)(0, 5, 10)
    ids = (ctypes.c_int32 * 2)(0, 1)
    devs = (ctypes.c_int32 * 2)(0, 0)
    # no plan pointer / no parts / too many parts / value size / exchange kind
    assert lib.mspmv_mg_plan_create(None, 2, 2, ids, devs, split, split, 10, 8, 0, None) == 1
    assert lib.mspmv_mg_plan_create(ctypes.byref(h), 0, 0, ids, devs, split, split, 10, 8, 0, None) == 1
    assert lib.mspmv_mg_plan_create(ctypes.byref(h), 65, 2, ids, devs, split, split, 10, 8, 0, None) == 1
    assert lib.mspmv_mg_plan_create(ctypes.byref(h), 2, 2, ids, devs, split, split, 10, 3, 0, None) == 1
    assert lib.mspmv_mg_plan_create(ctypes.byref(h), 2, 2, ids, devs, split, split, 10, 8, 7, None) == 1
    assert lib.mspmv_mg_plan_create(ctypes.byref(h), 2, 3, ids, devs, split, split, 10, 8, 0, None) == 1
    for fn in (lib.mspmv_mg_csrmv, lib.mspmv_mg_allgather_rows, lib.mspmv_mg_synchronize):
        assert fn(None) == 1
    assert lib.mspmv_mg_plan_destroy(None) == 0
    assert lib.mspmv_mg_plan_x(None, 0) is None and lib.mspmv_mg_plan_y(None, 0) is None
    assert lib.mspmv_mg_unique_id(None) == 1
    assert len(MG.unique_id()) == 128 and MG.unique_id() != MG.unique_id()


# ------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def _matrix(kind, seed, dtype, square=True):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        lens = rng.integers(0, 9, 30000)
    elif kind == "giant_middle":          # one row spanning several parts
        lens = rng.integers(0, 4, 20000); lens[10000] = 400000
    elif kind == "giant_first":
        lens = rng.integers(0, 3, 9000); lens[0] = 150000
    elif kind == "giant_last":
        lens = rng.integers(0, 3, 9000); lens[-1] = 150000
    elif kind == "empty_runs":
        lens = np.zeros(50000, np.int64); lens[::97] = 300
    elif kind == "all_empty":
        lens = np.zeros(64, np.int64)
    elif kind == "single_row":
        lens = np.array([77777])
    elif kind == "power_law":
        lens = np.minimum((rng.pareto(1.1, 40000) * 2).astype(np.int64), 60000)
    rows = lens.size
    cols = rows if square else 1234
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    csr = O.Csr(rows, cols, off.astype(np.int32), rng.integers(0, cols, nnz).astype(np.int32),
                rng.uniform(-1, 1, nnz).astype(dtype))
    return csr, rng.uniform(-1, 1, cols).astype(dtype)


def _plan_on_one_device(csr, parts, tdt, exchange):
    off = csr.row_offsets.astype(np.int64)
    row_split, nz_split = MG.partition(off, parts)
    plan = MG.MgPlan(row_split, nz_split, csr.cols, tdt, list(range(parts)), [0] * parts, exchange=exchange)
    for g in range(parts):
        lo = MG.local_offsets(off, row_split[g], row_split[g + 1], nz_split[g], nz_split[g + 1])
        a, b = int(nz_split[g]), int(nz_split[g + 1])
        plan.set_part(g, torch.from_numpy(csr.values[a:b].copy()).cuda(), torch.from_numpy(lo).cuda(),
                      torch.from_numpy(csr.column_indices[a:b].copy()).cuda())
    return plan, row_split


def _gather_y(plan, row_split, rows, dtype):
    y = np.full(rows, np.nan, dtype)
    for g in range(plan.parts):
        y[int(row_split[g]): int(row_split[g + 1])] = plan.y(g).cpu().numpy()
    return y


KINDS = ["uniform", "giant_middle", "giant_first", "giant_last", "empty_runs", "all_empty", "single_row", "power_law"]


@gpu
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("parts", [1, 2, 3, 8])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_single_process_plan_peer_exchange(kind, parts, prec):
    """G parts on one device: every part's CsrMV + the peer carry exchange + the owners' adds in ONE C call;
    y (assembled from the parts' owned rows) within the strict tolerance of the oracle, and bitwise the same
    on every repetition."""
    dtype, tdt = (np.float32, torch.float32) if prec == "f32" else (np.float64, torch.float64)
    csr, x = _matrix(kind, 3 + parts, dtype)
    plan, row_split = _plan_on_one_device(csr, parts, tdt, MG.EXCHANGE_PEER)
    try:
        info = plan.info()
        assert info["parts"] == parts and info["local_parts"] == parts and info["replicas"] == 1
        assert info["exchange"] == MG.EXCHANGE_PEER and info["carry_bytes_per_step"] == parts * dtype().itemsize
        plan.x(0).copy_(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        plan.csrmv(); plan.synchronize()
        y = _gather_y(plan, row_split, csr.rows, dtype)
        g, s = O.spmv_gold_acc64(csr, x)
        ok, worst = O.strict_check(csr, y, g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=parts))
        assert ok, (kind, parts, prec, worst)
        for _ in range(3):
            plan.csrmv(); plan.synchronize()
            assert np.array_equal(_gather_y(plan, row_split, csr.rows, dtype), y)
        assert plan.info()["steps"] == 4
        # every part's columns renumbered by reference count (mspmv_mg_plan_hot_columns): the same bits
        plan.hot_columns(True)
        for _ in range(2):
            plan.csrmv(); plan.synchronize()
            assert np.array_equal(_gather_y(plan, row_split, csr.rows, dtype), y)
        plan.hot_columns(False)
        plan.csrmv(); plan.synchronize()
        assert np.array_equal(_gather_y(plan, row_split, csr.rows, dtype), y)
    finally:
        plan.close()


@gpu
@pytest.mark.parametrize("parts", [1, 4, 8])
def test_plan_iterated_spmv_with_row_allgather(parts):
    """x_{k+1} = A x_k with the y -> x redistribution done by the plan (every part pushes
    its owned rows straight into the replica of x; no padding, no host loop), 4 steps against the host
    iteration."""
    csr, x0 = _matrix("uniform", 21, np.float64)
    csr.values[:] *= 0.3                                   # keep the iterates bounded
    plan, row_split = _plan_on_one_device(csr, parts, torch.float64, MG.EXCHANGE_PEER)
    try:
        info = plan.info()
        assert info["allgather_bytes_per_step"] == 0       # one replica: nothing crosses a link on this box
        plan.x(0).copy_(torch.from_numpy(x0).cuda())
        torch.cuda.synchronize()
        xh = x0.copy()
        for step in range(4):
            plan.csrmv()
            plan.allgather_rows()
            plan.synchronize()
            g, s = O.spmv_gold_acc64(csr, xh)
            got = plan.x(0).cpu().numpy()
            ok, worst = O.strict_check(csr, got, g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=parts))
            assert ok, (step, worst)
            xh = got                                       # follow the device iterate (errors do not compound in the check)
    finally:
        plan.close()


@gpu
def test_plan_rccl_backend_one_rank():
    """The RCCL backend end to end on a 1-GPU box: librccl is loaded at run time, a one-rank communicator is
    made (ncclCommInitAll), every step issues its ncclAllGather, the row all-gather its ncclBroadcast."""
    csr, x = _matrix("power_law", 5, np.float64)
    off = csr.row_offsets.astype(np.int64)
    row_split, nz_split = MG.partition(off, 1)
    plan = MG.MgPlan(row_split, nz_split, csr.cols, torch.float64, [0], [0], exchange=MG.EXCHANGE_RCCL)
    try:
        lo = MG.local_offsets(off, 0, csr.rows, 0, csr.nnz)
        plan.set_part(0, torch.from_numpy(csr.values).cuda(), torch.from_numpy(lo).cuda(), torch.from_numpy(csr.column_indices).cuda())
        assert plan.info()["exchange"] == MG.EXCHANGE_RCCL
        plan.x(0).copy_(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
        plan.csrmv(); plan.synchronize()
        y = plan.y(0).cpu().numpy()
        g, s = O.spmv_gold_acc64(csr, x)
        ok, worst = O.strict_check(csr, y, g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, 8))
        assert ok, worst
        plan.allgather_rows(); plan.synchronize()
        assert np.array_equal(plan.x(0).cpu().numpy(), y)
    finally:
        plan.close()
    # several RCCL ranks on one device are refused up front (one rank per device), not left to hang
    row_split, nz_split = MG.partition(off, 2)
    with pytest.raises(M.MspmvError):
        MG.MgPlan(row_split, nz_split, csr.cols, torch.float64, [0, 1], [0, 0], exchange=MG.EXCHANGE_RCCL)


@gpu
def test_plan_multi_process_form_single_rank():
    """What bench.py --gpus N does per rank (one part per process, ncclCommInitRank with a shipped id), with N = 1."""
    csr, x = _matrix("uniform", 9, np.float32)
    off = csr.row_offsets.astype(np.int64)
    row_split, nz_split = MG.partition(off, 1)
    # local_parts (1) == parts (1) would be the single-process form; a 1-rank job is exactly that
    plan = MG.MgPlan(row_split, nz_split, csr.cols, torch.float32, [0], [0], exchange=MG.EXCHANGE_RCCL, id128=MG.unique_id())
    try:
        lo = MG.local_offsets(off, 0, csr.rows, 0, csr.nnz)
        plan.set_part(0, torch.from_numpy(csr.values).cuda(), torch.from_numpy(lo).cuda(), torch.from_numpy(csr.column_indices).cuda())
        plan.x(0).copy_(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
        for _ in range(3):
            plan.csrmv()
        plan.synchronize()
        g, s = O.spmv_gold_acc64(csr, x)
        ok, worst = O.strict_check(csr, plan.y(0).cpu().numpy(), g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, 4))
        assert ok, worst
    finally:
        plan.close()


@gpu
def test_c5_full_size_eight_parts_every_row():
    """BASELINE.json config 5 at its stated size: fp64 R-MAT scale 26 (67 108 864^2), 2 000 000 000 edges, cut by
    mspmv_mg_partition into 8 parts that run through the HIP CsrMV one after the other on this GPU (the
    single-process plan), carries exchanged by peer reads.  EVERY row of y is checked against the oracle's
    fp64-accumulated gold under the strict bound; the 8-part result is also compared with the one-GPU call
    on the whole matrix.  Set MSPMV_TEST_C5_SCALE / _EDGES to shrink it on a smaller device."""
    from merge_spmv_amd import generators as G
    scale = int(os.environ.get("MSPMV_TEST_C5_SCALE", "26"))
    edges = int(os.environ.get("MSPMV_TEST_C5_EDGES", "2000000000"))
    free, total = torch.cuda.mem_get_info()
    if total < 200 * 2**30 and scale == 26:
        pytest.skip("needs the 288 GB of an MI355X for the full-size C5 matrix")
    parts = 8
    A = G.rmat_csr(scale, edges, dtype=torch.float64, seed=G.SEED_C5)
    n = 1 << scale
    assert A.rows == n and A.nnz == edges
    x = G.uniform_pm1(G.SEED_C5 + 2, n, torch.float64, "cuda")
    xh = x.cpu().numpy()
    off = A.row_offsets.cpu().numpy().astype(np.int64)
    row_split, nz_split = MG.partition(off, parts)
    total_items = n + edges
    for g in range(parts):      # equal swaths of the merge path, whatever the skew
        items = (row_split[g + 1] - row_split[g]) + (nz_split[g + 1] - nz_split[g])
        assert items <= -(-total_items // parts)
    # one-GPU reference run on the whole matrix (config 5 at G = 1)
    y_whole = M.csrmv(A.values, A.row_offsets, A.column_indices, x, num_cols=n).cpu().numpy()
    plan = MG.MgPlan(row_split, nz_split, n, torch.float64, list(range(parts)), [0] * parts, exchange=MG.EXCHANGE_PEER)
    gold = np.zeros(n + 1, np.float64); sabs = np.zeros(n + 1, np.float64)
    try:
        for g in range(parts):
            lo = MG.local_offsets(off, row_split[g], row_split[g + 1], nz_split[g], nz_split[g + 1])
            a, b = int(nz_split[g]), int(nz_split[g + 1])
            vals = A.values[a:b].clone(); cols = A.column_indices[a:b].clone()
            plan.set_part(g, vals, torch.from_numpy(lo).cuda(), cols)
            # the oracle on this part (host): partial sums of its local rows, incl. the open last row
            local = O.Csr(lo.size - 1, n, lo, cols.cpu().numpy(), vals.cpu().numpy())
            gl, sl = O.spmv_gold_acc64(local, xh)
            r0 = int(row_split[g])
            gold[r0: r0 + gl.size] += gl
            sabs[r0: r0 + sl.size] += sl
        del A
        torch.cuda.empty_cache()
        plan.x(0).copy_(x); torch.cuda.synchronize()
        plan.csrmv(); plan.synchronize()
        y = _gather_y(plan, row_split, n, np.float64)
    finally:
        plan.close()
    whole = O.Csr(n, n, off.astype(np.int32), np.zeros(0, np.int32), np.zeros(0, np.float64))   # row lengths for the bound
    ok, worst = O.strict_check(whole, y, gold[:n], sabs[:n], items_per_thread=M.serial_sum_depth(n, n, edges, 8, extra=parts))
    assert ok, worst
    ok, worst_whole = O.strict_check(whole, y_whole, gold[:n], sabs[:n], items_per_thread=M.serial_sum_depth(n, n, edges, 8))
    assert ok, worst_whole
    print(f"\nC5 scale {scale}, {edges} edges, {parts} parts: worst |error|/bound = {worst:.3g} (8 parts), {worst_whole:.3g} (one GPU)")


def _line_and_detail(stdout, detail_path):
    """bench.py's contract: the LAST stdout line is one JSON object under 4 KB; everything longer is in the --detail file.  Returns the
    detail record (a superset of the line's numbers) after checking the line against it."""
    import json
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{"), lines[-3:]          # the JSON line is the LAST line of stdout
    assert len(lines[-1].encode()) < 4096, len(lines[-1])
    line = json.loads(lines[-1])
    detail = json.load(open(detail_path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype"):
        assert line[k] == detail[k], k
    assert "roofline" in line and "frac" in line["roofline"]
    return detail


def _run_bench(env_extra, nproc, *args, timeout=600, preflight=False):
    import subprocess, sys, socket, tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ); env.update(env_extra)
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--detail", detail, *args]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        if preflight:                    # (a line only: nothing goes to a detail file)
            import json
            return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return _line_and_detail(r.stdout, detail)


@gpu
@pytest.mark.parametrize("kind,prec,nproc", [("giant", "f64", 2), ("short", "f32", 2), ("empty_parts", "f64", 4)])
def test_ipc_backend_two_processes_sharing_the_device(kind, prec, nproc):
    """The hipIpc peer backend of the one-process-per-GPU form (MSPMV_MG_EXCHANGE_IPC): `nproc` processes, one part each, all on
    cuda:0 of this box; carries go through hipIpc-opened mailboxes tagged with the step number, rows through the opened x
    replicas.  The worker checks every row against the oracle for repeated and iterated SpMV (tests/mg_ipc_worker.py)."""
    import subprocess, sys, socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "mg_ipc_worker.py"), kind, prec]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and f"IPC-OK {kind} {prec} {nproc}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@gpu
def test_hot_columns_are_automatic():
    """The parts' hot-column plans are AUTOMATIC (mspmv_mg_plan_hot_columns(-1), the default of a new plan): built per part when its
    matrix is attached iff x is beyond the Infinity Cache and the on-device sample says the columns come back (scale-free), not for
    uniformly spread columns or a band; y bit for bit the same with, without and forced; mspmv_mg_plan_exchange_ms answers.
    A process of its own: the cache size is faked through the environment, which the library reads once."""
    import subprocess, sys
    env = dict(os.environ); env["MSPMV_FAKE_INFINITY_CACHE_MIB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mg_auto_hot_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "AUTO-HOT-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@gpu
def test_bench_multi_rank_path_on_one_device():
    """bench.py --gpus N --workload c5 (BASELINE config 5, one R-MAT matrix cut N ways) end to end on this one-GPU box, at a
    reduced scale: (a) 2 ranks sharing the device, carries over gloo through the Python twin; (b) ONE rank forced through
    the N > 1 code path: nccl process group, shipped RCCL id, the C operator's multi-process form with its
    ncclAllGather inside the timed loop, the same-job single-GPU leg."""
    small = ("--steps", "3", "--warmup", "1", "--c5-scale", "18", "--c5-edges", "3000000", "--workload", "c5")
    out = _run_bench({"MSPMV_BENCH_ONE_DEVICE": "1", "MSPMV_BENCH_BACKEND": "gloo"}, 2, *small)
    assert out["per_rank"]["tile_ms_max"] >= out["per_rank"]["tile_ms_min"] > 0 and out["per_rank"]["nnz_per_rank_max"] > 0
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["dtype"] == "f64" and out["value"] > 0
    assert "C5 R-MAT scale 18" in out["config"]["workload"] and "merge-path diagonal split over 2 GPUs" in out["config"]["partition"]
    assert out["single_gpu_same_workload"]["n_gpus"] == 1 and out["single_gpu_same_workload"]["value"] > 0
    # (c) 2 ranks sharing the device through the C operator's hipIpc backend (no collective in the step)
    out = _run_bench({"MSPMV_BENCH_ONE_DEVICE": "1", "MSPMV_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, 2, *small, "--exchange", "ipc")
    assert out["n_gpus"] == 2 and out["exchange"]["exchange"] == MG.EXCHANGE_IPC and out["value"] > 0
    # rank 0's tiles are the first tiles of the one-GPU call of the same job: its rows agree bit for bit
    cmp = out["single_gpu_same_workload"]["rank0_rows_vs_single_gpu"]
    # ... except inside rank 0's last tile, which ends where the part ends instead of a tile_items further on
    assert cmp["same_tiling"] and (cmp["not_bitwise_equal"] == 0 or cmp["rows"] - cmp["first_differing_row"] <= cmp["tile_items"])
    # (d) an exchange that cannot be set up (two RCCL ranks on one device are refused) makes every rank fall back together
    out = _run_bench({"MSPMV_BENCH_ONE_DEVICE": "1", "MSPMV_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                      "MSPMV_BENCH_FORCE_C_OPERATOR": "1"}, 2, *small, "--exchange", "rccl")
    assert out["exchange"]["exchange"] == MG.EXCHANGE_IPC and out["exchange"]["fallbacks"][0]["exchange"] == "rccl" and out["value"] > 0
    assert out["per_rank"]["tile_ms_min"] > 0
    # (e) --preflight: communicator / exchange set-up, one step, a verdict in seconds
    out = _run_bench({"MSPMV_BENCH_ONE_DEVICE": "1", "MSPMV_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, 2, "--preflight", preflight=True)
    assert out["preflight"]["ok"] and out["preflight"]["ranks"] == 2 and out["preflight"]["seconds"] < 60
    out = _run_bench({"MSPMV_BENCH_FORCE_MG": "1"}, 1, *small)
    assert out["n_gpus"] == 1 and "C5 R-MAT scale 18" in out["config"]["workload"]
    assert out["exchange"]["exchange"] == MG.EXCHANGE_RCCL and out["exchange"]["carry_bytes_per_step"] == 8
    assert out["roofline"]["kernel_ms"] > 0 and out["single_gpu_same_workload"]["ms_per_step"] > 0


def _run_bench_plain(env_extra, *args, timeout=900):
    """`python bench.py --gpus N ...` with NO launcher and no WORLD_SIZE in the environment: bench.py starts its ranks itself."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)


@gpu
def test_bench_gpus_n_starts_its_own_ranks():
    """the driver's plain `python bench.py --gpus N --steps K --warmup W` must produce the line by itself.  Here
    N = 2 on this one-GPU box (MSPMV_BENCH_ONE_DEVICE=1: both ranks on cuda:0, gloo process group, the C operator with its
    exchange falling back rccl -> hipIpc on every rank together); the JSON line is the last line of stdout.  The headline is BASELINE
    config 2 weak-scaled (every rank holds the N = 1 line's matrix, so a scaling series compares like with like); config 5, one R-MAT
    matrix cut N ways, rides as the `c5_strong` leg of the same job."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        r = _run_bench_plain({"MSPMV_BENCH_ONE_DEVICE": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--c5-scale", "18", "--c5-edges", "3000000",
                             "--c5-single-gpu-leg", "--detail", detail)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert len(r.stdout.splitlines()[-1].encode()) < 4096
        out = _line_and_detail(r.stdout, detail)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak" and out["dtype"] == "f32" and out["value"] > 0
    assert "C2 uniform CSR: 6250000 x 3125000" in out["config"]["workload"] and out["per_rank"]["nnz_per_rank_max"] == 100_000_000
    assert out["exchange"]["exchange"] == MG.EXCHANGE_IPC and out["exchange"]["fallbacks"][0]["exchange"] == "rccl"
    c5 = out["c5_strong"]
    assert c5["n_gpus"] == 2 and c5["scaling"] == "strong" and c5["dtype"] == "f64" and c5["value"] > 0 and "C5 R-MAT scale 18" in c5["config"]["workload"]
    assert c5["per_rank"]["nnz_per_rank_max"] > 0 and c5["single_gpu_same_workload"]["value"] > 0
    cmp = c5["single_gpu_same_workload"]["rank0_rows_vs_single_gpu"]
    assert cmp["not_bitwise_equal"] == 0 or cmp["rows"] - cmp["first_differing_row"] <= cmp["tile_items"]


def test_bench_gpus_n_self_launch_reports_failure_of_its_ranks():
    """The same entry on a box without a GPU: the launcher is started, both ranks refuse ("needs a GPU"), and the plain command
    returns their failure instead of a line -- no SystemExit about WORLD_SIZE any more."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = _run_bench_plain({}, "--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "needs torch.distributed.run" not in r.stderr + r.stdout
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
