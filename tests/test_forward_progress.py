"""GPU tests (-m gpu): nothing in a one-launch call depends on another workgroup making progress.

Rows longer than the snap limit hand their pieces from tile to tile as tagged records (mspmv_kernels.hpp: LookBack).  The
reference's fp64 fix-up spins on such records without bound (cub/agent/single_pass_scan_operators.cuh:620-639); here the tile
in which the row ends polls a bounded number of times and then computes the missing sum from the matrix itself, so a stream
restricted to a quarter of the CUs, a long kernel hogging the device, or a poll budget of ONE look (mspmv_set_record_polls)
change the time of a call and, by a re-association, the last bits of such a row -- never whether y is complete and right.
"""
import ctypes

import numpy as np
import pytest

from oracle import oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import merge_spmv_amd as M_
    M_.load_library()
    M_.set_tuning(4); M_.set_tuning(8); M_.set_record_polls(0)
    yield M_
    M_.set_record_polls(0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _matrix(kind, dtype, rng):
    if kind == "giant":          # one row over ~150 tiles between short rows (the reference's --wheel shape, larger)
        rows = 30000
        lens = rng.integers(0, 4, rows); lens[rows // 3] = 400_000
    elif kind == "many_long":    # hundreds of rows of 2-15 tiles each, back to back: many tiles wait for many records.  (Under 8 M path items:
        rows = 3000              #  beyond that a matrix whose rows are ALL long runs the classic three launches, mspmv_api.hip: long_rows_rule)
        lens = np.where(np.arange(rows) % 7 == 0, rng.integers(4000, 28_000, rows), rng.integers(0, 30, rows))
    elif kind == "power_law":
        rows = 60000
        lens = np.minimum((rng.pareto(1.05, rows) * 3).astype(np.int64), 200_000)
    else:
        raise KeyError(kind)
    lens = np.asarray(lens, np.int64)
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1]); cols = 50_000
    col = np.sort(rng.integers(0, cols, nnz).astype(np.int32)) if kind == "giant" else rng.integers(0, cols, nnz).astype(np.int32)
    csr = O.Csr(rows, cols, off.astype(np.int32), col, rng.uniform(-1, 1, nnz).astype(dtype))
    return csr, rng.uniform(-1, 1, cols).astype(dtype)


def _run(M, csr, x, ws, stream=None):
    tdt = torch.float32 if csr.values.dtype == np.float32 else torch.float64
    y = torch.full((csr.rows,), float("nan"), dtype=tdt, device="cuda")
    M.csrmv(_run.d[0], _run.d[1], _run.d[2], _run.d[3], y=y, num_cols=csr.cols, workspace=ws, stream=stream)
    torch.cuda.synchronize()
    return y


def _check(M, csr, x, y, extra=2):
    g, s = O.spmv_gold_acc64(csr, x)
    ok, worst = O.strict_check(csr, y.cpu().numpy(), g, s,
                               items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, csr.values.dtype.itemsize, extra=extra))
    assert ok, worst


def _records_are_clean(M, csr, ws):
    """every record slot of the temp storage is (0, 0) once a launch has ended (mspmv_kernels.hpp: "EVERY RECORD SLOT IS CLEAN")"""
    info = M.launch_info(csr.rows, csr.nnz, csr.values.dtype.itemsize)
    torch.cuda.synchronize()
    lo, n = info["records_offset"], (info["num_tiles"] + info["num_tiles"] // 64 + 1) * 16
    assert int(ws.buffer[lo: lo + n].view(torch.int64).count_nonzero().item()) == 0


@pytest.mark.parametrize("kind", ["giant", "many_long", "power_law"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_one_look_poll_budget_recomputes_and_stays_right(M, kind, prec):
    """mspmv_set_record_polls(1 / -1): a tile in which a long row ends looks ONCE for each record, or not at all, and otherwise
    adds up the row's earlier nonzeros itself.  Every row within the strict bound of the oracle; afterwards (every record slot clean again: a
    cancelled slot's late record is wiped by its publisher) the default call gives bit for bit what it gave before, and the record region of
    the temp storage holds nothing but zeros."""
    dtype = np.float32 if prec == "f32" else np.float64
    csr, x = _matrix(kind, dtype, np.random.default_rng(len(kind) + (prec == "f64")))
    _run.d = (dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x))
    ws = M.CsrMVWorkspace(csr.rows, csr.nnz, torch.float32 if prec == "f32" else torch.float64)
    ws.buffer.zero_()                         # (slots no tile of this matrix ever publishes to keep whatever the allocation held: zeros here)
    y_ref = _run(M, csr, x, ws)
    _check(M, csr, x, y_ref)
    _records_are_clean(M, csr, ws)
    diag = M.launch_info(csr.rows, csr.nnz, csr.values.dtype.itemsize)["diag_offset"]
    epoch = lambda: int(ws.buffer[diag + 4: diag + 8].view(torch.int32).item())          # (the episode counter)
    e0 = epoch()
    assert epoch() == e0                      # (ordinary calls never touch it)
    try:
        for polls in (1, -1, -1, 1):          # one look; never look (every such tile recomputes)
            M.set_record_polls(polls)
            y1 = _run(M, csr, x, ws)
            assert not torch.isnan(y1).any()
            _check(M, csr, x, y1)
    finally:
        M.set_record_polls(0)
    assert epoch() != e0                      # the recomputing path was taken
    _records_are_clean(M, csr, ws)
    for _ in range(3):
        assert torch.equal(_run(M, csr, x, ws), y_ref)
    _records_are_clean(M, csr, ws)


def test_captured_call_replays_correctly_after_a_recomputing_episode(M):
    """A captured call replays the tags it was captured with.  No record outlives its launch -- a slot whose consumer gave up is
    cancelled and its late record wiped by the publisher -- so the replay -- same call tag, NEW x -- finds nothing to mistake for its own."""
    rng = np.random.default_rng(5)
    csr, x = _matrix("many_long", np.float64, rng)
    _run.d = (dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x))
    ws = M.CsrMVWorkspace(csr.rows, csr.nnz, torch.float64)
    y = torch.empty(csr.rows, dtype=torch.float64, device="cuda")
    xs = _run.d[3]
    call = lambda: M.csrmv(_run.d[0], _run.d[1], _run.d[2], xs, y=y, num_cols=csr.cols, workspace=ws)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call()
        with torch.cuda.graph(g, stream=s):
            call()
    g.replay(); torch.cuda.synchronize()
    _check(M, csr, x, y)
    try:
        M.set_record_polls(-1)
        for _ in range(3):
            _check(M, csr, x, _run(M, csr, x, ws))
    finally:
        M.set_record_polls(0)
    for k in range(3):
        x2 = rng.uniform(-1, 1, csr.cols)
        xs.copy_(dev(x2))
        g.replay(); torch.cuda.synchronize()
        _check(M, csr, x2, y)


def _cu_masked_stream(fraction):
    hip = ctypes.CDLL("libamdhip64.so")
    props = torch.cuda.get_device_properties(0)
    cus = props.multi_processor_count
    words = (cus + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    keep = max(int(cus * fraction), 8)
    for c in range(keep):
        mask[c // 32] |= 1 << (c % 32)
    stream = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), ctypes.c_uint32(words), mask)
    if rc != 0:
        pytest.skip(f"hipExtStreamCreateWithCUMask returned {rc}")
    return hip, stream, keep


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_giant_row_on_a_stream_restricted_to_a_quarter_of_the_cus(M, prec):
    """(a) of VERDICT r03 next #7: the residency the dispatcher derives from device attributes is four times what this stream has."""
    dtype = np.float32 if prec == "f32" else np.float64
    hip, stream, keep = _cu_masked_stream(0.25)
    try:
        for kind in ("giant", "many_long"):
            csr, x = _matrix(kind, dtype, np.random.default_rng(11))
            _run.d = (dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x))
            ws = M.CsrMVWorkspace(csr.rows, csr.nnz, torch.float32 if prec == "f32" else torch.float64)
            torch.cuda.synchronize()
            for _ in range(3):
                y = _run(M, csr, x, ws, stream=stream.value)
                assert hip.hipStreamSynchronize(stream) == 0
                _check(M, csr, x, y)
    finally:
        hip.hipStreamDestroy(stream)


def test_giant_row_beside_a_kernel_that_hogs_the_device(M):
    """(b): a long kernel on another stream holds workgroup slots while the call runs (big GEMMs queued first)."""
    csr, x = _matrix("many_long", np.float64, np.random.default_rng(12))
    _run.d = (dev(csr.values), dev(csr.row_offsets), dev(csr.column_indices), dev(x))
    ws = M.CsrMVWorkspace(csr.rows, csr.nnz, torch.float64)
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
    hog = torch.cuda.Stream()
    ys = []
    for rep in range(3):
        with torch.cuda.stream(hog):
            for _ in range(6):
                a = torch.tanh(a @ a) * 0.01
        y = torch.full((csr.rows,), float("nan"), dtype=torch.float64, device="cuda")
        M.csrmv(_run.d[0], _run.d[1], _run.d[2], _run.d[3], y=y, num_cols=csr.cols, workspace=ws)
        ys.append(y)
    torch.cuda.synchronize()
    for y in ys:
        _check(M, csr, x, y)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_group_records_of_rows_that_span_hundreds_of_tiles(M, prec):
    """A row over more than 64 tiles is summed through GROUP records (kernels: "GROUP RECORDS"): every 64th piece folds its 63
    predecessors into one record and the tile in which the row ends takes groups plus the singles at both ends.  Rows that begin at
    every kind of offset against the group boundaries (preceded by 0 ... 200 tiles of short rows), two long rows back to back, one
    row that is nearly the whole matrix; values and x small integers, so every sum is exact in either precision and y must EQUAL
    the integer result -- with the default poll budget, with one look per record and with none (leaders and consumers then
    compute their sums from the matrix), and bit for bit the same on a second call."""
    dtype = np.float32 if prec == "f32" else np.float64
    tdt = torch.float32 if prec == "f32" else torch.float64
    rng = np.random.default_rng(64)
    info = M.launch_info(1000, 3_000_000, dtype().itemsize)
    tile = info["tile_items"]
    cases = []
    for lead_tiles in (0, 1, 37, 63, 64, 65, 200):
        lens = [np.full(lead_tiles * tile // 4, 3), [70 * tile + 11], rng.integers(0, 5, 300), [130 * tile - 5], [64 * tile], rng.integers(0, 5, 50)]
        cases.append(np.concatenate([np.asarray(l, np.int64) for l in lens]))
    cases.append(np.concatenate([rng.integers(0, 3, 40), [400 * tile + 1], rng.integers(0, 3, 40)]).astype(np.int64))
    for lens in cases:
        rows = lens.size
        off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
        nnz = int(off[-1]); cols = 4096
        csr = O.Csr(rows, cols, off.astype(np.int32), rng.integers(0, cols, nnz).astype(np.int32), rng.integers(1, 3, nnz).astype(dtype))
        x = rng.integers(1, 3, cols).astype(dtype)
        expect = np.add.reduceat(np.concatenate([csr.values.astype(np.float64) * x[csr.column_indices], [0.0]]), np.minimum(off[:-1], nnz))
        expect = np.where(lens > 0, expect, 0.0)
        assert expect.max() < 2 ** 24
        d = [dev(v) for v in (csr.values, csr.row_offsets, csr.column_indices, x)]
        ws = M.CsrMVWorkspace(csr.rows, csr.nnz, tdt)
        try:
            for polls in (0, 1, -1):
                M.set_record_polls(polls)
                ys = []
                for _ in range(2):
                    y = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
                    M.csrmv(*d, y=y, num_cols=cols, workspace=ws)
                    torch.cuda.synchronize()
                    ys.append(y.cpu().numpy())
                assert np.array_equal(ys[0], expect.astype(dtype)), (prec, polls, rows, int((ys[0] != expect).sum()))
                assert np.array_equal(ys[0], ys[1])
        finally:
            M.set_record_polls(0)
