"""Clock-scheduled column bands (csrc/mspmv_tdm.hpp; include/mspmv_dev.h: mspmv_set_tdm): the one-pass form of the column-band
organisation that serves the calls the passes are offered to.  A block sorts its tile's nonzeros by column band in LDS and gathers
band by band, the band "on air" read off the chip-wide clock; products and reduction are those of the classic one-sweep tile kernel,
so y must be BIT FOR BIT that kernel's -- whatever the band width, the slot length, the lookahead, i.e. whatever the clock does.
CPU: argument checks, the policy.  GPU: the bitwise comparison on every shape family (forced on small matrices), the reference's
golden matrices, alpha / beta, graph replays, and the automatic path on a matrix the windows accept."""
import os

import numpy as np
import pytest

import merge_spmv_amd as M
from oracle import oracle as O

torch = pytest.importorskip("torch")
gpu = pytest.mark.gpu

NO_FUSED = 16
TWO_LAUNCH = 0x40000000


def test_set_tdm_argument_checks():
    M.use_library("dev")
    lib = M.load_library()
    for vb in (4, 8):
        for ok in ((0, 0, 0, 0), (-1, 0, 0, 0), (1, 500, 3, 12), (5, 100000, 33, 20), (1, 1, 1, 1)):
            assert lib.mspmv_set_tdm(vb, *ok) == 0
        for bad in ((0, -1, 0, 0), (0, 0, -1, 0), (0, 0, 34, 0), (0, 0, 0, 21), (0, 0, 0, -1)):
            assert lib.mspmv_set_tdm(vb, *bad) == 1
        assert lib.mspmv_set_tdm(vb, 0, 0, 0, 0) == 0
    assert lib.mspmv_set_tdm(2, 0, 0, 0, 0) == 1


def test_the_clocked_form_serves_what_the_passes_are_offered():
    """host-side rule (csrc/mspmv_api.hip: band_passes_for, tdm_shift_for): the same candidates -- being a candidate costs a matrix the
    windows then refuse the one-launch kernel, so the range was not widened for the faster form"""
    assert M.active_library() == "product"
    mib = lambda m, vb: m * 2**20 // vb
    rows, nnz = 3_000_000, 96_000_000
    assert M.band_passes(rows, mib(32, 4), nnz, 4) == 4 and M.band_passes(rows, mib(48, 4), nnz, 4) == 0
    try:
        M.set_tdm(4, -1)
        assert M.band_passes(rows, mib(32, 4), nnz, 4) == 4 and M.band_passes(rows, mib(48, 4), nnz, 4) == 0
    finally:
        M.set_tdm(4, 0)


def _random(rng, rows, cols, lens, dtype):
    off = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=off[1:])
    nnz = int(off[-1])
    col = rng.integers(0, cols, nnz).astype(np.int32)
    return O.Csr(rows, cols, off.astype(np.int32), col, rng.uniform(-1, 1, nnz).astype(dtype))


SHAPES = {
    "short_rows": lambda rng: (60000, 50000, rng.integers(0, 12, 60000)),
    "rows_of_32": lambda rng: (40000, 300000, np.full(40000, 32)),
    "power_law": lambda rng: (20000, 30000, np.minimum((rng.pareto(1.1, 20000) * 2).astype(np.int64), 20000)),
    "giant_row": lambda rng: (3000, 100000, np.where(np.arange(3000) == 1500, 300000, rng.integers(0, 3, 3000))),
    "mostly_empty": lambda rng: (400000, 7000, np.where(np.arange(400000) % 97 == 0, 50, 0)),
    "all_empty": lambda rng: (5000, 500, np.zeros(5000, np.int64)),
    "few_cols": lambda rng: (50000, 3, rng.integers(0, 3, 50000)),
    "one_tile_and_a_bit": lambda rng: (100, 4000, np.full(100, 30)),
    "ragged_tail": lambda rng: (30001, 70001, np.where(np.arange(30001) == 30000, 3, rng.integers(0, 40, 30001))),
}


def _reset():
    for vb in (4, 8):
        M.set_tuning(vb); M.set_band_passes(vb, 0); M.set_tdm(vb, 0)


def _classic(vb, ipt):
    """the reference: the classic three launches' one-sweep tile kernel on the same tile shape"""
    M.set_tuning(vb, 256, ipt, NO_FUSED | TWO_LAUNCH); M.set_band_passes(vb, -1); M.set_tdm(vb, 0)


def _clocked(vb, ipt, slot=0, look=0, shift=0):
    M.set_tuning(vb, 256, ipt, NO_FUSED); M.set_band_passes(vb, 3); M.set_tdm(vb, 1, slot, look, shift)


# (slot per mille, lookahead + 1, band shift): defaults; narrow bands; a clock that is far too fast / far too slow with every band
# eligible; one band in all (shift 20)
CLOCKS = [(0, 0, 0), (0, 0, 11), (1, 1, 13), (100000, 33, 12), (0, 2, 20)]


@gpu
@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("prec,ipt", [("f32", 11), ("f64", 11), ("f64", 7)])
def test_clocked_bands_are_bitwise_the_classic_one_sweep(shape, prec, ipt):
    dtype, tdt = (np.float32, torch.float32) if prec == "f32" else (np.float64, torch.float64)
    vb = 4 if prec == "f32" else 8
    rng = np.random.default_rng(len(shape) * 7 + ipt)
    rows, cols, lens = SHAPES[shape](rng)
    csr = _random(rng, rows, cols, np.asarray(lens, np.int64), dtype)
    x = rng.uniform(-1, 1, cols).astype(dtype)
    d = lambda a: torch.from_numpy(a).cuda()
    val, off, col, dx = d(csr.values), d(csr.row_offsets), d(csr.column_indices), d(x)
    ws = M.CsrMVWorkspace(rows, csr.nnz, tdt)
    y0 = rng.uniform(-1, 1, rows).astype(dtype)
    try:
        _classic(vb, ipt)
        ref = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, dx, y=ref, num_cols=cols, workspace=ws)
        ref_ab = d(y0.copy())
        M.csrmv(val, off, col, dx, y=ref_ab, num_cols=cols, workspace=ws, alpha=-1.5, beta=0.5)
        torch.cuda.synchronize()
        g, s = O.spmv_gold_acc64(csr, x)
        ok, worst = O.strict_check(csr, ref.cpu().numpy(), g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, vb))
        assert ok, worst
        for slot, look, shift in CLOCKS:
            _clocked(vb, ipt, slot, look, shift)
            taken = M.band_passes(rows, cols, csr.nnz, vb) == 3           # (fewer than 3 columns, 4 nonzeros, 3 rows: the ordinary path)
            y = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
            M.csrmv(val, off, col, dx, y=y, num_cols=cols, workspace=ws)
            if taken:
                assert torch.equal(y, ref), (shape, prec, ipt, slot, look, shift, int((y != ref).sum()))
            else:
                assert O.strict_check(csr, y.cpu().numpy(), g, s)[0]
            yab = d(y0.copy())
            M.csrmv(val, off, col, dx, y=yab, num_cols=cols, workspace=ws, alpha=-1.5, beta=0.5)
            if taken:
                assert torch.equal(yab, ref_ab), (shape, prec, ipt, slot, look, shift, "alpha/beta")
    finally:
        _reset()


def _golden_cases():
    from conftest import load_golden
    return load_golden("matrices.json")["cases"]


@gpu
@pytest.mark.parametrize("case", _golden_cases(), ids=[c["label"] for c in _golden_cases()])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_clocked_bands_on_the_reference_golden_matrices(case, prec):
    """The reference's own test inputs under its own protocol (x = 1, SpmvGold, its PASS rule: gpu_spmv.cu:521-525, utils.h:692-742)
    with the clocked form forced: exact where the arithmetic is exact, tile coordinates and carry keys bit for bit the oracle's."""
    from conftest import ROOT
    dtype, tdt = (np.float32, torch.float32) if prec == "f32" else (np.float64, torch.float64)
    vb = 4 if prec == "f32" else 8
    args = [os.path.join(ROOT, case["args"][0])] if case["kind"] == "mtx" else case["args"]
    csr = O.make(case["kind"], *args, dtype=dtype)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    try:
        _clocked(vb, 11, 0, 0, 6)
        ws = M.CsrMVWorkspace(csr.rows, csr.nnz, tdt)
        info = M.launch_info(csr.rows, csr.nnz, vb)
        x = np.ones(csr.cols, dtype)
        y = torch.full((csr.rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(d(csr.values), d(csr.row_offsets), d(csr.column_indices), d(x), y=y, num_cols=csr.cols, workspace=ws)
        torch.cuda.synchronize()
        yh = y.cpu().numpy()
        gold = O.spmv_gold(csr, x)
        assert O.compare_results(yh, gold) == 0
        if case["kind"] != "mtx":
            assert np.array_equal(yh, gold)
        g, s = O.spmv_gold_acc64(csr, x)
        ok, worst = O.strict_check(csr, yh, g, s, items_per_thread=M.serial_sum_depth(csr.rows, csr.cols, csr.nnz, vb, extra=3))
        assert ok, worst
        coords, keys, vals = M.debug_read_tiles(ws.buffer, csr.rows, csr.nnz, vb)
        want = O.tile_coords(csr, info["tile_items"])
        assert np.array_equal(coords, want[: info["num_tiles"] + 1])
        _, ck, cv = O.tiled_csrmv(csr, x, info["tile_items"])
        assert np.array_equal(keys, ck)
    finally:
        _reset()


def _uniform(rows, cols, per_row, tdt, seed=5):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    col = torch.randint(0, cols, (rows, per_row), generator=g, device="cuda", dtype=torch.int32).sort(dim=1).values.reshape(-1).contiguous()
    val = (torch.rand(rows * per_row, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    off = (torch.arange(rows + 1, device="cuda", dtype=torch.int64) * per_row).to(torch.int32)
    x = (torch.rand(cols, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)
    return val, off, col, x


@gpu
@pytest.mark.parametrize("prec,cols", [("f32", 2_400_000), ("f64", 1_600_000), ("f32", 9_000_000)])
def test_automatic_call_takes_the_clocked_form_on_spread_columns(prec, cols):
    """34 M uniformly spread nonzeros, x of 9-36 MB: the PRODUCT library's stateless call is a candidate, all 64 windows say "spread",
    and y is bit for bit the classic one-sweep kernel's -- plain, prepared, and replayed from a captured graph; with the clocked form
    switched off the passes run instead (another association)."""
    tdt = torch.float32 if prec == "f32" else torch.float64
    vb = 4 if prec == "f32" else 8
    rows, per_row = 1_062_500, 32
    val, off, col, x = _uniform(rows, cols, per_row, tdt)
    nnz = rows * per_row
    ws = M.CsrMVWorkspace(rows, nnz, tdt)
    assert M.active_library() == "product" and M.band_passes(rows, cols, nnz, vb) >= 2
    y = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
    M.csrmv(val, off, col, x, y=y, num_cols=cols, workspace=ws)
    torch.cuda.synchronize()
    assert int(M.debug_band_windows(ws, rows, nnz, vb).sum()) == 64
    # prepared call (the stand-alone sampler instead of the coordinate pass's sampling blocks), graph replay: the same bits
    ws2 = M.CsrMVWorkspace(rows, nnz, tdt).prepare(off)
    yp = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
    M.csrmv(val, off, col, x, y=yp, num_cols=cols, workspace=ws2)
    assert torch.equal(y, yp)
    stream = torch.cuda.Stream()
    yg = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
    with torch.cuda.stream(stream):
        M.csrmv(val, off, col, x, y=yg, num_cols=cols, workspace=ws)      # warm-up outside the capture
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            M.csrmv(val, off, col, x, y=yg, num_cols=cols, workspace=ws)
    for _ in range(3):
        yg.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, yg)
    try:
        M.set_tuning(vb, flags=TWO_LAUNCH); M.set_band_passes(vb, -1)
        ref = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
        M.csrmv(val, off, col, x, y=ref, num_cols=cols, workspace=ws)
        assert torch.equal(y, ref), "the automatic call is not the classic one-sweep result: the clocked form did not run (or is wrong)"
        M.set_tuning(vb); M.set_band_passes(vb, 0); M.set_tdm(vb, -1)
        if M.band_passes(rows, cols, nnz, vb) >= 2:
            yb = torch.full((rows,), float("nan"), dtype=tdt, device="cuda")
            M.csrmv(val, off, col, x, y=yb, num_cols=cols, workspace=ws)
            assert not torch.equal(y, yb), "the passes re-associate: identical bits mean they did not run"
    finally:
        _reset()
